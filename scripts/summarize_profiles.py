"""Copies the rocprofv3 summaries of one profiling round (scripts/profile_round.sh <tag> <workload>) from gpurun_out/ into
profiles/ and derives the per-step HBM traffic that bench.py attaches to its `roofline.traffic`.

FETCH_SIZE is in KiB.  The guide (MI355X_MICROARCH.md, HBM section) says gfx950 tallies 128-byte requests of wide
coalesced reads at 64 bytes (-> x2) and that other access widths must be calibrated on a known byte count.  The flat
kernels' row loads are 8 or 16 B/lane over rows of 512 B .. 4 KiB, so the x2 is checked on the same bench with the early
exit disabled, where the dominant kernel requests exactly the algorithmic n*h*W*8 row bytes (plus < 0.5 % hashes /
metadata): FETCH_SIZE * 1024 * 2 / algorithmic is reported as `calibration_check` (1.0 = exact).  The HIBF kernels fetch
32-byte rows (8 B/lane, four lanes per row).  That pattern was calibrated in round 3 (scripts/calib_gather.hip ->
profiles/r03_calib_gather.json: a bare gather with a KNOWN number of row requests): the L2 issues exactly one fabric request
per row request for 32-, 64- and 128-byte rows alike (TCC_EA0_RDREQ = TCC_MISS = requests), each a 128-byte line tallied
at 64 bytes -- so the x2 of the guide holds for it too, and every 32-byte row costs a whole line: traffic = 4 x algorithmic.
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wl = sys.argv[2] if len(sys.argv) > 2 else "flat8g"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{wl}")
dst = os.path.join(ROOT, "profiles")
# kernels whose launches make up one "count/select" step of the workload
HIBF = ["gn_hibf_pack_kernel", "gn_hibf_reg_kernel", "gn_hibf_level_kernel"]
KERNELS = {"hibf64k": HIBF, "hibf64k_top1g": HIBF, "hibf64k_skew": HIBF, "hibf64k_p001": HIBF, "hibf64k_skew_p001": HIBF, "split32k": ["gn_ibf_count_split_kernel"]}.get(wl, ["gn_ibf_count_fast_kernel"])


def find(sub, suffix):
    # gpurun merges every call's files into the same local directory: take the newest one
    hits = glob.glob(os.path.join(src, sub, "**", f"*{suffix}"), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counter_per_step(path, counter, steps):
    """sum of `counter` over every launch of the workload's count kernels, divided by the steps of the run"""
    tot, n, names = 0.0, 0, set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in KERNELS):
                tot += float(row["Counter_Value"])
                n += 1
                names.add(row["Kernel_Name"].split("(")[0])
    return tot / steps, n, sorted(names)


def copy(sub, suffix, name, only_ours=False):
    """only_ours: keep the rows of this library's kernels (gn_*), drop runtime fill/copy and rocprim helpers"""
    p = find(sub, suffix)
    if not p:
        return None
    if not only_ours:
        shutil.copyfile(p, os.path.join(dst, name))
        return p
    with open(p, newline="") as f, open(os.path.join(dst, name), "w", newline="") as g:
        r = csv.reader(f)
        w = csv.writer(g, quoting=csv.QUOTE_NONNUMERIC)
        head = next(r)
        w.writerow(head)
        k = head.index("Kernel_Name")
        for row in r:
            if "gn_" in row[k]:
                w.writerow(row)
    return p


_lines = open(os.path.join(src, "bench.json")).read().splitlines()
_detail = [ln[len("bench_detail: "):] for ln in _lines if ln.startswith("bench_detail: ")]   # (round 5: the full result is the earlier line)
bench = json.loads(_detail[-1] if _detail else [ln for ln in _lines if ln.startswith('{"metric"')][-1])
with open(os.path.join(dst, f"{tag}_bench_{wl}.json"), "w") as f:
    json.dump(bench, f, indent=1)
copy("trace", "kernel_stats.csv", f"{tag}_{wl}_kernel_stats.csv")
copy("trace", "kernel_trace.csv", f"{tag}_{wl}_kernel_trace.csv", only_ours=True)
copy("trace", "agent_info.csv", f"{tag}_agent_info.csv")
p_f = copy("pmc_fetch", "counter_collection.csv", f"{tag}_{wl}_pmc_FETCH_SIZE.csv", only_ours=True)
p_n = copy("pmc_fetch_noee", "counter_collection.csv", f"{tag}_{wl}_pmc_FETCH_SIZE_no_early_exit.csv", only_ours=True)
p_w = copy("pmc_write", "counter_collection.csv", f"{tag}_{wl}_pmc_WRITE_SIZE.csv", only_ours=True)
copy("pmc_sq", "counter_collection.csv", f"{tag}_{wl}_pmc_SQ.csv", only_ours=True)

STEPS = 6  # the counter passes run `bench.py --steps 5 --warmup 1`
rf = bench["roofline"]
algo = rf["algo_bytes_per_launch"] * rf.get("launches_per_step", 1)
fetched = rf.get("fetched_bytes_per_launch", rf["algo_bytes_per_launch"]) * rf.get("launches_per_step", 1)
out = {"workload": wl, "tag": tag, "file": f"profiles/{tag}_{wl}_pmc_FETCH_SIZE.csv", "algo_bytes_per_step": algo, "kernel_fetched_bytes_per_step": fetched,
       "count_ms_bench": bench["config"]["kernel_ms"]["count_select"]}
if p_f:
    raw, n, names = counter_per_step(p_f, "FETCH_SIZE", STEPS)
    out.update({"kernels": names, "launches_seen": n, "FETCH_SIZE_KB_per_step": raw})
    if p_n:
        raw_cal, _, _ = counter_per_step(p_n, "FETCH_SIZE", STEPS)
        out.update({
            "FETCH_SIZE_KB_per_step_no_early_exit": raw_cal,
            "calibration_check": round(raw_cal * 1024.0 * 2.0 / algo, 4),
            "hbm_bytes_per_launch": int(raw * 1024.0 * 2.0),
            "hbm_over_kernel_fetched": round(raw * 1024.0 * 2.0 / fetched, 4),
            "hbm_over_algorithmic": round(raw * 1024.0 * 2.0 / algo, 4),
            "source": f"profiles/{tag}_{wl}_pmc_FETCH_SIZE.csv (normal run) calibrated with "
                      f"profiles/{tag}_{wl}_pmc_FETCH_SIZE_no_early_exit.csv: separate `rocprofv3 --pmc FETCH_SIZE` passes over "
                      "bench.py; FETCH_SIZE is in KiB and, on gfx950, tallies the 128-byte requests of coalesced reads at 64 bytes "
                      "(MI355X_MICROARCH.md HBM section) -> bytes = value * 1024 * 2; the x2 is checked on the run whose kernel "
                      "requests exactly the algorithmic rows (calibration_check)",
        })
    else:
        x1 = raw * 1024.0
        lines = sum(lv["line_bytes"] for lv in rf.get("levels", [])) or None
        out.update({
            "hbm_bytes_per_launch": int(x1 * 2.0),
            "hbm_bytes_if_x1": int(x1), "hbm_bytes_if_x2": int(x1 * 2.0),
            "line_bytes_per_step": lines,
            "fabric_over_line_bytes": round(x1 * 2.0 / lines, 4) if lines else None,
            "fabric_over_algorithmic": round(x1 * 2.0 / algo, 4),
            "source": f"profiles/{tag}_{wl}_pmc_FETCH_SIZE.csv: separate `rocprofv3 --pmc FETCH_SIZE` pass over bench.py, KiB summed over "
                      f"the launches of {names} per step; " + (
                          "512 B .. 4 KiB coalesced rows like the fast kernel's: x2 (calibrated on flat8g in the same round)"
                          if wl == "split32k" else
                          "32-byte rows, 8 B per lane: x2 as calibrated on a bare gather of such rows (profiles/r03_calib_gather.json): one "
                          "128-byte line per row request, tallied at 64 bytes; lines served by the Infinity Cache are counted too"),
        })
if p_w:
    w_raw, _, _ = counter_per_step(p_w, "WRITE_SIZE", STEPS)
    out["WRITE_SIZE_KB_per_step"] = w_raw
with open(os.path.join(dst, f"pmc_fetch_{wl}.json"), "w") as f:
    json.dump(out, f, indent=1)
# the bench line of this round was taken BEFORE the counter passes: stamp the traffic of the same round into its copy so
# that profiles/<tag>_bench_<workload>.json and pmc_fetch_<workload>.json agree
if "hbm_bytes_per_launch" in out:
    bench["roofline"]["traffic"] = out["hbm_bytes_per_launch"]
    bench["roofline"]["traffic_source"] = out.get("source")
    with open(os.path.join(dst, f"{tag}_bench_{wl}.json"), "w") as f:
        json.dump(bench, f, indent=1)
print(json.dumps(out, indent=1))
