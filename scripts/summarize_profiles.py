"""Copies the rocprofv3 summaries of one profiling round (scripts/profile_round.sh) from gpurun_out/ into profiles/
and derives the per-launch HBM traffic that bench.py attaches to its `roofline.traffic`.

FETCH_SIZE is in KiB.  The guide (MI355X_MICROARCH.md, HBM section) says gfx950 tallies 128-byte requests of wide
coalesced reads at 64 bytes (-> x2) and that other access widths must be calibrated on a known byte count.  The
kernel's row loads are 8 B/lane (512 contiguous bytes per wave instruction), so the x2 is checked on the same bench
with the early exit disabled, where the dominant kernel requests exactly the algorithmic n*h*W*8 row bytes (plus
< 0.5 % hashes/metadata): FETCH_SIZE * 1024 * 2 / algorithmic is reported as `calibration_check` (1.0 = exact).
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
KERNEL = "gn_ibf_count_fast_kernel"


def find(sub, suffix):
    # gpurun merges every call's files into the same local directory: take the newest one
    hits = glob.glob(os.path.join(src, sub, "**", f"*{suffix}"), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counter_avg(path, counter):
    path = path if os.path.exists(path) else path
    vals = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    name = max(vals, key=lambda k: len(vals[k]))
    v = vals[name]
    return name, sum(v) / len(v), len(v)


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


bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
wl = bench["config"]["workload"].split(":")[0]
with open(os.path.join(dst, f"{tag}_bench_{wl}.json"), "w") as f:
    json.dump(bench, f, indent=1)
copy("trace", "kernel_stats.csv", f"{tag}_{wl}_kernel_stats.csv")
copy("trace", "kernel_trace.csv", f"{tag}_{wl}_kernel_trace.csv", only_ours=True)
copy("trace", "agent_info.csv", f"{tag}_agent_info.csv")
p_f = copy("pmc_fetch", "counter_collection.csv", f"{tag}_{wl}_pmc_FETCH_SIZE.csv", only_ours=True)
p_n = copy("pmc_fetch_noee", "counter_collection.csv", f"{tag}_{wl}_pmc_FETCH_SIZE_no_early_exit.csv", only_ours=True)
p_w = copy("pmc_write", "counter_collection.csv", f"{tag}_{wl}_pmc_WRITE_SIZE.csv", only_ours=True)
copy("pmc_sq", "counter_collection.csv", f"{tag}_{wl}_pmc_SQ.csv", only_ours=True)

rf = bench["roofline"]
algo = rf["algo_bytes_per_launch"]
fetched = rf.get("fetched_bytes_per_launch", algo)
out = {"workload": wl, "tag": tag, "algo_bytes_per_launch": algo, "kernel_fetched_bytes_per_launch": fetched}
if p_f and p_n:
    name, raw, n = counter_avg(p_f, "FETCH_SIZE")
    _, raw_cal, n_cal = counter_avg(p_n, "FETCH_SIZE")
    factor = 2.0
    out.update({
        "kernel": name,
        "FETCH_SIZE_KB_raw": raw, "launches_averaged": n,
        "FETCH_SIZE_KB_raw_no_early_exit": raw_cal,
        "calibration_check": round(raw_cal * 1024.0 * factor / algo, 4),
        "hbm_bytes_per_launch": int(raw * 1024.0 * factor),
        "hbm_over_kernel_fetched": round(raw * 1024.0 * factor / fetched, 4),
        "hbm_over_algorithmic": round(raw * 1024.0 * factor / algo, 4),
        "source": f"profiles/{tag}_{wl}_pmc_FETCH_SIZE.csv (normal run) calibrated with "
                  f"profiles/{tag}_{wl}_pmc_FETCH_SIZE_no_early_exit.csv: separate `rocprofv3 --pmc FETCH_SIZE` passes over "
                  "bench.py; FETCH_SIZE is in KiB and, on gfx950, tallies the 128-byte requests of coalesced reads at 64 bytes "
                  "(MI355X_MICROARCH.md HBM section) -> bytes = value * 1024 * 2; the x2 is checked on the run whose kernel "
                  "requests exactly the algorithmic rows (calibration_check)",
    })
if p_w:
    _, w_raw, _ = counter_avg(p_w, "WRITE_SIZE")
    out["WRITE_SIZE_KB_raw"] = w_raw
with open(os.path.join(dst, f"pmc_fetch_{wl}.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
