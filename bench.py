#!/usr/bin/env python3
"""bench.py -- ganon read classification on MI355X: Mreads/s classified + IBF-lookup GB/s vs the HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (minimiser kernels -> IBF/HIBF count+select kernels -> match grouping, plus
the sparse-match exchange for a partitioned filter) over one batch of synthetic 150 bp reads that is already
resident in HBM; the filter is resident too.  Workloads (BASELINE.json `configs`):

  flat8g    configs[1]  8 GiB flat IBF, 4096 bins, h=4, 10 M reads                      <- the default / headline
  hibf64k   configs[2]  2-level HIBF, 65 536 user bins (256 x 256), 10 M reads
  flat128g  configs[3]  128 GiB flat IBF (32 768 bins, 4 KiB rows), 12.5 M pairs 2x150 = the per-GPU shard of the
                        100 M-pair job; with N ranks every rank holds a replica and its own shard ("weak")
  slice1t   configs[4]  one rank's 128 GiB column slice (32 768 of 262 144 bins) of a 1 TiB filter; every rank sees
                        every pair, sparse matches go to the read's owner over RCCL (world 1 on one GPU)

The 128 GiB filters are generated on the device (gn_filter_fill_random); nothing that large exists on the host.
Output (rank 0): everything measured goes to `bench_detail.json` (next to this file, and under gpurun_out/ when that
exists) and to ONE EARLIER stdout line `bench_detail: {...}`; the LAST stdout line is the driver's record -- one
compact JSON object (a few KB, never more than 8 KB: tests/test_bench_line.py) holding exactly
  metric value unit n_gpus steps warmup ms_per_step higher_is_better scaling vs_baseline dtype data
  config        {workload, reads_per_gpu, parallelism, oracle_mismatching_reads, <= 20 flat scalars}
  roofline      dominant kernel, three named fractions of the 8000 GB/s HBM peak (hipEvent times on the library's stream):
                  frac           SURVEY 8(d) algorithmic bytes (n*h*W*8 per read) / kernel time, measured on the
                                 instantiation that FETCHES EVERY ALGORITHMIC ROW (exact early exit ablated) = `achieved`
                  frac_fetched   the product kernel (early exit on): row bytes it really requested / its time
                  algo_over_peak the product kernel: algorithmic bytes / its time; may exceed 1 because rows the exact
                                 early exit skips are never fetched and cost nothing
                traffic = HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE pass (profiles/pmc_fetch_*.json)
  cpu_baseline  {value, unit, cores, kind, sample}: the real ganon-classify when one that is not ours is on PATH
                (kind "reference"), else the CPU oracle (kind "port"), on a bounded sample of the same reads
The detail file holds, besides the full headline objects:
  variants         the same resident batch with the early exit disabled and at the binary's default --rel-cutoff 0.2
  other_workloads  the other BASELINE configs at full size: at N=1 each in a child process (hibf64k, its variant with an
                   HBM-resident top level, flat128g, slice1t); at N>1 in this job, one after the other: flat128g (configs[3]:
                   12.5 M pairs PER RANK against a replica, weak) and slice1t (configs[4]: rank r holds column slice r, every
                   rank classifies all pairs, sparse matches go to the pair's owner over RCCL) -- so a scaling run yields the
                   points BASELINE.json quotes its scaling target on
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    "flat8g": dict(kind="flat", bins=4096, rows=1 << 24, h=4, reads=10_000_000, paired=False, config=1),
    "flat1g": dict(kind="flat", bins=4096, rows=1 << 21, h=4, reads=2_000_000, paired=False, config=None),
    "tiny": dict(kind="flat", bins=4096, rows=1 << 14, h=4, reads=100_000, paired=False, config=None),
    "flat32k": dict(kind="flat", bins=32768, rows=1 << 21, h=4, reads=2_000_000, paired=False, config=None),
    # split-bin map (what ganon-build makes of targets larger than max_hashes_bin): two technical bins per target
    "split32k": dict(kind="flat", bins=32768, rows=1 << 21, h=4, reads=2_000_000, paired=False, config=None, bins_per_target=2),
    # long reads (5 kbp, ~710 minimisers each): every read goes through the generic count kernel (LDS counters), none through the fast one
    "long8g": dict(kind="flat", bins=4096, rows=1 << 24, h=4, reads=300_000, paired=False, config=None, read_len=5000),
    "long32k": dict(kind="flat", bins=32768, rows=1 << 21, h=4, reads=300_000, paired=False, config=None, read_len=5000),
    "hibf64k": dict(kind="hibf", user_bins=65536, tmax=256, rows=1 << 20, h=3, reads=10_000_000, paired=False, config=2),
    # the same tree with a top-level IBF of 1 GiB (2^25 rows of 32 bytes): level 0 no longer fits the 256 MiB Infinity Cache
    "hibf64k_top1g": dict(kind="hibf", user_bins=65536, tmax=256, rows=1 << 20, rows_top=1 << 25, h=3, reads=10_000_000, paired=False, config=2),
    # raptor-like layout: log-normal user-bin sizes -> split user bins at the top, merged bins of different cardinality, children of 2 ... 1024
    # technical bins with different numbers of rows, three levels; Bernoulli(3/8) bits (p^h = 0.053); a tenth of the reads descends into two children
    "hibf64k_skew": dict(kind="hibf", skew=True, user_bins=65536, h=3, reads=10_000_000, paired=False, config=2),
    # The HIBF at the reference's own defaults: `ganon build --filter-type hibf` passes --max-fp 0.001 --hash-functions 4 to raptor
    # (/root/reference/src/ganon/config.py:140-143,1258-1260, build_update.py:487-489): h = 4, bits Bernoulli(3/16) (0.1875^4 = 0.0012),
    # the uniform 256 x 256 tree and the raptor-like skewed one.  The h=3 workloads above are 50 x that false-positive rate.
    "hibf64k_p001": dict(kind="hibf", user_bins=65536, tmax=256, rows=1 << 20, h=4, fill="3/16", reads=10_000_000, paired=False, config=2),
    "hibf64k_skew_p001": dict(kind="hibf", skew=True, user_bins=65536, h=4, fill="3/16", reads=10_000_000, paired=False, config=2),
    "hibf_p001_tiny": dict(kind="hibf", user_bins=4096, tmax=64, rows=1 << 12, h=4, fill="3/16", reads=100_000, paired=False, config=None),
    "hibf_skew_p001_tiny": dict(kind="hibf", skew=True, user_bins=8192, h=4, fill="3/16", reads=100_000, paired=False, config=None, rows_scale=0.01),
    "hibf_skew_tiny": dict(kind="hibf", skew=True, user_bins=8192, h=3, reads=100_000, paired=False, config=None, rows_scale=0.01),
    "hibf_tiny": dict(kind="hibf", user_bins=4096, tmax=64, rows=1 << 12, h=3, reads=100_000, paired=False, config=None),
    "flat128g": dict(kind="flat", bins=32768, rows=1 << 25, h=4, reads=12_500_000, paired=True, config=3),
    "slice1t": dict(kind="slice", bins=32768, slices=8, rows=1 << 25, h=4, reads=12_500_000, paired=True, config=4),
    "slice_tiny": dict(kind="slice", bins=4096, slices=8, rows=1 << 14, h=4, reads=100_000, paired=True, config=None),
}
EXTRA_WORKLOADS = ["hibf64k", "hibf64k_p001", "hibf64k_skew", "hibf64k_skew_p001", "flat128g", "slice1t"]   # N = 1: child processes
EXTRA_WORKLOADS_MULTI = ["flat128g", "slice1t"]                           # N > 1: in this job (BASELINE's scaling configs)

# What a random gather of 128-byte lines reaches on MI355X by residency of the table (scripts/calib_gather.hip, measured:
# profiles/r03_calib_gather.json): every row narrower than a line still moves the whole line (TCC_EA0_RDREQ = 1 per row
# request for 32-, 64- and 128-byte rows alike; FETCH_SIZE tallies a line at 64 B, i.e. the guide's x2 holds here too).
GATHER_ROOF_GBS = [(64 << 20, 8394.0, "infinity cache (table <= 64 MiB)"), (256 << 20, 7267.0, "infinity cache (table <= 256 MiB)"),
                   (2 << 30, 7158.0, "HBM behind a partly caching MALL (table <= 2 GiB)"), (1 << 62, 6407.0, "HBM")]


def gather_roof(table_bytes: int):
    for lim, gbs, where in GATHER_ROOF_GBS:
        if table_bytes <= lim:
            return gbs, where
    return GATHER_ROOF_GBS[-1][1:]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def run_extra(name: str, timeout: int):
    """another BASELINE config at full size in a child process (a crash or a timeout there cannot lose the main line)"""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-extra", "--no-variants"]
    t0 = time.time()
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
        line = [ln[len("bench_detail: "):] for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("bench_detail: ")]
        if p.returncode == 0 and line:
            r = json.loads(line[-1])
            return slim(name, r, time.time() - t0)
        return {"workload": name, "error": f"rc {p.returncode}: " + p.stderr.decode(errors="replace")[-400:]}
    except subprocess.TimeoutExpired:
        return {"workload": name, "error": f"timeout after {timeout}s"}
    except Exception as e:  # noqa: BLE001 -- the extras never take the main line down
        return {"workload": name, "error": repr(e)}


def run_e2e(budget: int, devices: str = "", only: str = "", reads: int = 0):
    """the product binary end to end on generated files that classify (bench_e2e.py), in a child process.  `devices` = the binary's
    --device list: an N-GPU job ends with `ganon-classify --device all` on plain FASTQ and .fq.gz, so that a scaling curve shows the
    BINARY (one reader, N workers, ordered post stage) and not only resident batches"""
    cmd = [sys.executable, os.path.join(ROOT, "bench_e2e.py"), "--budget", str(budget)]
    if devices:
        cmd += ["--devices", devices]
    if only:
        cmd += ["--only", only]
    if reads:
        cmd += ["--reads", str(reads)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget + 240, cwd=ROOT)
        line = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode == 0 and line:
            return json.loads(line[-1])
        return {"error": f"rc {p.returncode}: " + p.stderr.decode(errors="replace")[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": f"timeout after {budget + 240}s"}
    except Exception as e:  # noqa: BLE001 -- never lose the kernel line over the end-to-end leg
        return {"error": repr(e)}


LINE_CAP = 8192      # the driver keeps ~9 KB of stdout tail: a longer last line is cut and cannot be parsed (BENCH_r04.json)
LINE_TARGET = 4096
MAX_EXTRA = 44       # flat scalars in `config` after the four named ones


def _short(text, n: int) -> str:
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def compact_line(result: dict) -> dict:
    """The driver's record: the headline scalars, `config` with <= 20 flat scalars after the four named ones, `roofline` and
    `cpu_baseline` as flat objects of scalars.  Everything else lives in bench_detail.json.  Pure (tests/test_bench_line.py
    feeds it the 22.7 KB line of round 4)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: result.get(k) for k in keep}
    c = result.get("config") or {}
    chk = c.get("oracle_spot_check") or {}
    cfg = {"workload": _short(c.get("workload_short") or c.get("workload", "?"), 240),
           "reads_per_gpu": c.get("reads_per_gpu"), "parallelism": _short(c.get("parallelism", "?"), 80),
           "oracle_mismatching_reads": c.get("oracle_mismatching_reads", chk.get("mismatching_reads"))}
    extra = []   # (name, scalar), in order of importance; cut at MAX_EXTRA
    extra.append(("oracle_reads_checked", chk.get("reads_checked")))
    for k in ("mean_minimisers_per_read", "match_checksum_all_ranks"):
        extra.append((k, c.get(k)))
    km = c.get("kernel_ms") or {}
    extra.append(("minimiser_ms", km.get("minimiser")))
    extra.append(("count_select_ms", km.get("count_select")))
    hw = str(c.get("workload_short") or c.get("workload", "")).split(" ")[0] or "headline"
    va = result.get("variants") or {}
    # the headline batch at the BINARY's own default --rel-cutoff 0.2 (Config.hpp:32), plain and under the wrapper's filter rules
    extra.append((f"{hw}_cutoff0.2_mreads_s", (va.get("rel_cutoff_0.2") or {}).get("mreads_per_s")))
    extra.append((f"{hw}_cutoff0.2_count_select_ms", (va.get("rel_cutoff_0.2") or {}).get("count_select_ms")))
    extra.append((f"{hw}_wrapper_defaults_mreads_s", (va.get("wrapper_defaults_device_filter_matches") or {}).get("mreads_per_s")))
    for o in result.get("other_workloads") or []:
        w = o.get("workload", "?")
        if "error" in o:
            extra.append((f"{w}_error", _short(o["error"], 80)))
            continue
        extra.append((f"{w}_{str(o.get('unit', 'Mreads/s')).replace('/s', '_s').lower()}", o.get("value")))
        low = (o.get("variants") or {}).get("low_cutoff_device_filter_matches")
        if low:   # the binary's --rel-cutoff 0.2 under the wrapper's filter rules
            extra.append((f"{w}_cutoff0.2_mreads_s", low.get("mreads_per_s")))
        orf = o.get("roofline") or {}
        # SURVEY 8(d) fraction of every workload: algorithmic bytes / count kernels' time / 8000 GB/s; HIBF: plus the levels' LINE rates
        # over the calibrated gather roofs ("0.99/0.84/0.67", top level first) and traffic / algorithmic bytes when a PMC pass exists
        extra.append((f"{w}_frac", orf.get("frac")))
        if orf.get("levels"):
            extra.append((f"{w}_level_line_fracs", "/".join(f"{lv.get('frac_of_gather_roof', 0):.2f}" for lv in orf["levels"])))
        if orf.get("traffic") and orf.get("algo_bytes_per_step"):
            extra.append((f"{w}_traffic_over_algo", round(orf["traffic"] / orf["algo_bytes_per_step"], 3)))
        ex = (o.get("ranks") or {}).get("exchange_ms_max")
        if ex is not None:
            extra.append((f"{w}_exchange_ms", ex))
    bad_extra = sum(int(((o.get("config") or {}).get("oracle_spot_check") or {}).get("mismatching_reads") or 0)
                    for o in result.get("other_workloads") or [] if "error" not in o)
    if result.get("other_workloads"):
        extra.append(("other_workloads_mismatching_reads", bad_extra))
    e2e = result.get("e2e") or {}
    if "error" in e2e:
        extra.append(("e2e_error", _short(e2e["error"], 80)))
    if e2e.get("devices"):
        extra.append(("e2e_devices", e2e["devices"]))
    for name, r in (e2e.get("inputs") or {}).items():
        if "error" in r:
            extra.append((f"e2e_{name}_error", _short(r["error"], 80)))
        else:
            extra.append((f"e2e_{name}_{'mpairs_s' if name.startswith('paired') else 'mreads_s'}_median", (r.get("rate") or {}).get("median")))
    for k, v in [(k, v) for k, v in extra if v is not None][:MAX_EXTRA]:
        cfg[k] = v
    line["config"] = cfg

    r = result.get("roofline") or {}
    roof = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_fetched", "algo_over_peak", "traffic",
                                  "achieved_fetched", "avg_launch_ms", "avg_launch_ms_every_row", "launches_per_step",
                                  "algo_bytes_per_launch", "fetched_bytes_per_launch")}
    roof["frac_measured_on"] = _short(r.get("frac_measured_on", ""), 120)
    roof["note"] = _short(r.get("note", ""), 200)
    if r.get("traffic") is not None and r.get("fetched_bytes_per_launch"):
        roof["traffic_over_fetched"] = round(r["traffic"] / r["fetched_bytes_per_launch"], 4)
    # `traffic` is NOT measured in this run: counters cannot share a run with timing, so it is read from the committed summary of a
    # separate `rocprofv3 --pmc FETCH_SIZE` pass over the same batch (same algorithmic bytes, checked) -- say so, and which file
    roof["traffic_measured_in_this_run"] = False
    roof["traffic_from"] = ("none" if r.get("traffic") is None else
                            _short(r.get("traffic_from") or str(r.get("traffic_source", "?")).split(" ")[0].rstrip(":"), 100))
    line["roofline"] = roof
    rk = result.get("ranks")
    if isinstance(rk, dict):   # the proof that n_gpus ranks on n_gpus devices measured the line (flat scalars)
        line["ranks"] = {k: (_short(v, 260) if isinstance(v, str) else v) for k, v in rk.items() if not isinstance(v, (dict, list))}

    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample", ""), 300)}
        for k in ("port_value", "agrees_with_ours"):
            if k in cb:
                line["cpu_baseline"][k] = cb[k]
    else:
        line["cpu_baseline"] = None
    line["detail"] = "bench_detail.json"
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_TARGET:      # never expected; shed prose before scalars
        line["roofline"].pop("note", None)
        line["roofline"].pop("frac_measured_on", None)
        (line.get("ranks") or {}).pop("ms_per_step_by_rank", None)
        if line["cpu_baseline"]:
            line["cpu_baseline"]["sample"] = _short(line["cpu_baseline"]["sample"], 120)
        line["config"]["workload"] = _short(line["config"]["workload"], 120)
        named = ("workload", "reads_per_gpu", "parallelism", "oracle_mismatching_reads")
        while len(json.dumps(line, separators=(",", ":"))) > LINE_TARGET - 64:   # then the least important scalars, last first
            last = [k for k in line["config"] if k not in named]
            if not last:
                break
            line["config"].pop(last[-1])
            line["config_scalars_dropped"] = line.get("config_scalars_dropped", 0) + 1
    return line


def emit(result: dict, write_files: bool = True) -> None:
    """detail file + an earlier stdout line, then the compact record as the LAST stdout line"""
    detail = json.dumps(result)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if write_files else ():
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError as e:
                log("bench.py: could not write", os.path.join(d, "bench_detail.json"), repr(e))
    sys.stdout.write("bench_detail: " + detail + "\n")
    text = json.dumps(compact_line(result), separators=(",", ":"))
    assert len(text) < LINE_CAP, f"driver line is {len(text)} bytes"
    sys.stdout.write(text + "\n")
    sys.stdout.flush()


def slim(name: str, r: dict, wall: float) -> dict:
    out = {"workload": name, "value": r["value"], "unit": r["unit"], "n_gpus": r["n_gpus"], "scaling": r["scaling"],
           "ms_per_step": r["ms_per_step"], "steps": r["steps"], "config": r["config"], "roofline": r["roofline"],
           "wall_s": round(wall, 1)}
    if r.get("ranks"):
        out["ranks"] = r["ranks"]
    if r.get("variants"):
        out["variants"] = r["variants"]
    return out


def ALLOW_SHARED_GPU() -> bool:
    return os.environ.get("GANON_BENCH_ALLOW_SHARED_GPU", "") == "1"


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher: become the launcher.  N ranks of this script, one per GPU, with the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR/PORT on 127.0.0.1);
    rank 0 keeps this process's stdout (its last line is the record), the other ranks' stdout goes to stderr.  A rank that fails
    takes the job down (the others are sent SIGTERM by PID) and its exit code is returned: no line is better than a line from fewer
    ranks.  The N-worker structure mirrors the reference's N classify threads over additive totals (GanonClassify.cpp:1579-1597,
    :475-490)."""
    import socket
    try:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:  # noqa: BLE001
        have = 0
    if have < n and not ALLOW_SHARED_GPU():
        log(f"bench.py: --gpus {n} but {have} GPU(s) visible -- refusing to run {n} ranks on fewer GPUs "
            f"(GANON_BENCH_ALLOW_SHARED_GPU=1 permits it for a dry run)")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GANON_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=None if r == 0 else sys.stderr))
    log(f"bench.py: started {n} ranks myself (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                log(f"bench.py: rank {r} (pid {procs[r].pid}) left with code {c}: stopping the other ranks")
                for o in sorted(live):
                    procs[o].terminate()
        time.sleep(0.2)
    return rc


def rank_proof(torch, gdist, world: int, dev_index: int, red_dev: str, backend: str) -> dict:
    """What shows that a line with n_gpus = N was measured by N ranks on N GPUs: the number of ranks that answered an all-reduce
    of ones over the job's backend (nccl = RCCL), and every rank's device identity (PCI domain:bus:device + uuid)."""
    p = torch.cuda.get_device_properties(dev_index)
    try:
        ident = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except Exception:  # noqa: BLE001
        ident = f"cuda{dev_index}"
    uuid = str(getattr(p, "uuid", "") or "")
    ident += "/" + uuid.replace("-", "")[:12]
    idents = gdist.all_gather_text(ident, device=red_dev)
    seen = gdist.sum_over_ranks(1, device=red_dev)
    return {"ranks_seen": int(seen), "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else "none (one rank)",
            "launcher": os.environ.get("GANON_BENCH_LAUNCHER") or ("torchrun/env" if "WORLD_SIZE" in os.environ else "none"),
            "devices": ",".join(sorted(idents)), "distinct_devices": len(set(idents)),
            "shared_gpu_dry_run": bool(len(set(idents)) < world)}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: $WORLD_SIZE, else 1).  N > 1 without a launcher "
                                                          "(no $WORLD_SIZE): this process starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("GANON_BENCH_WORKLOAD", "flat8g"), choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=0, help="override reads (pairs) per GPU")
    ap.add_argument("--rows", type=int, default=0, help="override filter rows (dry runs)")
    ap.add_argument("--rel-cutoff", type=float, default=0.75, help="0.75 = `ganon classify` default (src/ganon/config.py:597)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="do not run the other BASELINE configs after the headline")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-every-row", action="store_true", help="skip the 3-step measurement with the early exit ablated (profile passes: every launch of the "
                                                                "count kernel is then one of the timed steps; roofline.frac falls back to the product kernel)")
    ap.add_argument("--extra-timeout", type=int, default=420)
    ap.add_argument("--no-e2e", action="store_true", help="do not run the product binary end to end after the kernels (bench_e2e.py)")
    ap.add_argument("--e2e-budget", type=int, default=200, help="seconds the end-to-end leg may take")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU baseline sample (0 = auto, ~15 s)")
    ap.add_argument("--check", type=int, default=2000, help="reads re-checked against the oracle (rank 0)")
    ap.add_argument("--emit-from", default="", help="dry run without a GPU: read a full result (a bench_detail.json) and print it the "
                                                     "way a real run does (tests/test_bench_line.py)")
    args = ap.parse_args()
    if args.emit_from:
        with open(args.emit_from) as f:
            emit(json.loads(f.read()), write_files=False)
        return 0

    # ---- how many ranks, and who starts them.  `--gpus N` is a REQUEST for N ranks on N distinct GPUs and is checked against
    # what the job really is; it is never just a label (a bare `python bench.py --gpus 8` measured one GPU).
    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus or 1) > 1:
        return self_launch(args.gpus, sys.argv[1:])
    import torch

    import ganon_amd
    import bench_workload as bw
    from ganon_amd import dist as gdist
    from ganon_amd import partition as gp

    rank, local_rank, world = gdist.env_rank_world()
    if args.gpus is not None and args.gpus != world:
        log(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks -- refusing to print a line whose n_gpus "
            f"would not be what was asked for")
        return 2
    if not torch.cuda.is_available():
        log("bench.py: no GPU visible -- the hot path has no CPU fallback")
        return 2
    n_visible = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if n_visible < local_world and not ALLOW_SHARED_GPU():
        log(f"bench.py: {local_world} ranks on this node but only {n_visible} GPU(s) visible -- ranks would share a GPU and the line would "
            f"not be a {world}-GPU measurement.  (GANON_BENCH_ALLOW_SHARED_GPU=1 permits it for a dry run; the line then says so.)")
        return 2
    dev_index = local_rank % n_visible   # == local_rank on a real N-GPU node
    torch.cuda.set_device(dev_index)
    # nccl == RCCL.  RCCL refuses two ranks on one device ("duplicate GPU"), so the shared-GPU dry run -- and only it -- talks gloo
    dist_backend = os.environ.get("GANON_BENCH_DIST", "gloo" if n_visible < local_world else "nccl")
    red_dev = "cuda" if dist_backend == "nccl" else "cpu"
    if world > 1:
        gdist.init(dist_backend, torch.device("cuda", dev_index))
    proof = rank_proof(torch, gdist, world, dev_index, red_dev, dist_backend)
    if proof["ranks_seen"] != world or (proof["distinct_devices"] < min(world, local_world) and not ALLOW_SHARED_GPU()):
        log(f"bench.py: the process group does not hold {world} ranks on distinct GPUs: {proof}")
        return 2

    def run_one(name: str, headline: bool) -> dict:
        spec = dict(WORKLOADS[name])
        kind = spec["kind"]
        steps, warmup = (args.steps, args.warmup) if headline else (3, 1)
        if kind == "slice":  # the partitioned filter exchanges matches over RCCL even at world size 1
            gdist.init(dist_backend, torch.device("cuda", dev_index))

        n_reads = (args.reads if headline else 0) or spec["reads"]
        rows = (args.rows if headline else 0) or spec.get("rows", 0)
        paired = spec["paired"]
        t0 = time.time()
        part = None
        if kind == "hibf" and spec.get("skew"):
            wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, name, spec["user_bins"], spec["h"], n_reads, rel_cutoff=args.rel_cutoff, seed=42,
                                                        shard=rank, device=dev_index, rows_scale=spec.get("rows_scale", 1.0),
                                                        fill=ganon_amd.FILL_3_OF_16 if spec.get("fill") == "3/16" else 0)
            off2 = None
            lay = wl.layout
            desc = (f"{lay['depth']}-level HIBF {wl.filter_bytes / 2**30:.2f} GiB replicated per GPU, {spec['user_bins']} user bins of log-normal size: top IBF "
                    f"{lay['top_bins']} bins ({lay['top_split_user_bins']} user bins split over {lay['top_split_technical_bins']}, the rest merged), "
                    f"{lay['ibfs'] - 1} lower IBFs of {lay['child_bins_min']}..{lay['child_bins_max']} bins (median {lay['child_bins_median']}) and "
                    f"{lay['rows_min']}..{lay['rows_max']} rows, h={spec['h']}, {lay['fill']} bits, {lay['genomes_in_two_user_bins']} genomes in two user bins")
            kernel_name = "gn_hibf_pack_kernel"   # 69 % of the device time of the skewed tree (profiles/r05_hibf64k_skew_kernel_stats.csv)
            row_bytes = ((lay["top_bins"] + 63) >> 6) * 8
        elif kind == "hibf":
            rows_top = spec.get("rows_top", rows) if not (headline and args.rows) else rows
            wl, flt = bw.make_hibf_device_workload(ganon_amd, name, spec["user_bins"], spec["tmax"], rows_top, rows, spec["h"],
                                                   n_reads, rel_cutoff=args.rel_cutoff, seed=42, shard=rank, device=dev_index,
                                                   fill=ganon_amd.FILL_3_OF_16 if spec.get("fill") == "3/16" else 1)
            off2 = None
            desc = (f"2-level HIBF {wl.filter_bytes / 2**30:.2f} GiB replicated per GPU: top IBF {spec['tmax']} merged bins -> "
                    f"{spec['tmax']} child IBFs x {spec['user_bins'] // spec['tmax']} user bins = {spec['user_bins']} user bins, "
                    f"S={rows_top} rows (top, {rows_top * 32 / 2**20:.0f} MiB) / {rows} rows (children), h={spec['h']}")
            kernel_name = "gn_hibf_pack_kernel"
            row_bytes = ((spec["tmax"] + 63) >> 6) * 8
        else:
            slices = spec.get("slices", 1)
            W_local = (spec["bins"] + 63) >> 6
            sl_idx = rank % slices if kind == "slice" else 0
            read_len = spec.get("read_len", 150)
            wl = bw.make_device_flat_workload(name, spec["bins"], rows, spec["h"], n_reads, paired, rel_cutoff=args.rel_cutoff,
                                              seed=42, shard=0 if kind == "slice" else rank, word_lo=sl_idx * W_local,
                                              row_words_total=W_local * slices, read_len=read_len, genome_len=max(3000, 4 * read_len))
            bpt = spec.get("bins_per_target", 1)
            if bpt > 1:
                flt, n_planted = bw.device_filter(ganon_amd, wl, dev_index, (np.arange(spec["bins"], dtype=np.uint32) // bpt).astype(np.uint32),
                                                  spec["bins"] // bpt)
            else:
                flt, n_planted = bw.device_filter(ganon_amd, wl, dev_index)
            off2 = wl.off2
            row_bytes = W_local * 8
            if kind == "slice":
                sl = gp.Slice(rank, sl_idx * W_local, (sl_idx + 1) * W_local, spec["bins"], np.arange(spec["bins"], dtype=np.uint32),
                              (np.arange(spec["bins"], dtype=np.uint32) + np.uint32(sl_idx * spec["bins"])))
                part = gp.PartitionedIbf(sl, rank, world, gp.HipLocalFilter(flt, dev_index), comm_device=red_dev)
                desc = (f"column slice {sl_idx} of {slices} ({spec['bins']} of {spec['bins'] * slices} technical bins, "
                        f"{wl.filter_bytes / 2**30:.2f} GiB of a {wl.filter_bytes * slices / 2**40:.2f} TiB flat IBF), W_local={W_local} "
                        f"({row_bytes} B rows), S={rows} rows, h={spec['h']}; every rank sees every read, sparse matches go to the "
                        f"read's owner with one all-to-all over RCCL")
            else:
                desc = (f"flat IBF {wl.filter_bytes / 2**30:.2f} GiB replicated per GPU, {wl.bins} technical bins (W={W_local}, "
                        f"{row_bytes} B rows), S={rows} rows, h={spec['h']}" + (f", {bpt} bins per target" if bpt > 1 else ""))
            kernel_name = "gn_ibf_count_split_kernel" if bpt > 1 else "gn_ibf_count_fast_kernel"
            if read_len > 1000:
                kernel_name = "gn_ibf_count_kernel"  # more than 127 minimisers per read: the generic kernel takes them all
        unit_name = "pairs (2x150 bp)" if paired else f"reads ({spec.get('read_len', 150)} bp)"
        log(f"[rank {rank}] workload {name}: filter {wl.filter_bytes / 2**30:.2f} GiB filled on the device, {n_reads} "
            f"{unit_name}, set up in {time.time() - t0:.1f}s")

        if part is None:
            # (matches per read the stream's buffers are made for; low-cutoff probes ask for more up front: $GANON_BENCH_MATCHES_PER_READ)
            st = ganon_amd.HipStream(flt, n_reads, wl.bases.size, max_matches=n_reads * int(os.environ.get("GANON_BENCH_MATCHES_PER_READ", "2")))
            st.upload(wl.bases, wl.off, off2)
            st.sync()

            def step(cutoff):
                st.classify(wl.k, wl.w, cutoff)
                st.sync()
        else:
            owned = {}

            def step(cutoff):  # classify against the slice, exchange, merge on the owner; the result stays in HBM
                owned["out"] = part.classify(wl.bases, wl.off, off2, wl.k, wl.w, cutoff, fetch=False)
            step(args.rel_cutoff)  # creates the stream and uploads the batch (it stays resident)
            st = part.local.st

        def barrier():
            torch.cuda.synchronize()
            gdist.barrier()
            torch.cuda.synchronize()

        def timed(cutoff, steps, warmup):
            for _ in range(warmup):
                step(cutoff)
            cms, mms, tms = [], [], []
            barrier()
            t_begin = time.perf_counter()
            for _ in range(steps):
                step(cutoff)
                tm = st.timings()           # hipEvent durations on the stream the kernels ran on
                cms.append(tm["ms_count"])
                mms.append(tm["ms_minimiser"])
                tms.append(tm["ms_total"])
            barrier()
            return time.perf_counter() - t_begin, cms, mms, tms, st.timings()

        def rates(tm, cms):
            """one measurement of the count+select kernel(s): hipEvent time per step, row bytes requested, algorithmic bytes"""
            avg = float(np.mean(cms))
            nl = max(1, tm["n_count_launches"])
            return {"ms": avg, "launches": int(nl), "fetched_bytes": int(tm["fetched_bytes"]), "algo_bytes": int(tm["algo_bytes"]),
                    "fetched_gbs": tm["fetched_bytes"] / (avg * 1e-3) / 1e9, "algo_gbs": tm["algo_bytes"] / (avg * 1e-3) / 1e9}

        def variant_of(rt, tms):
            return {"achieved": round(rt["fetched_gbs"], 1), "frac": round(rt["fetched_gbs"] / HBM_PEAK_GBS, 4),
                    "effective_gbs": round(rt["algo_gbs"], 1), "avg_launch_ms": round(rt["ms"] / rt["launches"], 4),
                    "ms_per_step": round(float(np.mean(tms)), 3), "mreads_per_s": round(n_reads / float(np.mean(tms)) / 1e3, 2)}

        elapsed, count_ms, mini_ms, total_ms, tm = timed(args.rel_cutoff, steps, warmup)
        per_rank_ms = [e * 1e3 / max(1, steps) for e in gdist.all_gather_float(elapsed, device=red_dev)]   # every rank's own clock
        elapsed = gdist.max_over_ranks(elapsed, device=red_dev)       # slowest rank defines the step time
        exchange = None
        if part is not None:   # the partitioned filter's exchange step (variable all-to-all over RCCL + merge on the owner), per step
            ex = [gdist.all_gather_float(float(np.mean(v[-steps:])), device=red_dev) for v in (part.exchange_ms, part.merge_ms)]
            exchange = {"exchange_ms_max": round(max(ex[0]), 3), "exchange_ms_min": round(min(ex[0]), 3), "merge_ms_max": round(max(ex[1]), 3)}
        ee = rates(tm, count_ms)                                      # the product kernel: exact early exit on
        # SURVEY 8(d)'s fraction belongs to the instantiation that fetches every algorithmic row: the same resident batch
        # with the early exit ablated (3 steps, outside `value`).  Split-bin and HIBF kernels have no early exit.
        full, full_tms = ee, total_ms
        if kind in ("flat", "slice") and spec.get("bins_per_target", 1) == 1 and not args.no_every_row:
            with ganon_amd.ablate("early_exit"):
                _, cms0, _, full_tms, tm0 = timed(args.rel_cutoff, 3, 1)
            full = rates(tm0, cms0)
            step(args.rel_cutoff)                                     # the product result is what the checks below fetch
        if kind == "slice":
            total_reads = n_reads                                      # every rank classifies the same reads against its columns
        else:
            total_reads = gdist.sum_over_ranks(n_reads, device=red_dev)    # whole-job reads per step
        if part is None:
            nh, status, mo, matches = st.fetch()
        else:
            lo, hi = owned["out"][:2]
            nh, status = st.fetch_read_info()
            matches = part.fetch_owned()
            mo = np.searchsorted(matches["read"], np.arange(n_reads + 1)).astype(np.uint64)
        n_class = int(np.count_nonzero(np.diff(mo.astype(np.int64))))
        ms_per_step = elapsed * 1e3 / max(1, steps)
        value = total_reads / (elapsed / max(1, steps)) / 1e6  # Mreads/s (M pairs/s for paired workloads), whole job

        roof = {"bound": "hbm", "kernel": kernel_name,
                "achieved": round(full["algo_gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(full["algo_gbs"] / HBM_PEAK_GBS, 4),
                "frac_fetched": round(ee["fetched_gbs"] / HBM_PEAK_GBS, 4),
                "algo_over_peak": round(ee["algo_gbs"] / HBM_PEAK_GBS, 4),
                "achieved_fetched": round(ee["fetched_gbs"], 1),
                "avg_launch_ms": round(ee["ms"] / ee["launches"], 4),
                "avg_launch_ms_every_row": round(full["ms"] / full["launches"], 4),
                "launches_per_step": ee["launches"],
                "algo_bytes_per_step": ee["algo_bytes"],
                "algo_bytes_per_launch": ee["algo_bytes"] // ee["launches"],
                "fetched_bytes_per_launch": ee["fetched_bytes"] // ee["launches"],
                "frac_measured_on": ("early exit ablated (every algorithmic row fetched), 3 steps on the same resident batch" if full is not ee
                                     else ("the product kernel (--no-every-row)" if args.no_every_row else "the product kernel (it has no early exit)")),
                "note": "algo_over_peak may exceed 1: rows the exact early exit skips are never fetched"}
        if kind == "hibf":
            # The roofline per tree level: a level's rows are gathered from the IBFs at that depth, and whether those sit in
            # the 256 MiB Infinity Cache or in HBM decides the roof.  A row narrower than a 128-byte line still moves the line
            # (calibrated: one fabric request per row request for 32-, 64- and 128-byte rows), so the physical rate is the
            # LINE rate; the roof is what a bare random gather of lines reaches at that residency (scripts/calib_gather.hip).
            levels = []
            for li, lv in enumerate(st.hibf_levels()):
                rb = max(1, lv["row_bytes"])
                line = lv["line_bytes"]   # the level's row requests in the 128-byte lines they occupy (counted by the kernels: widths differ inside a level)
                roof_gbs, where = gather_roof(lv["table_bytes"])
                sec = max(lv["ms"], 1e-6) * 1e-3
                levels.append({"level": li, "ms": round(lv["ms"], 4), "row_bytes": rb, "table_bytes": lv["table_bytes"], "resident_in": where,
                               "algorithmic_bytes": lv["algo_bytes"], "line_bytes": int(line),
                               "algorithmic_gbs": round(lv["algo_bytes"] / sec / 1e9, 1), "line_gbs": round(line / sec / 1e9, 1),
                               "gather_roof_gbs": roof_gbs, "frac_of_gather_roof": round(line / sec / 1e9 / roof_gbs, 4),
                               "line_frac_of_hbm_peak": round(line / sec / 1e9 / HBM_PEAK_GBS, 4)})
            roof["levels"] = levels
            roof["note"] = ("HIBF rows are narrower than a 128-byte line: `levels` (detail file) gives per tree level the rate in lines against "
                            "the measured gather roof for a table of that size (profiles/r03_calib_gather.json)")
        roof["traffic"] = None

        result = {
            "metric": "Mreads/s classified (150 bp) + IBF-lookup GB/s vs HBM roofline",
            "value": round(value, 3),
            "unit": "Mpairs/s" if paired else "Mreads/s",
            "n_gpus": gdist.group_size(),        # from the process group, not from the command line
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if kind == "slice" else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": (f"{name} (BASELINE.json configs[{spec['config']}]): {wl.filter_bytes / 2**30:.2f} GiB "
                             f"{'HIBF, ' + str(spec.get('user_bins')) + ' user bins' if kind == 'hibf' else ('column slice of a flat IBF, ' if kind == 'slice' else 'flat IBF, ') + str(spec.get('bins')) + ' technical bins'}"
                             f", h={spec['h']}, k={wl.k} w={wl.w}, {n_reads} synthetic {unit_name} per GPU, rel_cutoff={args.rel_cutoff}"),
                "workload_detail": f"{desc}, k={wl.k} w={wl.w}, {n_reads} synthetic "
                                   f"{unit_name} per GPU (50% cut from {4096} planted genomes), rel_cutoff={args.rel_cutoff}, seeded "
                                   f"{'Bernoulli(3/16)' if spec.get('fill') == '3/16' else 'Bernoulli(3/8)' if spec.get('skew') else 'Bernoulli(0.5)'} fill generated on the device, seed 42",
                "reads_per_gpu": n_reads,
                "parallelism": (f"bin-range partitioned x{world} of {spec.get('slices')} slices" if kind == "slice"
                                else f"read-sharded x{world}, filter replicated"),
                "mean_minimisers_per_read": round(tm["n_hashes"] / max(1, n_reads), 3),
                "classified_reads_rank0": n_class,
                "matches_rank0": int(len(matches)),
                "kernel_ms": {"minimiser": round(float(np.mean(mini_ms)), 3), "count_select": round(float(np.mean(count_ms)), 3),
                              "device_total": round(float(np.mean(total_ms)), 3)},
            },
            "roofline": roof,
            # the proof that n_gpus ranks on n_gpus devices measured this (rank_proof) + every rank's own step time
            "ranks": dict(proof, ms_per_step_min=round(min(per_rank_ms), 3), ms_per_step_max=round(max(per_rank_ms), 3),
                          ms_per_step_by_rank=",".join(f"{x:.2f}" for x in per_rank_ms), **(exchange or {})),
        }

        # HBM traffic of the dominant kernel comes from a SEPARATE rocprofv3 --pmc FETCH_SIZE pass (counter collection
        # cannot share a run with timing); its committed summary is attached when it was taken on this workload.
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_fetch_{name}.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                # evidence about THIS run only if the profile was taken on the same batch: same algorithmic bytes per step
                same = abs(pmc.get("algo_bytes_per_step", 0) - tm["algo_bytes"]) <= 1e-3 * max(1, tm["algo_bytes"])
                if same:
                    roof["traffic"] = pmc["hbm_bytes_per_launch"]
                    roof["traffic_source"] = pmc["source"]
                    roof["traffic_from"] = pmc.get("file") or str(pmc["source"]).split(" ")[0].rstrip(":")
                else:
                    roof["traffic_source"] = (f"none: {os.path.basename(pmc_path)} was taken on a batch of {pmc.get('algo_bytes_per_step')} "
                                              f"algorithmic bytes, this run's is {tm['algo_bytes']}")
            except Exception as e:  # noqa: BLE001
                log("bench.py: could not read", pmc_path, repr(e))

        # the same resident batch under the two conditions the headline does not show (not part of `value`)
        if headline and not args.no_variants and part is None and kind == "flat":
            variants = {}
            bpt = spec.get("bins_per_target", 1)
            if bpt == 1:  # (the split-bin kernel has no early exit); measured above for the roofline
                variants["no_early_exit"] = variant_of(full, full_tms)
            _, cms, _, tms, tmv = timed(0.2, 3, 1)        # ganon-classify's own default (Config.hpp:32): T ~ 4, nothing to exit from
            variants["rel_cutoff_0.2"] = dict(variant_of(rates(tmv, cms), tms), matches=int(tmv["n_matches"]))
            # ... and with the binary's low cutoff under the filter rules `ganon classify` passes by default (--rel-cutoff 0.2 --rel-filter 0.1 --fpr-query 1e-5,
            # /root/reference/src/ganon/config.py), the pre-pass of filter_matches running on the device (gn_stream_set_postfilter);
            # per-target fpr 0.5^h = what the Bernoulli(0.5) bits are
            st.set_postfilter(0.1, 1e-5, np.full(wl.bins // bpt, 1.0 - (1.0 - 0.5 ** spec["h"]) ** bpt, dtype=np.float64))
            if os.environ.get("GANON_BENCH_AB_PREDROP"):  # A/B: every match written, then judged
                with ganon_amd.ablate("predrop"):
                    _, cms, _, tms, tmv = timed(0.2, 3, 1)
                _, d_fil, d_fpr = st.fetch_postfilter()
                variants["wrapper_defaults_every_match_written"] = dict(ms_per_step=round(float(np.mean(tms)), 3),
                                                                        count_select_ms=round(float(np.mean(cms)), 3), dropped_rel_filter=d_fil,
                                                                        dropped_fpr_query=d_fpr, matches_after=int(st.fetch()[2][-1]))
            _, cms, _, tms, tmv = timed(0.2, 3, 1)
            _, d_fil, d_fpr = st.fetch_postfilter()
            variants["wrapper_defaults_device_filter_matches"] = dict(
                thresholds="--rel-cutoff 0.2 (the binary's default) --rel-filter 0.1 --fpr-query 1e-5 (what `ganon classify` passes)",
                ms_per_step=round(float(np.mean(tms)), 3), count_select_ms=round(float(np.mean(cms)), 3),
                mreads_per_s=round(n_reads / float(np.mean(tms)) / 1e3, 2), matches_before=int(variants["rel_cutoff_0.2"]["matches"]),
                dropped_rel_filter=d_fil, dropped_fpr_query=d_fpr, matches_after=int(st.fetch()[2][-1]))
            st.set_postfilter(None)
            result["variants"] = variants
            step(args.rel_cutoff)  # leave the headline batch in the stream for the checks below

        p001 = kind == "hibf" and spec.get("fill") == "3/16"
        if kind == "hibf" and (p001 or (headline and not args.no_variants and os.environ.get("GANON_BENCH_HIBF_LOW_CUTOFF"))):
            # The binary's own --rel-cutoff 0.2, plain and under the filter rules `ganon classify` passes.  At the reference's HIBF defaults
            # (p^h = 0.0012) chance matches are rare and this always runs; on the h = 3 / p^h = 0.05..0.125 workloads it is an A/B on request
            # (needs --reads <= 500000: 3 000 chance matches per read there): --rel-cutoff 0.2 plain, with the filter_matches pre-pass judging
            # every pair after the sort, and with the pairs it is bound to drop left out of the sort
            variants = {}
            small = p001 or n_reads <= 500_000   # (h = 3 without the pre-pass: the RESULT of 10 M reads is 27 G matches = 330 GB: only the pre-pass run fits)
            if small:
                _, cms, _, tms, tmv = timed(0.2, 3, 1)
                variants["rel_cutoff_0.2"] = dict(ms_per_step=round(float(np.mean(tms)), 3), matches=int(tmv["n_matches"]),
                                                  mreads_per_s=round(n_reads / float(np.mean(tms)) / 1e3, 2))
            st.set_postfilter(0.1, 1e-5, np.full(spec["user_bins"], 0.001 if p001 else 0.05, dtype=np.float64))
            for tag, env in ((("every_pair_sorted", "1"),) if small else ()) + (("device_filter_matches", None),):
                with ganon_amd.ablate("predrop" if env else ""):
                    _, cms, _, tms, tmv = timed(0.2, 3 if small else 2, 1)
                _, d_fil, d_fpr = st.fetch_postfilter()
                variants["low_cutoff_" + tag] = dict(thresholds="--rel-cutoff 0.2 --rel-filter 0.1 --fpr-query 1e-5", ms_per_step=round(float(np.mean(tms)), 3),
                                                     mreads_per_s=round(n_reads / float(np.mean(tms)) / 1e3, 2), dropped_rel_filter=d_fil,
                                                     dropped_fpr_query=d_fpr, matches_after=int(st.fetch()[2][-1]),
                                                     raw_pairs=int(d_fil + d_fpr + int(st.fetch()[2][-1])))
            st.set_postfilter(None)
            result["variants"] = variants
            step(args.rel_cutoff)

        # Every rank re-derives a sample of ITS OWN shard / owned range with the oracle (a wrong shard on rank 5 must not pass
        # because rank 0's was right); the mismatch counts are summed over the ranks into the line, and so are the per-rank
        # order-independent checksums of the match records (bench_workload.checksum_matches) -- with the same seed rank 0's
        # shard is the world-1 shard, so `match_checksum_rank0` of an N-GPU run equals the 1-GPU run's.
        import bench_cpu
        detail, ok = None, True
        if args.check:
            if kind == "hibf":
                ok, detail = bench_cpu.spot_check_hibf(wl, flt, nh, status, mo, matches, min(args.check, 600))
            else:
                ok, detail = bench_cpu.spot_check(wl, flt, nh, status, mo, matches, args.check if rank == 0 else max(200, args.check // 4),
                                                  target_offset=(sl_idx * spec["bins"] if kind == "slice" else 0),
                                                  bins_per_target=spec.get("bins_per_target", 1),
                                                  read_range=(lo, hi) if kind == "slice" else None,
                                                  own_targets_only=kind == "slice" and world > 1)
            if world > 1:
                detail = {"reads_checked": int(gdist.sum_over_ranks(detail["reads_checked"], device=red_dev)),
                          "matches_checked": int(gdist.sum_over_ranks(detail["matches_checked"], device=red_dev)),
                          "mismatching_reads": int(gdist.sum_over_ranks(detail["mismatching_reads"], device=red_dev)),
                          "ranks_checked": world}
                ok = detail["mismatching_reads"] == 0
        csum = bw.checksum_matches(matches)
        if world > 1:  # (a 64-bit sum does not fit a float reduction: two 32-bit halves)
            lo32 = int(gdist.sum_over_ranks(csum & 0xFFFFFFFF, device=red_dev))
            hi32 = int(gdist.sum_over_ranks(csum >> 32, device=red_dev))
            csum_all = (lo32 + (hi32 << 32)) & 0xFFFFFFFFFFFFFFFF
        else:
            csum_all = csum
        if rank == 0:
            result["config"]["match_checksum_rank0"] = f"{csum:016x}"
            result["config"]["match_checksum_all_ranks"] = f"{csum_all:016x}"
            if detail is not None:
                result["config"]["oracle_spot_check"] = detail
                result["config"]["oracle_mismatching_reads"] = detail["mismatching_reads"]   # (flat: survives the driver's record)
                if not ok:
                    log("bench.py: ORACLE SPOT CHECK FAILED:", detail)
                    result["value"] = None
            if headline and not args.no_cpu_baseline and world == 1 and kind == "flat" and wl.filter_bytes <= (16 << 30) and spec.get("bins_per_target", 1) == 1:
                try:  # the CPU baseline is an N=1 measurement on a filter the host can hold
                    result["cpu_baseline"] = bench_cpu.cpu_baseline(wl, flt, args.cpu_sample, gpu_result=(nh, mo, matches))
                except Exception as e:  # noqa: BLE001 -- a reported extra; never lose the GPU line over it
                    log("bench.py: cpu_baseline failed:", repr(e))
                    result["cpu_baseline"] = None
            if headline:
                result.setdefault("cpu_baseline", None)
        if part is not None:
            part.close()
        else:
            st.destroy()
        flt.free()
        return result

    result = run_one(args.workload, True)
    default_run = not args.no_extra and args.workload == "flat8g" and not args.reads and not args.rows
    extras_multi = EXTRA_WORKLOADS_MULTI
    if os.environ.get("GANON_BENCH_EXTRAS"):   # dry runs: small stand-ins for the extras, e.g. "tiny,slice_tiny"
        extras_multi = [w for w in os.environ["GANON_BENCH_EXTRAS"].split(",") if w in WORKLOADS]
        default_run = not args.no_extra
    if default_run and world == 1 and rank == 0:
        result["other_workloads"] = [run_extra(w, args.extra_timeout) for w in EXTRA_WORKLOADS]
        if not args.no_e2e:
            result["e2e"] = run_e2e(args.e2e_budget)
    elif default_run and world > 1:
        # the configs BASELINE.json quotes its scaling target on, with the real world size (every rank takes part).
        # They run behind a watchdog: if a rank fails inside a collective the others would wait for ever, and the headline
        # measured above must not be lost over an extra -- after --extra-timeout seconds rank 0 prints the line it has and
        # every rank leaves.
        import threading
        others = []

        def give_up():
            log(f"[rank {rank}] bench.py: the in-job extras did not finish within {args.extra_timeout} s -- leaving without them")
            if rank == 0:
                result["other_workloads"] = others + [{"workload": "(in-job extras)", "error": f"timeout after {args.extra_timeout}s"}]
                emit(result)
            os._exit(0)

        watchdog = threading.Timer(args.extra_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        for w in extras_multi:
            t0 = time.time()
            try:
                r = run_one(w, False)
                others.append(slim(w, r, time.time() - t0))
            except Exception as e:  # noqa: BLE001 -- the headline line is never lost over an extra
                log(f"[rank {rank}] bench.py: workload {w} failed:", repr(e))
                others.append({"workload": w, "error": repr(e)[:400]})
                break   # (the ranks may be out of step now: no further collective work)
        result["other_workloads"] = others
        watchdog.cancel()
    # RCCL writes a version banner through C stdio (buffered until exit when stdout is a pipe).  Every rank pushes its
    # buffered C output out BEFORE the last barrier, rank 0 prints the JSON line after it: the JSON is the last line of
    # the job's stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    import torch.distributed as dist
    if dist.is_initialized():
        out_of_step = any("error" in o for o in result.get("other_workloads", [])) if world > 1 else False
        if out_of_step:   # (an extra failed on some rank: a barrier could wait for ever)
            if rank == 0:
                emit(result)
            sys.stdout.flush()
            os._exit(0)
        gdist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if default_run and world > 1 and not args.no_e2e:
            # The N-GPU end-to-end leg, after the ranks have left the job (the barrier above was their last act; their device memory
            # goes with their processes): this process frees what it holds and runs the product binary over every GPU of the node.
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            time.sleep(3.0)
            dev_list = "all" if not proof["shared_gpu_dry_run"] else ",".join(["0"] * world)
            # (16 M reads as at N = 1: writing and compressing the input files is most of this leg's time, and the scaling run has a clock too)
            result["e2e"] = run_e2e(min(args.e2e_budget, 150), devices=dev_list, only="fastq,gz",
                                    reads=int(os.environ.get("GANON_BENCH_E2E_READS", "0")) or 16_000_000)
        emit(result)
    return 0


if __name__ == "__main__":
    sys.exit(main())
