#!/usr/bin/env python3
"""host_ceiling.py -- what the HOST side of `ganon-classify --device a,b,...` sustains when the devices cost (next to) nothing.

The reference runs ONE parser thread (GanonClassify.cpp:1436-1441, parse_reads :1220-1287) in front of N classify threads
(:1579-1597); this repo mirrors that: one reader task, one worker per --device entry, a post pool and an ordered merge + write.
On one GPU plain FASTQ is bound by the link; what the reader + post stage + writer could feed to EIGHT GPUs is not something a
one-GPU box shows directly.  This script measures it as far as one box can:

  * `$GANON_HIP_ABLATE=fake_count`: upload, record index and minimisers run, the count + select kernels do not, every second read
    gets one made-up match -- the device step costs ~1 ms per million reads, the host does everything it does in a real run
    (read the file, page-locked copies, header fetch, filter_matches, LCA bookkeeping, .all / .rep text, ordered write);
  * `--device 0,0,0,0,0,0,0,0`: eight workers with their lanes, as on an eight-GPU node -- but ONE link, so plain FASTQ is still
    capped by ~50 GB/s of PCIe (~160 Mreads/s); `.fq.gz` moves a quarter of the bytes;
  * `[host cpu]`: CPU seconds per group of host threads, per million reads.  A group of T threads that needs c CPU-seconds per
    million reads cannot pass more than T / c Mreads/s: the table's `ceiling` column.  The smallest ceiling is the host's.

usage (GPU box): python scripts/host_ceiling.py [--reads 16000000] [--workers 8] [--runs 3] > gpurun_out/host_ceiling.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# threads per group as the binary starts them: reader 1 (slab parsers are its helpers and only run when the device has no text
# waiting), workers = --device entries, post pool = $GANON_HOST_POST_THREADS (default: see classify.cpp), merge + write 1
GROUP_THREADS = {"reader": 1, "merge_and_write": 1}


def e2e(extra, budget=400):
    cmd = [sys.executable, os.path.join(ROOT, "bench_e2e.py"), "--budget", str(budget)] + extra
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=budget + 300)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"rc {p.returncode}: {p.stderr.decode(errors='replace')[-600:]}"}
    return json.loads(lines[-1])


def table(run, workers, post_threads):
    out = {}
    for name, r in (run.get("inputs") or {}).items():
        if "error" in r or "host_cpu_s_per_munit" not in r:
            out[name] = {"error": r.get("error", "no [host cpu] line")}
            continue
        per = r["host_cpu_s_per_munit"]
        rows = {}
        for g, c in per.items():
            if g in ("whole_process", "classify_phase_all_threads") or c <= 0:
                continue
            t = GROUP_THREADS.get(g, workers if g == "device_workers" else post_threads if g == "post_pool" else 1)
            rows[g] = {"cpu_s_per_munit": c, "threads": t, "ceiling_munits_s": round(t / c, 1)}
        # every thread of the process between the level's start and its end: the steady state's CPU bill (the whole process' includes the
        # runtime's start-up, the filter load and the parsers' head start)
        whole = per.get("classify_phase_all_threads") or per.get("whole_process", 0.0)
        out[name] = {"measured_munits_s": r["rate"], "groups": rows, "classify_phase_cpu_s_per_munit": whole,
                     "whole_process_cpu_s_per_munit": per.get("whole_process"),
                     "ceiling_by_cores": {str(k): round(k / whole, 1) for k in (16, 32, 64, 128)} if whole else None,
                     "smallest_group_ceiling": min(((v["ceiling_munits_s"], g) for g, v in rows.items()), default=None),
                     "host_stalls": r.get("host_stalls", "")[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=16_000_000)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--post-threads", type=int, default=0, help="0 = the binary's default")
    ap.add_argument("--only", default="fastq,gz,paired")
    ap.add_argument("--skip-real", action="store_true", help="only the N-worker run with the device step ablated")
    args = ap.parse_args()
    dev = ",".join(["0"] * args.workers)
    common = ["--reads", str(args.reads), "--runs", str(args.runs), "--only", args.only]
    env = ["--env", "GANON_HIP_ABLATE=fake_count"]
    if args.post_threads:
        env += ["--env", f"GANON_HOST_POST_THREADS={args.post_threads}"]
    post = args.post_threads or max(3, min(args.workers, 8))
    res = {"reads": args.reads, "workers": args.workers, "post_threads": post}
    if not args.skip_real:
        res["real_one_worker_set"] = e2e(common)                                   # the product as it runs on one GPU (device work included)
        res["fake_one_device_entry"] = e2e(common + env + ["--devices", "0"])      # one worker, device step ~free: the link + one worker's host side
    res["fake_n_workers"] = e2e(common + env + ["--devices", dev])             # N workers on the one GPU: the host side of an N-GPU node, one link
    res["table_fake_n_workers"] = table(res["fake_n_workers"], args.workers, post)
    if not args.skip_real:
        res["table_real_one_gpu"] = table(res["real_one_worker_set"], 3, 3)
    print(json.dumps(res), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
