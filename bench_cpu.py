"""bench.py's checker legs -- the only places outside tests/ and smoke() that touch oracle/:

  spot_check    re-derives a random sample of the GPU results with the CPU oracle (parity gate on the bench data)
  cpu_baseline  times the oracle (OpenMP port of the reference's per-read loop) on a bounded sample of the same
                workload on this box's host cores
"""
from __future__ import annotations

import os
import time

import numpy as np


def _oracle_native():
    """Rebuild the oracle with -march=native into /tmp when possible (fair CPU baseline on this host);
    fall back to the portable in-tree build."""
    import ctypes
    import shutil
    import subprocess
    import tempfile

    import oracle
    try:
        src = os.path.join(os.path.dirname(oracle.__file__), "ganon_oracle.c")
        out = os.path.join(tempfile.gettempdir(), f"libganon_oracle_native_{os.getuid()}.so")
        if shutil.which("gcc"):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-fopenmp", "-shared", "-o", out,
                                   src, "-lm"], stderr=subprocess.DEVNULL)
            ctypes.CDLL(out)
            oracle._LIB_PATH = out
            oracle._lib = None
            oracle.build = lambda force=False: out
            return "native"
    except Exception:
        pass
    return "portable"


def spot_check(wl, flt, nh, status, mo, matches, n_sample: int):
    """GPU (read, target, count) lists of a random read sample == oracle select_matches on the device's bits."""
    import bench_workload as bw
    import oracle

    bw.download_filter(flt, wl)
    _, ibf = bw.oracle_filter(wl)
    rng = np.random.default_rng(123)
    n = wl.n_reads
    idx = np.unique(rng.integers(0, n, size=min(n_sample, n)))
    bad = 0
    checked_matches = 0
    for r in idx.tolist():
        seq = wl.bases[int(wl.off[r]):int(wl.off[r + 1])]
        hashes = oracle.minimiser_hash(oracle.to_ranks(seq), wl.k, wl.w)
        counts = ibf.bulk_count(hashes).astype(np.int64)
        thr = oracle.threshold_cutoff(len(hashes), wl.rel_cutoff)
        capped = np.minimum(counts, len(hashes))
        tg = np.nonzero(capped >= thr)[0]
        exp = [(int(t), int(capped[t])) for t in tg]
        got = [(int(x["target"]), int(x["count"])) for x in matches[int(mo[r]):int(mo[r + 1])]]
        if nh[r] != len(hashes) or status[r] != 0 or got != exp:
            bad += 1
        checked_matches += len(exp)
    return bad == 0, {"reads_checked": int(len(idx)), "matches_checked": int(checked_matches), "mismatching_reads": int(bad)}


def cpu_baseline(wl, flt, n_sample: int = 0):
    import bench_workload as bw
    import oracle

    build = _oracle_native()
    threads = os.cpu_count() or 1
    ofl, ibf = bw.oracle_filter(wl)
    ranks_all = None

    def run(n):
        nonlocal ranks_all
        n = min(n, wl.n_reads)
        seg = wl.bases[: int(wl.off[n])]
        ranks = oracle.to_ranks(seg)
        t0 = time.perf_counter()
        total, nh, nm, ck = oracle.baseline_classify(ofl, ranks, wl.off[: n + 1], wl.k, wl.w, threads)
        return n, time.perf_counter() - t0, total

    if n_sample <= 0:
        run(20_000)                                  # warm-up (page-in, thread pool)
        n, dt, _ = run(200_000)                      # probe
        rate = n / max(dt, 1e-6)
        n_sample = int(min(wl.n_reads, max(200_000, rate * 20.0)))  # ~20 s of CPU work
    n, dt, total = run(n_sample)
    return {
        "value": round(n / dt / 1e6, 4),
        "unit": "Mreads/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {n} reads of the same workload against the same filter bits, {threads} OpenMP threads, "
                  f"{dt:.1f} s, oracle build: {build}; minimiser + bulk_count + select per read "
                  f"(GanonClassify.cpp:676-735), {total} matches",
    }
