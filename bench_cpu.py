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


def spot_check(wl, flt, nh, status, mo, matches, n_sample: int, target_offset: int = 0, bins_per_target: int = 1,
               read_range=None, own_targets_only: bool = False):
    """GPU (read, target, count) lists of a random read sample == oracle select_matches on the device's bits.  Only
    the rows the sample touches are fetched from the device (gn_filter_download_row_list), so this also works on the
    128 GiB filters; `target_offset` = first global target of a column slice."""
    import bench_workload as bw

    ibf = bw.sampled_oracle_ibf(flt, wl)
    rng = np.random.default_rng(123)
    n = wl.n_reads
    lo, hi = read_range if read_range is not None else (0, n)   # partitioned filter, N > 1: this rank holds the matches of [lo, hi)
    idx = np.unique(rng.integers(lo, hi, size=min(n_sample, hi - lo)))
    bad = 0
    checked_matches = 0
    t_lo, t_hi = target_offset, target_offset + wl.bins // bins_per_target
    for r in idx.tolist():
        n_h, exp = bw.oracle_read_matches(ibf, wl, r, bins_per_target)
        exp = [(t + target_offset, c) for t, c in exp]
        got = [(int(x["target"]), int(x["count"])) for x in matches[int(mo[r]):int(mo[r + 1])]]
        if own_targets_only:  # the other ranks' targets arrive through the exchange; this rank can only re-derive its own
            got = [g for g in got if t_lo <= g[0] < t_hi]
        if nh[r] != n_h or status[r] != 0 or got != exp:
            bad += 1
        checked_matches += len(exp)
    return bad == 0, {"reads_checked": int(len(idx)), "matches_checked": int(checked_matches), "mismatching_reads": int(bad)}


def spot_check_hibf(wl, flt, nh, status, mo, matches, n_sample: int):
    """same for an HIBF workload: the oracle's counting_agent_type::bulk_count on the downloaded IBFs"""
    import bench_workload as bw
    import oracle

    bw.download_hibf(flt, wl)
    hb = oracle.Hibf([oracle.Ibf(b, s, h, r) for (r, b, s, h) in wl.ibfs], wl.next_ibf_id, wl.bin_to_user, wl.n_user_bins)
    rng = np.random.default_rng(123)
    idx = np.unique(rng.integers(0, wl.n_reads, size=min(n_sample, wl.n_reads)))
    bad = 0
    checked_matches = 0
    for r in idx.tolist():
        seq = wl.bases[int(wl.off[r]):int(wl.off[r + 1])]
        hh = oracle.minimiser_hash(oracle.to_ranks(seq), wl.k, wl.w)
        counts = hb.bulk_count(hh, oracle.threshold_cutoff(len(hh), wl.rel_cutoff))
        nz = np.nonzero(counts)[0]
        exp = [(int(u), int(min(int(counts[u]), len(hh)))) for u in nz]
        got = [(int(x["target"]), int(x["count"])) for x in matches[int(mo[r]):int(mo[r + 1])]]
        if nh[r] != len(hh) or status[r] != 0 or got != exp:
            bad += 1
        checked_matches += len(exp)
    return bad == 0, {"reads_checked": int(len(idx)), "matches_checked": int(checked_matches), "mismatching_reads": int(bad)}


def usable_cores() -> int:
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max) -- the GPU boxes
    show 256 logical CPUs but grant a quota of 16; 256 OpenMP threads on 16 cores cost 25 x per read (r01's 650 us)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:  # noqa: BLE001
            pass
    return n


def _host_filter_buffer(n_words: int):
    """Host memory for the CPU baseline's copy of the filter: anonymous mapping with transparent huge pages requested
    and, while it is first touched, pages interleaved over the NUMA nodes (set_mempolicy) -- on 4 KiB pages bound to one
    node every row of a multi-GiB filter costs a TLB miss and all threads queue on one memory controller.
    Returns (uint64 array, description)."""
    import ctypes
    import mmap
    note = []
    nbytes = n_words * 8
    mm = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    try:
        mm.madvise(mmap.MADV_HUGEPAGE)
        note.append("THP")
    except Exception:  # noqa: BLE001
        pass
    try:
        nodes = open("/sys/devices/system/node/online").read().strip()
        ids = []
        for part in nodes.split(","):
            a, _, b = part.partition("-")
            ids += list(range(int(a), int(b or a) + 1))
        if len(ids) > 1:
            mask = ctypes.c_ulong(sum(1 << i for i in ids))
            libc = ctypes.CDLL(None, use_errno=True)
            if libc.syscall(238, 3, ctypes.byref(mask), ctypes.c_ulong(max(ids) + 2)) == 0:  # set_mempolicy(MPOL_INTERLEAVE)
                note.append(f"interleaved over {len(ids)} NUMA nodes")
    except Exception:  # noqa: BLE001
        pass
    arr = np.frombuffer(mm, dtype=np.uint64)
    return arr, mm, ", ".join(note) or "default pages"


def _reset_mempolicy():
    try:
        import ctypes
        ctypes.CDLL(None).syscall(238, 0, None, 0)  # MPOL_DEFAULT
    except Exception:  # noqa: BLE001
        pass


def cpu_baseline(wl, flt, n_sample: int = 0):
    import bench_workload as bw
    import oracle

    build = _oracle_native()
    threads = usable_cores()
    mem_note = "numpy pages"
    if getattr(wl, "filter_rows", None) is None:  # device-generated filter: the oracle needs its bits on the host
        arr, keep, mem_note = _host_filter_buffer(wl.rows * wl.bin_words)
        wl.filter_rows = arr.reshape(wl.rows, wl.bin_words)
        wl._filter_keep = keep
        bw.download_filter(flt, wl)          # first touch happens here, under the interleave policy
        _reset_mempolicy()
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
        "us_per_read_per_thread": round(dt * threads / n * 1e6, 1),
        "sample": f"first {n} reads of the same workload against the same filter bits, {threads} OpenMP threads (= usable cores: "
                  f"affinity {len(os.sched_getaffinity(0))}, cgroup quota applied), "
                  f"{dt:.1f} s, oracle build: {build}, filter memory: {mem_note}, rows software-prefetched per read; "
                  f"minimiser + bulk_count + select per read (GanonClassify.cpp:676-735), {total} matches",
    }
