"""bench.py's checker legs -- the only places outside tests/ and smoke() that touch oracle/:

  spot_check    re-derives a random sample of the GPU results with the CPU oracle (parity gate on the bench data)
  cpu_baseline  times the oracle (OpenMP port of the reference's per-read loop) on a bounded sample of the same
                workload on this box's host cores -- and, when a REAL `ganon-classify` (one that is not this repo's) is on PATH
                or named by $GANON_REFERENCE_CLASSIFY, runs that on the same reads against an .ibf of the same bits (BASELINE.md
                3.1), diffs its .all with the GPU's matches and reports it as `"kind": "reference"`
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


OUR_CLASSIFY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ganon_amd", "host", "ganon-classify")


def find_reference_classify():
    """-> (path, why_not): a ganon-classify that is not this repo's.  $GANON_REFERENCE_CLASSIFY names one explicitly (the test suite
    points it at our own binary to exercise this leg: there is no SeqAn3 build in the image); otherwise PATH is searched and a binary
    whose --version carries this repo's tag is refused."""
    import shutil
    import subprocess
    forced = os.environ.get("GANON_REFERENCE_CLASSIFY")
    cand = forced or shutil.which("ganon-classify")
    if not cand or not os.access(cand, os.X_OK):
        return None, "no ganon-classify on PATH"
    if not forced:
        try:
            if os.path.exists(OUR_CLASSIFY) and os.path.samefile(cand, OUR_CLASSIFY):
                return None, "the ganon-classify on PATH is this repo's"
            v = subprocess.run([cand, "--version"], capture_output=True, text=True, timeout=30)
            if "mi355x" in (v.stdout + v.stderr):
                return None, "the ganon-classify on PATH is this repo's"
        except Exception as e:  # noqa: BLE001
            return None, f"could not run {cand}: {e!r}"
    return cand, ""


def reference_baseline(binary: str, wl, flt, nh, mo, matches, n: int, threads: int, workdir: str = ""):
    """the given ganon-classify with `--threads <cores> --output-all` on the first n reads of the workload (FASTQ,
    ids r<idx>) and an .ibf of the SAME bits written by this repo's writer (ibf_file.save_ibf: bin b is target "b"); its own
    "classifying+printing elapsed" line (GanonClassify.cpp:1047) -> Mreads/s; its .all is compared with the GPU's matches."""
    import re
    import shutil
    import subprocess
    import tempfile

    from ganon_amd import ibf_file
    n = int(min(n, wl.n_reads))
    base = workdir or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    need = wl.filter_bytes + n * 2 * (wl.read_len + 16) + (64 << 20)
    if shutil.disk_usage(base).free < need:
        raise RuntimeError(f"{base} has less than {need >> 20} MiB free for the .ibf and the FASTQ")
    d = tempfile.mkdtemp(prefix="ganon_ref_baseline_", dir=base)
    try:
        import bench_e2e
        L = wl.read_len
        if getattr(wl, "paired", False) or int(wl.off[n]) != n * L:
            raise RuntimeError("the reference leg takes single-end reads of one length")
        fq = os.path.join(d, "reads.fq")
        step = 1 << 20
        with open(fq, "wb") as f:
            for a in range(0, n, step):
                b = min(n, a + step)
                f.write(bench_e2e.fastq_matrix(wl.bases[a * L:b * L], b - a, L, first_id=a).tobytes())
        ibf = os.path.join(d, "filter.ibf")
        cfg = dict(n_bins=wl.bins, max_hashes_bin=1, hash_functions=wl.hash_funs, kmer_size=wl.k, window_size=wl.w, bin_size_bits=wl.rows,
                   max_fp=0.05, true_max_fp=0.05, true_avg_fp=0.05)
        ibf_file.save_ibf(ibf, flt, cfg, [], [(b, str(b)) for b in range(wl.bins)], wl.bins, wl.rows, wl.hash_funs)
        out = os.path.join(d, "out")
        cmd = [binary, "--ibf", ibf, "--single-reads", fq, "--output-prefix", out, "--output-all", "--threads", str(threads),
               "--rel-cutoff", repr(float(wl.rel_cutoff)), "--rel-filter", "1", "--skip-lca"]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            raise RuntimeError(f"{binary} rc {p.returncode}: {p.stderr[-300:]}")
        m = re.search(r"classifying\+printing elapsed \(s\): ([0-9.eE+-]+)", p.stderr)
        sec = float(m.group(1)) if m else wall
        # its .all (read id, target, count) against the device's matches of the same reads
        got = np.loadtxt(out + ".all", dtype=str, delimiter="\t", ndmin=2) if os.path.getsize(out + ".all") else np.zeros((0, 3), dtype=str)
        theirs = sorted((int(r[0][1:]), int(r[1]), int(r[2])) for r in got)
        ours = sorted((int(x["read"]), int(x["target"]), int(x["count"])) for x in matches[: int(mo[n])])
        return {"value": round(n / sec / 1e6, 4), "seconds": round(sec, 3), "process_wall_s": round(wall, 2), "reads": n,
                "matches": len(theirs), "agrees_with_ours": theirs == ours, "timed_by": "its classifying+printing line" if m else "process wall",
                "binary": binary}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(wl, flt, n_sample: int = 0, gpu_result=None):
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
    port = {
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
    # the real binary when there is one; the port's figure stays beside it
    binary, why_not = find_reference_classify()
    if binary and gpu_result is not None:
        try:
            nh, mo, matches = gpu_result
            ref = reference_baseline(binary, wl, flt, nh, mo, matches, min(n, max(100_000, int(port["value"] * 1e6 * 10))), threads)
            return {"value": ref["value"], "unit": "Mreads/s", "cores": threads, "kind": "reference", "agrees_with_ours": ref["agrees_with_ours"],
                    "port_value": port["value"],
                    "sample": f"{ref['binary']} --threads {threads} --output-all on the first {ref['reads']} reads of the same workload (FASTQ) and an .ibf of the "
                              f"same bits written by this repo's writer: {ref['seconds']} s by {ref['timed_by']} (process wall {ref['process_wall_s']} s), "
                              f"{ref['matches']} .all lines {'==' if ref['agrees_with_ours'] else '!='} the GPU's matches; the CPU port: {port['value']} Mreads/s"}
        except Exception as e:  # noqa: BLE001 -- the port's figure is still a baseline
            port["sample"] += f"; reference binary {binary} failed: {e!r}"[:300]
    else:
        port["sample"] += f"; no reference binary ({why_not})" if not binary else ""
    return port
