"""Synthetic workload builders shared by bench.py, __graft_entry__.smoke() and the full-size GPU tests.

Everything is seeded and generated with numpy on the host; planted genomes are hashed and emplaced on the
device through the C ABI, and the filter is downloaded back before the CPU oracle looks at it, so parity is
checked on exactly the device's bits (SURVEY.md 8d 'Synthetic inputs').  Nothing here touches oracle/ except
oracle_filter(), which only the checker legs (smoke, tests, cpu_baseline) call.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _fill_random_u64(out: np.ndarray, seed: int, threads: int) -> None:
    """iid Bernoulli(0.5) bits, generated in parallel chunks (PCG64 streams keyed by (seed, chunk))."""
    flat = out.reshape(-1)
    n = flat.size
    chunk = 1 << 24  # 128 MiB
    spans = [(i, min(n, i + chunk)) for i in range(0, n, chunk)]

    def work(idx_span):
        idx, (a, b) = idx_span
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        flat[a:b] = rng.integers(0, 1 << 64, size=b - a, dtype=np.uint64)

    with cf.ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(work, enumerate(spans)))


@dataclass
class FlatWorkload:
    name: str
    bins: int
    rows: int          # bin_size S
    hash_funs: int
    k: int
    w: int
    rel_cutoff: float
    read_len: int
    n_reads: int
    planted_fraction: float
    seed: int
    filter_rows: np.ndarray      # uint64 [S, W] host copy == device content
    bases: np.ndarray            # uint8 ASCII [n_reads*read_len]
    off: np.ndarray              # uint64 [n_reads+1]
    density: float = 0.5
    genomes: Optional[np.ndarray] = None      # uint8 ASCII [n_genomes, genome_len]
    genome_bins: Optional[np.ndarray] = None  # uint32 [n_genomes]

    @property
    def bin_words(self) -> int:
        return (self.bins + 63) >> 6

    @property
    def filter_bytes(self) -> int:
        return self.rows * self.bin_words * 8


def make_flat_workload(name: str, bins: int, rows: int, hash_funs: int, n_reads: int, read_len: int = 150, k: int = 19,
                       w: int = 31, rel_cutoff: float = 0.75, planted_fraction: float = 0.5, genome_len: int = 3000,
                       n_genomes: Optional[int] = None, seed: int = 42, threads: Optional[int] = None,
                       shard: int = 0) -> FlatWorkload:
    """Flat IBF whose bit matrix is iid Bernoulli(0.5) (per-hash per-bin false-positive rate 0.5**h); reads are
    uniform iid ACGT with `planted_fraction` of them cut from `n_genomes` random genomes.  The genomes'
    minimisers still have to be OR-ed into the filter: `plant_genomes()` does that on the device through the C
    ABI (gn_submit_batch -> hashes -> gn_filter_emplace).  `shard` only changes the read stream (filter
    replicas are identical on every rank)."""
    threads = threads or min(32, os.cpu_count() or 1)
    W = (bins + 63) >> 6
    data = np.empty((rows, W), dtype=np.uint64)
    _fill_random_u64(data, seed, threads)
    if bins & 63:
        data[:, W - 1] &= np.uint64((1 << (bins & 63)) - 1)

    rng = np.random.default_rng([seed, 1])
    n_genomes = n_genomes if n_genomes is not None else min(bins, 4096)
    genomes = rng.integers(0, 4, size=(n_genomes, genome_len), dtype=np.uint8)
    gbins = ((np.arange(n_genomes, dtype=np.uint64) * np.uint64(max(1, bins // n_genomes))) % np.uint64(bins)).astype(np.uint32)

    rrng = np.random.default_rng([seed, 2, shard])
    reads = rrng.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)
    n_pl = int(n_reads * planted_fraction)
    if n_pl and genome_len > read_len:
        which = rrng.integers(0, n_genomes, size=n_pl)
        pos = rrng.integers(0, genome_len - read_len, size=n_pl)
        idx = pos[:, None] + np.arange(read_len)[None, :]
        sel = np.arange(n_pl) * 2 if n_pl * 2 <= n_reads else np.arange(n_pl)  # interleave planted / random reads
        reads[sel] = genomes[which[:, None], idx]
    bases = ACGT[reads].reshape(-1)
    off = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    wl = FlatWorkload(name, bins, rows, hash_funs, k, w, rel_cutoff, read_len, n_reads, planted_fraction, seed,
                      data, bases, off)
    wl.genomes = ACGT[genomes]
    wl.genome_bins = gbins
    return wl


def plant_genomes(hip_filter, wl: FlatWorkload) -> int:
    """OR the minimisers of wl.genomes into their bins ON THE DEVICE (product path only): the genomes are
    hashed by gn_minimiser_kernel and scattered by gn_emplace_kernel.  Returns the number of (hash, bin) pairs."""
    from ganon_amd import HipStream
    g = wl.genomes
    n, L = g.shape
    st = HipStream(hip_filter, n, n * L)
    st.upload(g.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L), None)
    st.minimisers(wl.k, wl.w)
    ho, hs = st.fetch_hashes()
    st.destroy()
    bins = np.repeat(wl.genome_bins, np.diff(ho).astype(np.int64)).astype(np.uint32)
    hip_filter.emplace(hs, bins)
    return len(hs)


def download_filter(hip_filter, wl: FlatWorkload) -> None:
    """refresh the host copy of the filter from the device (so the CPU oracle sees exactly the device bits)"""
    step = max(1, (1 << 30) // (wl.bin_words * 8))
    for r0 in range(0, wl.rows, step):
        n = min(step, wl.rows - r0)
        wl.filter_rows[r0:r0 + n] = hip_filter.download_rows(r0, n, wl.bin_words)


def oracle_filter(wl: FlatWorkload):
    """oracle.Filter over the workload's flat IBF (identity bin->target map)."""
    import oracle
    ibf = oracle.Ibf(wl.bins, wl.rows, wl.hash_funs, wl.filter_rows)
    return oracle.Filter(ibf=ibf, targets=[str(i) for i in range(wl.bins)], target_bins=[[i] for i in range(wl.bins)],
                         rel_cutoff=wl.rel_cutoff), ibf


# ---------------------------------------------------------------------------------------------------------------
# Device-generated flat workloads (BASELINE.json configs[3] and [4]: 128 GiB filters never exist on the host).
# The bit matrix is gn_filter_fill_random's seeded Bernoulli(0.5) (ganon_amd.fill_random_words is its numpy twin),
# planted genomes are emplaced on the device, and the oracle side fetches only the rows a read sample touches.
# ---------------------------------------------------------------------------------------------------------------
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)  # complement of ranks A C G T
_ACGT256 = np.tile(ACGT, 64)                    # random byte -> uniform random base


@dataclass
class DeviceFlatWorkload:
    name: str
    bins: int                 # bins of THIS filter (a column slice when row_words_total > bin_words)
    rows: int
    hash_funs: int
    k: int
    w: int
    rel_cutoff: float
    read_len: int
    n_reads: int              # reads, or pairs when paired
    paired: bool
    planted_fraction: float
    seed: int
    word_lo: int              # first global word of this column slice
    row_words_total: int      # words per row of the whole filter
    bases: np.ndarray         # mate-1 block, then mate-2 block
    off: np.ndarray           # off1
    off2: Optional[np.ndarray]
    genomes: np.ndarray       # ASCII [n_genomes, genome_len]
    genome_bins: np.ndarray   # LOCAL bin of each genome
    planted_genome: np.ndarray  # genome index of read i (-1: random read)
    filter_rows: Optional[np.ndarray] = None  # host copy, only when a checker downloaded it (download_filter)

    @property
    def bin_words(self) -> int:
        return (self.bins + 63) >> 6

    @property
    def filter_bytes(self) -> int:
        return self.rows * self.bin_words * 8


def make_device_flat_workload(name: str, bins: int, rows: int, hash_funs: int, n_reads: int, paired: bool = False,
                              read_len: int = 150, k: int = 19, w: int = 31, rel_cutoff: float = 0.75,
                              planted_fraction: float = 0.5, genome_len: int = 3000, n_genomes: int = 4096,
                              fragment_len: int = 400, seed: int = 42, shard: int = 0, word_lo: int = 0,
                              row_words_total: int = 0, threads: Optional[int] = None) -> DeviceFlatWorkload:
    """Reads only (the filter is filled on the device by `device_filter`).  Every second read / pair is cut from
    one of `n_genomes` random genomes; a pair is the two ends of a `fragment_len` fragment, mate 2 reverse-
    complemented (canonical minimisers make it hit the genome's bin like mate 1).  Generated in parallel chunks with
    their own PCG64 streams keyed by (seed, shard, chunk)."""
    threads = threads or min(32, os.cpu_count() or 1)
    W = (bins + 63) >> 6
    rng = np.random.default_rng([seed, 1])
    genomes = rng.integers(0, 4, size=(n_genomes, genome_len), dtype=np.uint8)
    gbins = ((np.arange(n_genomes, dtype=np.uint64) * np.uint64(max(1, bins // n_genomes))) % np.uint64(bins)).astype(np.uint32)
    n_mates = 2 if paired else 1
    bases = np.empty(n_reads * read_len * n_mates, dtype=np.uint8)
    m1 = bases[: n_reads * read_len].reshape(n_reads, read_len)
    m2 = bases[n_reads * read_len:].reshape(n_reads, read_len) if paired else None
    planted = np.full(n_reads, -1, dtype=np.int32)
    span = fragment_len if paired else read_len
    chunk = 1 << 19
    spans = [(a, min(n_reads, a + chunk)) for a in range(0, n_reads, chunk)]

    # genomes and their reverse complements as flat ASCII; a read is one row of a sliding-window view (a 150-byte copy)
    from numpy.lib.stride_tricks import sliding_window_view
    g_fw = np.ascontiguousarray(ACGT[genomes]).reshape(-1)
    g_rc = np.ascontiguousarray(ACGT[_COMP[genomes[:, ::-1]]]).reshape(-1)
    win_fw = sliding_window_view(g_fw, read_len)
    win_rc = sliding_window_view(g_rc, read_len)

    def work(idx_span):
        idx, (a, b) = idx_span
        r = np.random.default_rng([seed, 2, shard, idx])
        n = b - a
        pl = np.arange(a, b) % 2 == 0
        if planted_fraction < 0.5:
            pl &= r.random(n) < 2 * planted_fraction
        if genome_len <= span:
            pl[:] = False
        half = planted_fraction >= 0.5 and genome_len > span  # planted = the even reads: strided views, no index arrays
        rnd = slice(a + 1, b, 2) if half else a + np.nonzero(~pl)[0]
        n_rnd = len(range(a + 1, b, 2)) if half else len(rnd)
        for dst in ([m1, m2] if paired else [m1]):
            dst[rnd] = np.take(_ACGT256, np.frombuffer(r.bytes(n_rnd * read_len), dtype=np.uint8)).reshape(n_rnd, read_len)
        npl = int(pl.sum())
        if npl:
            which = r.integers(0, n_genomes, size=npl)
            pos = r.integers(0, genome_len - span, size=npl)
            sel = slice(a, b, 2) if half else a + np.nonzero(pl)[0]
            m1[sel] = win_fw[which * genome_len + pos]
            if paired:  # revcomp of G[e-read_len+1 .. e], e = pos+span-1  ==  rcG[L-1-e .. +read_len)
                m2[sel] = win_rc[which * genome_len + (genome_len - span - pos)]
            planted[sel] = which

    with cf.ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(work, enumerate(spans)))
    off1 = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    off2 = off1 + np.uint64(n_reads * read_len) if paired else None
    return DeviceFlatWorkload(name, bins, rows, hash_funs, k, w, rel_cutoff, read_len, n_reads, paired, planted_fraction,
                              seed, word_lo, row_words_total or W, bases, off1, off2, ACGT[genomes], gbins, planted)


def device_filter(hip, wl: DeviceFlatWorkload, device: int = 0, bin2target=None, n_targets=None):
    """allocate the filter on the device, fill it there, emplace the planted genomes' minimisers"""
    flt = hip.HipFilter.ibf(None, wl.bins, wl.rows, wl.hash_funs, bin2target, n_targets, device=device)
    flt.fill_random(wl.seed, 1, wl.word_lo, wl.row_words_total)
    n = plant_genomes(flt, wl)
    return flt, n


def sampled_oracle_ibf(flt, wl):
    """oracle.SampledIbf whose rows come from the device filter (gn_filter_download_row_list)"""
    import oracle
    return oracle.SampledIbf(wl.bins, wl.rows, wl.hash_funs, lambda idx: flt.download_row_list(idx, wl.bin_words))


def oracle_read_matches(ibf, wl, r: int, bins_per_target: int = 1):
    """(n_hashes, [(target, count)]) of read / pair r per GanonClassify.cpp:690-735; target t owns bins
    [t*bins_per_target, (t+1)*bins_per_target) (1 = identity map)"""
    import oracle
    s1 = wl.bases[int(wl.off[r]):int(wl.off[r + 1])]
    hh = oracle.minimiser_hash(oracle.to_ranks(s1), wl.k, wl.w)
    if getattr(wl, "off2", None) is not None:
        s2 = wl.bases[int(wl.off2[r]):int(wl.off2[r + 1])]
        if len(s2) >= wl.w:
            hh = np.concatenate([hh, oracle.minimiser_hash(oracle.to_ranks(s2), wl.k, wl.w)])
    counts = ibf.bulk_count(hh).astype(np.int64)
    if bins_per_target > 1:
        counts = counts.reshape(-1, bins_per_target).sum(axis=1)  # :516-523
    counts = np.minimum(counts, len(hh))                           # :525-526
    thr = oracle.threshold_cutoff(len(hh), wl.rel_cutoff)
    return len(hh), [(int(t), int(counts[t])) for t in np.nonzero(counts >= thr)[0]]


def checksum_matches(matches: np.ndarray) -> int:
    """order-independent checksum of (read, target, count) records; same formula as gno_baseline_classify."""
    if len(matches) == 0:
        return 0
    r = matches["read"].astype(np.uint64) + np.uint64(1)
    t = matches["target"].astype(np.uint64)
    c = matches["count"].astype(np.uint64)
    with np.errstate(over="ignore"):
        v = r * np.uint64(0x9E3779B97F4A7C15) + t * np.uint64(1000003) + c
        return int(np.sum(v, dtype=np.uint64))


# ---------------------------------------------------------------------------------------------------------------
# numpy restatement of the IBF row hash (only used to build synthetic HIBF workloads on the host; the flat
# workloads plant their genomes through the device path instead)
# ---------------------------------------------------------------------------------------------------------------
_IBF_SEEDS = np.array([13572355802537770549, 13043817825332782213, 10650232656628343401, 16499269484942379435,
                       4893150838803335377], dtype=np.uint64)


def np_ibf_row(v: np.ndarray, i: int, rows: int) -> np.ndarray:
    """row index of hash function i for uint64 values v; 64x64->high-64 multiply in 32-bit limbs"""
    shift = np.uint64(64 - int(rows).bit_length())
    with np.errstate(over="ignore"):
        x = v * _IBF_SEEDS[i]
        x ^= x >> shift
        x = x * np.uint64(11400714819323198485)
        S = np.uint64(rows)
        m32 = np.uint64(0xFFFFFFFF)
        xl, xh = x & m32, x >> np.uint64(32)
        sl, sh = S & m32, S >> np.uint64(32)
        ll, lh, hl, hh = xl * sl, xl * sh, xh * sl, xh * sh
        mid = (ll >> np.uint64(32)) + (lh & m32) + (hl & m32)
        return hh + (lh >> np.uint64(32)) + (hl >> np.uint64(32)) + (mid >> np.uint64(32))


def np_emplace(rows_arr: np.ndarray, hashes: np.ndarray, bins: np.ndarray, hash_funs: int) -> None:
    S, W = rows_arr.shape
    for i in range(hash_funs):
        r = np_ibf_row(hashes, i, S).astype(np.int64)
        np.bitwise_or.at(rows_arr, (r, (bins >> 6).astype(np.int64)), np.uint64(1) << (bins & 63).astype(np.uint64))


@dataclass
class HibfWorkload:
    name: str
    k: int
    w: int
    rel_cutoff: float
    read_len: int
    n_reads: int
    ibfs: list               # [(rows uint64[S*W], bins, S, h)]
    next_ibf_id: list
    bin_to_user: list
    n_user_bins: int
    bases: np.ndarray
    off: np.ndarray
    filter_bytes: int


def make_hibf_workload(hip, name: str, n_user_bins: int, tmax: int, rows_top: int, rows_child: int, hash_funs: int,
                       n_reads: int, read_len: int = 150, k: int = 19, w: int = 31, rel_cutoff: float = 0.75,
                       planted_fraction: float = 0.5, genome_len: int = 3000, n_genomes: int = 4096, seed: int = 42,
                       shard: int = 0) -> HibfWorkload:
    """2-level HIBF as `raptor layout` makes it for tmax = sqrt(user bins) (src/ganon/build_update.py:487): a top IBF
    of `tmax` merged bins, each pointing to a child IBF of n_user_bins/tmax leaf bins.  Bit matrices are
    Bernoulli(0.5); the minimisers of `n_genomes` genomes (hashed on the device through gn_stream_minimisers) are
    OR-ed into their leaf bin and into the merged bin above it."""
    per_child = n_user_bins // tmax
    assert per_child * tmax == n_user_bins
    threads = min(32, os.cpu_count() or 1)
    Wt, Wc = (tmax + 63) >> 6, (per_child + 63) >> 6
    top = np.empty((rows_top, Wt), dtype=np.uint64)
    _fill_random_u64(top, seed, threads)
    children = []
    for c in range(tmax):
        a = np.empty((rows_child, Wc), dtype=np.uint64)
        _fill_random_u64(a, seed + 1 + c, threads)
        children.append(a)

    rng = np.random.default_rng([seed, 1])
    genomes = rng.integers(0, 4, size=(n_genomes, genome_len), dtype=np.uint8)
    g_user = (np.arange(n_genomes, dtype=np.int64) * (n_user_bins // n_genomes)) % n_user_bins
    # hash the genomes on the device (any filter will do for a hash-only stream)
    tmpf = hip.HipFilter.ibf(None, 64, 64, 1)
    st = hip.HipStream(tmpf, n_genomes, n_genomes * genome_len)
    st.upload(ACGT[genomes].reshape(-1), np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(genome_len), None)
    st.minimisers(k, w)
    ho, hs = st.fetch_hashes()
    st.destroy()
    tmpf.free()
    cnt = np.diff(ho).astype(np.int64)
    ub = np.repeat(g_user, cnt)
    np_emplace(top, hs, (ub // per_child).astype(np.uint64), hash_funs)
    order = np.argsort(ub // per_child, kind="stable")
    hs_s, ub_s = hs[order], ub[order]
    bounds = np.searchsorted(ub_s // per_child, np.arange(tmax + 1))
    for c in range(tmax):
        a, b = bounds[c], bounds[c + 1]
        if b > a:
            np_emplace(children[c], hs_s[a:b], (ub_s[a:b] % per_child).astype(np.uint64), hash_funs)

    rrng = np.random.default_rng([seed, 2, shard])
    reads = rrng.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)
    n_pl = int(n_reads * planted_fraction)
    which = rrng.integers(0, n_genomes, size=n_pl)
    pos = rrng.integers(0, genome_len - read_len, size=n_pl)
    reads[np.arange(n_pl) * 2 if n_pl * 2 <= n_reads else np.arange(n_pl)] = genomes[which[:, None], pos[:, None] + np.arange(read_len)[None, :]]
    ibfs = [(top.reshape(-1), tmax, rows_top, hash_funs)] + [(a.reshape(-1), per_child, rows_child, hash_funs) for a in children]
    next_ids = [np.arange(1, tmax + 1, dtype=np.int64)] + [np.full(per_child, c + 1, dtype=np.int64) for c in range(tmax)]
    b2u = [np.full(tmax, -1, dtype=np.int64)] + [np.arange(c * per_child, (c + 1) * per_child, dtype=np.int64) for c in range(tmax)]
    fbytes = top.nbytes + sum(a.nbytes for a in children)
    return HibfWorkload(name, k, w, rel_cutoff, read_len, n_reads, ibfs, next_ids, b2u, n_user_bins, ACGT[reads].reshape(-1),
                        np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len), fbytes)


def make_hibf_device_workload(hip, name: str, n_user_bins: int, tmax: int, rows_top: int, rows_child: int, hash_funs: int,
                              n_reads: int, read_len: int = 150, k: int = 19, w: int = 31, rel_cutoff: float = 0.75,
                              planted_fraction: float = 0.5, genome_len: int = 3000, n_genomes: int = 4096, seed: int = 42,
                              shard: int = 0, device: int = 0, fill: int = 1):
    """`fill` = gn_filter_fill_random's and_words code: 1 = Bernoulli(1/2), hip.FILL_3_OF_16 = the density of an HIBF at the reference's
    defaults (--max-fp 0.001, four hash functions).
    Same 2-level HIBF as make_hibf_workload, but built on the device: every IBF is allocated empty, filled by
    gn_filter_fill_random (seed + ibf index) and the genomes' minimisers are emplaced with gn_filter_emplace_ibf.
    Returns (workload, filter); workload.ibfs holds (None, bins, rows, h) until download_hibf() fetches the bits."""
    per_child = n_user_bins // tmax
    assert per_child * tmax == n_user_bins
    rd = make_device_flat_workload(name, per_child, rows_child, hash_funs, n_reads, False, read_len, k, w, rel_cutoff,
                                   planted_fraction, genome_len, n_genomes, seed=seed, shard=shard)
    g_user = (np.arange(n_genomes, dtype=np.int64) * (n_user_bins // n_genomes)) % n_user_bins
    ibfs = [(None, tmax, rows_top, hash_funs)] + [(None, per_child, rows_child, hash_funs) for _ in range(tmax)]
    next_ids = [np.arange(1, tmax + 1, dtype=np.int64)] + [np.full(per_child, c + 1, dtype=np.int64) for c in range(tmax)]
    b2u = [np.full(tmax, -1, dtype=np.int64)] + [np.arange(c * per_child, (c + 1) * per_child, dtype=np.int64) for c in range(tmax)]
    flt = hip.HipFilter.hibf(ibfs, next_ids, b2u, n_user_bins, device=device)
    for i in range(len(ibfs)):
        flt.fill_random(seed + i, fill, ibf_idx=i)
    st = hip.HipStream(flt, n_genomes, n_genomes * genome_len)
    st.upload(rd.genomes.reshape(-1), np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(genome_len), None)
    st.minimisers(k, w)
    ho, hs = st.fetch_hashes()
    st.destroy()
    ub = np.repeat(g_user, np.diff(ho).astype(np.int64))
    flt.emplace(hs, (ub // per_child).astype(np.uint32), ibf_idx=0)
    order = np.argsort(ub // per_child, kind="stable")
    hs_s, ub_s = hs[order], ub[order]
    bounds = np.searchsorted(ub_s // per_child, np.arange(tmax + 1))
    for c in range(tmax):
        a, b = bounds[c], bounds[c + 1]
        if b > a:
            flt.emplace(hs_s[a:b], (ub_s[a:b] % per_child).astype(np.uint32), ibf_idx=c + 1)
    fbytes = sum(r * ((b + 63) >> 6) * 8 for (_, b, r, _) in ibfs)
    wl = HibfWorkload(name, k, w, rel_cutoff, read_len, n_reads, ibfs, next_ids, b2u, n_user_bins, rd.bases, rd.off, fbytes)
    wl.planted_genome = rd.planted_genome
    wl.genome_user_bin = g_user
    return wl, flt


def download_hibf(flt, wl: HibfWorkload) -> None:
    """fetch every IBF's bits from the device into wl.ibfs (for the CPU oracle)"""
    out = []
    for i, (_, bins, rows, h) in enumerate(wl.ibfs):
        W = (bins + 63) >> 6
        out.append((flt.download_rows(0, rows, W, ibf_idx=i).reshape(-1), bins, rows, h))
    wl.ibfs = out


# ---------------------------------------------------------------------------------------------------------------
# An HIBF that looks like raptor's output on real reference sets (bench.py workload hibf64k_skew): user-bin sizes are
# log-normal, so the layout has SPLIT user bins in the top level (the few huge ones), merged bins of very different
# cardinality (equal total size each), child IBFs of different widths (64 ... 1024 technical bins, not multiples of 64) and
# different numbers of rows, and a third level under the merged bins that hold more user bins than a child may have
# technical bins.  Bits are Bernoulli(3/8) -- 0.375^3 = 0.053, BASELINE.md's p^h ~ 0.05 for h = 3; the planted genomes'
# minimisers are emplaced on the device along their path (every merged bin above a user bin holds its content,
# hibf.hpp:124-136), a split user bin's content round-robin over its technical bins, and every fifth genome lives in a
# second user bin under another merged bin, so that a tenth of the reads descends into two children.
# ---------------------------------------------------------------------------------------------------------------
def skew_layout(n_user_bins: int, seed: int, top_bins: int = 512, n_top_split: int = 24, child_max: int = 1024, sigma: float = 1.6):
    """-> (ibfs [(bins, rows)], next_ibf_id, bin_to_user, paths {user bin: [(ibf, [technical bins])] from the top down}, summary)"""
    rng = np.random.default_rng([seed, 7])
    sizes = rng.lognormal(0.0, sigma, n_user_bins)
    order = np.argsort(-sizes, kind="stable")
    ibfs, nxt, b2u = [], [], []
    paths = {}

    def rows_for(largest: float, lo=1 << 18, hi=1 << 23) -> int:
        # an IBF is sized by its largest technical bin (odd numbers of rows on purpose: the row map is a 128-bit multiply, not a mask)
        return int(min(hi, max(lo, largest * 48001.0))) | 1

    def new_ibf():
        ibfs.append(None)
        nxt.append([])
        b2u.append([])
        return len(ibfs) - 1

    def leaf_or_deeper(members, above):
        """child IBF of the merged bin that holds `members` (user bins, descending size); `above` = path prefix of its user bins"""
        idx = new_ibf()
        m = len(members)
        if m <= child_max:
            for b, u in enumerate(members):
                nxt[idx].append(idx)
                b2u[idx].append(int(u))
                paths[int(u)] = above + [(idx, [b])]
            ibfs[idx] = (m, rows_for(sizes[members[0]]))
            return idx
        # more user bins than technical bins: the largest stay here, the tail goes under merged bins of this IBF (a third level)
        n_merged = int(np.ceil((m - child_max) / (child_max - 1))) + 1
        n_single = child_max - n_merged
        for b, u in enumerate(members[:n_single]):
            nxt[idx].append(idx)
            b2u[idx].append(int(u))
            paths[int(u)] = above + [(idx, [b])]
        tail = members[n_single:]
        cuts = np.linspace(0, len(tail), n_merged + 1).astype(int)
        for g in range(n_merged):
            b = n_single + g
            nxt[idx].append(-1)
            b2u[idx].append(-1)
            nxt[idx][b] = leaf_or_deeper(tail[cuts[g]:cuts[g + 1]], above + [(idx, [b])])
        ibfs[idx] = (child_max, rows_for(sizes[members[0]]))
        return idx

    top = new_ibf()
    split = order[:n_top_split]
    unit = sizes[order[n_top_split]]
    b = 0
    for u in split:
        k = int(min(8, max(2, np.ceil(sizes[u] / unit / 1.5))))
        for _ in range(k):
            nxt[top].append(top)
            b2u[top].append(int(u))
        paths[int(u)] = [(top, list(range(b, b + k)))]
        b += k
    rest = order[n_top_split:]
    n_merged = top_bins - b
    csum = np.cumsum(sizes[rest])
    cuts = np.searchsorted(csum, np.linspace(0, csum[-1], n_merged + 1)[1:-1])
    groups = np.split(rest, cuts)
    for g in groups:
        if len(g) == 0:
            continue
        bb = len(b2u[top])
        nxt[top].append(-1)
        b2u[top].append(-1)
        nxt[top][bb] = leaf_or_deeper(g, [(top, [bb])])
    ibfs[top] = (len(b2u[top]), (1 << 22) | 1)
    depth = max(len(p) for p in paths.values())
    widths = sorted(bn for bn, _ in ibfs[1:])
    summary = dict(ibfs=len(ibfs), depth=depth, top_bins=ibfs[top][0], top_split_user_bins=int(n_top_split), top_split_technical_bins=int(b),
                   child_bins_min=int(widths[0]), child_bins_median=int(widths[len(widths) // 2]), child_bins_max=int(widths[-1]),
                   rows_min=int(min(r for _, r in ibfs)), rows_max=int(max(r for _, r in ibfs)),
                   user_bins_at_depth={d: int(sum(1 for p in paths.values() if len(p) == d)) for d in range(1, depth + 1)})
    return ibfs, nxt, b2u, paths, summary


def make_hibf_skew_device_workload(hip, name: str, n_user_bins: int, hash_funs: int, n_reads: int, read_len: int = 150, k: int = 19, w: int = 31,
                                   rel_cutoff: float = 0.75, planted_fraction: float = 0.5, genome_len: int = 3000, n_genomes: int = 4096,
                                   seed: int = 42, shard: int = 0, device: int = 0, rows_scale: float = 1.0, fill: int = 0):
    """`fill`: gn_filter_fill_random's code, 0 = hip.FILL_3_OF_8 (p^3 = 0.053), hip.FILL_3_OF_16 for the reference's HIBF defaults (p^4 = 0.0012).
    -> (HibfWorkload, HipFilter); workload.layout = skew_layout's summary.  rows_scale < 1 shrinks every IBF (dry runs, tests)."""
    shapes, nxt, b2u, paths, summary = skew_layout(n_user_bins, seed)
    shapes = [(b, max(1031, int(r * rows_scale)) | 1) for b, r in shapes]
    rd = make_device_flat_workload(name, 64, 64, hash_funs, n_reads, False, read_len, k, w, rel_cutoff, planted_fraction, genome_len, n_genomes,
                                   seed=seed, shard=shard)
    ibfs = [(None, b, r, hash_funs) for b, r in shapes]
    next_ids = [np.asarray(a, dtype=np.int64) for a in nxt]
    bin_user = [np.asarray(a, dtype=np.int64) for a in b2u]
    flt = hip.HipFilter.hibf(ibfs, next_ids, bin_user, n_user_bins, device=device)
    fill = fill or hip.FILL_3_OF_8
    for i in range(len(ibfs)):
        flt.fill_random(seed + i, fill, ibf_idx=i)
    # genomes -> user bins: the first ones are the huge split user bins of the top level, the others spread over the rest; every fifth one
    # also lives in a second user bin (another strain of it) somewhere else in the tree
    grng = np.random.default_rng([seed, 8])
    split_users = [u for u, p in paths.items() if len(p) == 1]
    others = np.array([u for u in paths if len(paths[u]) > 1], dtype=np.int64)
    pick = grng.permutation(others)
    g_user = np.empty(n_genomes, dtype=np.int64)
    ns = min(len(split_users), n_genomes // 8)
    g_user[:ns] = split_users[:ns]
    g_user[ns:] = pick[: n_genomes - ns]
    second = {g: int(pick[n_genomes + i]) for i, g in enumerate(range(ns, n_genomes, 5))}
    st = hip.HipStream(flt, n_genomes, n_genomes * genome_len)
    st.upload(rd.genomes.reshape(-1), np.arange(n_genomes + 1, dtype=np.uint64) * np.uint64(genome_len), None)
    st.minimisers(k, w)
    ho, hs = st.fetch_hashes()
    st.destroy()
    per_ibf = {}
    for g in range(n_genomes):
        hv = hs[int(ho[g]):int(ho[g + 1])]
        for u in [int(g_user[g])] + ([second[g]] if g in second else []):
            for ibf_idx, tb in paths[u]:
                bins = np.asarray(tb, dtype=np.uint32)[np.arange(len(hv)) % len(tb)]   # a split user bin: round-robin over its technical bins
                per_ibf.setdefault(ibf_idx, []).append((hv, bins))
    for ibf_idx, lst in per_ibf.items():
        flt.emplace(np.concatenate([a for a, _ in lst]), np.concatenate([b for _, b in lst]).astype(np.uint32), ibf_idx=ibf_idx)
    fbytes = sum(r * ((b + 63) >> 6) * 8 for (_, b, r, _) in ibfs)
    wl = HibfWorkload(name, k, w, rel_cutoff, read_len, n_reads, ibfs, next_ids, bin_user, n_user_bins, rd.bases, rd.off, fbytes)
    wl.planted_genome = rd.planted_genome
    wl.genome_user_bin = g_user
    wl.genome_second_user_bin = second
    wl.layout = dict(summary, filter_gib=round(fbytes / 2**30, 2), genomes_in_two_user_bins=len(second),
                     fill="Bernoulli(3/16)" if fill == hip.FILL_3_OF_16 else "Bernoulli(3/8)")
    return wl, flt
