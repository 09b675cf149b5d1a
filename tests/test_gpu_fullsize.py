"""BASELINE.json configs[1] at FULL size (8 GiB flat IBF, 4096 bins, h=4, 10 M reads of 150 bp) through the C ABI:
size-independent properties + an oracle check on a random sample of reads against the device's own filter bits.
Set GANON_FULLSIZE_ROWS / GANON_FULLSIZE_READS to shrink it for a quick run."""
import os

import numpy as np

import gpu_util as gu
import pytest

import bench_workload as bw
import oracle

pytestmark = pytest.mark.gpu

ROWS = int(os.environ.get("GANON_FULLSIZE_ROWS", 1 << 24))
READS = int(os.environ.get("GANON_FULLSIZE_READS", 10_000_000))


@pytest.fixture(scope="module")
def full():
    import ganon_amd
    wl = bw.make_flat_workload("full", 4096, ROWS, 4, READS, seed=1234)
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs)
    bw.plant_genomes(flt, wl)
    st = ganon_amd.HipStream(flt, READS, wl.bases.size, READS * 2)
    st.upload(wl.bases, wl.off, None)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    out = st.fetch()
    yield ganon_amd, wl, flt, st, out
    st.destroy()
    flt.free()


def test_structure(full):
    hip, wl, flt, st, (nh, status, mo, m) = full
    assert flt.info()["device_bytes"] == ROWS * 512
    assert (status == 0).all() and nh.min() >= 1 and nh.max() <= 120
    assert int(nh.sum(dtype=np.uint64)) == st.timings()["n_hashes"]
    assert st.timings()["algo_bytes"] == int(nh.sum(dtype=np.uint64)) * 4 * 512
    assert mo[0] == 0 and mo[-1] == len(m) and (np.diff(mo.astype(np.int64)) >= 0).all()
    key = m["read"].astype(np.uint64) << np.uint64(32) | m["target"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()          # grouped by read, ascending target, no duplicates
    assert (m["count"] <= nh[m["read"]]).all()                 # capped at n_hashes (GanonClassify.cpp:525-526)
    thr = np.maximum(1, np.ceil(nh[m["read"]].astype(np.float64) * wl.rel_cutoff)).astype(np.uint32)
    assert (m["count"] >= thr).all()                           # every reported match reaches the read's cutoff


def test_planted_reads_are_found(full):
    # reads cut from a planted genome must report the genome's bin with count == n_hashes (no false negatives)
    hip, wl, flt, st, (nh, status, mo, m) = full
    n_pl = int(wl.n_reads * wl.planted_fraction)
    rrng = np.random.default_rng([wl.seed, 2, 0])
    rrng.integers(0, 4, size=(wl.n_reads, wl.read_len), dtype=np.uint8)    # replay the generator of make_flat_workload
    which = rrng.integers(0, len(wl.genome_bins), size=n_pl)
    sel = np.arange(n_pl) * 2
    first = mo[sel].astype(np.int64)
    assert (mo[sel + 1].astype(np.int64) > first).all()
    # the genome's bin is among the read's matches with the full count
    ok = np.zeros(n_pl, dtype=bool)
    for off in range(0, 4):
        idx = np.minimum(first + off, len(m) - 1)
        hit = (m["read"][idx] == sel) & (m["target"][idx] == wl.genome_bins[which]) & (m["count"][idx] == nh[sel])
        ok |= hit
    assert ok.mean() > 0.9999, ok.mean()


def test_idempotent_and_order_independent(full):
    hip, wl, flt, st, (nh, status, mo, m) = full
    ck = bw.checksum_matches(m)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh2, status2, mo2, m2 = st.fetch()
    assert np.array_equal(nh, nh2) and np.array_equal(mo, mo2) and np.array_equal(m, m2)
    assert bw.checksum_matches(m2) == ck
    # a shuffled sub-batch gives the same per-read answers (reads are independent, GanonClassify.cpp:676-831)
    rng = np.random.default_rng(3)
    pick = rng.choice(wl.n_reads, size=200_000, replace=False)
    reads = wl.bases.reshape(wl.n_reads, wl.read_len)[pick]
    st2 = hip.HipStream(flt, len(pick), reads.size)
    st2.submit(reads.reshape(-1), np.arange(len(pick) + 1, dtype=np.uint64) * np.uint64(wl.read_len), None, wl.k, wl.w, wl.rel_cutoff)
    nh3, _, mo3, m3 = st2.fetch()
    assert np.array_equal(nh3, nh[pick])
    assert np.array_equal(np.diff(mo3.astype(np.int64)), np.diff(mo.astype(np.int64))[pick])
    j = rng.integers(0, len(pick), size=2000)
    for x in j.tolist():
        a = m3[int(mo3[x]):int(mo3[x + 1])]
        b = m[int(mo[pick[x]]):int(mo[pick[x] + 1])]
        assert np.array_equal(a["target"], b["target"]) and np.array_equal(a["count"], b["count"])
    st2.destroy()


def test_sample_against_oracle(full):
    hip, wl, flt, st, (nh, status, mo, m) = full
    bw.download_filter(flt, wl)
    _, ibf = bw.oracle_filter(wl)
    rng = np.random.default_rng(77)
    for r in np.unique(rng.integers(0, wl.n_reads, size=3000)).tolist():
        seq = wl.bases[int(wl.off[r]):int(wl.off[r + 1])]
        hh = oracle.minimiser_hash(oracle.to_ranks(seq), wl.k, wl.w)
        counts = np.minimum(ibf.bulk_count(hh).astype(np.int64), len(hh))
        thr = oracle.threshold_cutoff(len(hh), wl.rel_cutoff)
        exp = [(int(t), int(counts[t])) for t in np.nonzero(counts >= thr)[0]]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[r]):int(mo[r + 1])]]
        assert nh[r] == len(hh) and got == exp, (r, got, exp)


def test_filter_matches_prepass_at_full_size(full, monkeypatch):
    # the device pre-pass of filter_matches at a low cutoff (--rel-cutoff 0.2 --rel-filter 0.1 --fpr-query 1e-5: ~100 chance matches
    # per read) on all 10 M reads: nothing is lost or invented (dropped + survivors = matches of the plain run), the count kernel's
    # pre-drop changes no survivor, mark, maximum or total, and on a sample of reads the survivors are what the oracle's
    # filter_matches leaves of the unfiltered matches
    from test_gpu_parity import _exact_filter_matches
    hip, wl, flt, st, _ = full
    tfpr = np.full(wl.bins, 0.5 ** wl.hash_funs)
    res = {}
    for tag in ("predrop", "plain"):
        if tag == "plain":
            gu.SW.on("predrop")
        st.set_postfilter(0.1, 1e-5, tfpr)
        st.classify(wl.k, wl.w, 0.2)
        nh, status, mo, m = st.fetch()
        mx, d_fil, d_fpr = st.fetch_postfilter()
        res[tag] = (mo, m, mx, d_fil, d_fpr)
        gu.SW.off("predrop")
    mo, m, mx, d_fil, d_fpr = res["predrop"]
    assert np.array_equal(mo, res["plain"][0]) and np.array_equal(m, res["plain"][1]) and np.array_equal(mx, res["plain"][2])
    assert (d_fil, d_fpr) == res["plain"][3:]
    st.set_postfilter(None)
    st.classify(wl.k, wl.w, 0.2)
    st.sync()
    raw_n = st.timings()["n_matches"]
    assert raw_n > 50 * wl.n_reads * (READS >= 1_000_000) and d_fil + d_fpr + len(m) == raw_n
    # a sub-batch through a second stream without the pass gives the unfiltered matches of its reads
    rng = np.random.default_rng(5)
    pick = rng.choice(wl.n_reads, size=min(50_000, wl.n_reads), replace=False)
    reads = wl.bases.reshape(wl.n_reads, wl.read_len)[pick]
    st2 = hip.HipStream(flt, len(pick), reads.size)
    st2.submit(reads.reshape(-1), np.arange(len(pick) + 1, dtype=np.uint64) * np.uint64(wl.read_len), None, wl.k, wl.w, 0.2)
    nh3, _, mo3, m3 = st2.fetch()
    for x in rng.integers(0, len(pick), size=1500).tolist():
        r = int(pick[x])
        raw = [(int(e["target"]), int(e["count"])) for e in m3[int(mo3[x]):int(mo3[x + 1])]]
        kept, nf, nq, emx = _exact_filter_matches(raw, nh3[x], 0.1, 1e-5, tfpr)
        got = [(int(e["target"]), int(e["count"]) & 0x7FFFFFFF) for e in m[int(mo[r]):int(mo[r + 1])]]
        assert int(mx[r]) == emx and set(kept) <= set(got) <= set(raw), r
        exact_drop = set(raw) - set(kept)
        assert [e for e in got if e not in exact_drop] == kept, r  # (the host's exact --fpr-query check on the survivors ends there)
    st2.destroy()
    st.classify(wl.k, wl.w, wl.rel_cutoff)  # (the fixture's batch again)
    st.sync()


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: 2-level HIBF, 65 536 user bins (top IBF of 256 merged bins -> 256 children of 256 bins)
# ---------------------------------------------------------------------------------------------------------------
HIBF_READS = int(os.environ.get("GANON_FULLSIZE_HIBF_READS", 10_000_000))  # BASELINE states 10 M
HIBF_ROWS = int(os.environ.get("GANON_FULLSIZE_HIBF_ROWS", 1 << 20))   # 257 IBFs x 2^20 rows x 32 B = 8 GiB


@pytest.fixture(scope="module")
def hibf_full():
    import ganon_amd
    # built on the device: every IBF allocated empty, seeded fill, genomes emplaced per IBF (gn_filter_emplace_ibf)
    wl, flt = bw.make_hibf_device_workload(ganon_amd, "hibf64k", 65536, 256, HIBF_ROWS, HIBF_ROWS, 3, HIBF_READS, seed=99)
    st = ganon_amd.HipStream(flt, HIBF_READS, wl.bases.size, HIBF_READS * 2)
    st.upload(wl.bases, wl.off, None)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    out = st.fetch()
    yield ganon_amd, wl, flt, st, out
    st.destroy()
    flt.free()


def test_hibf_fullsize_structure_and_oracle_sample(hibf_full):
    hip, wl, flt, st, (nh, status, mo, m) = hibf_full
    info = flt.info()
    assert info["is_hibf"] and info["n_ibf"] == 257 and info["n_targets"] == 65536
    assert mo[-1] == len(m) and (np.diff(mo.astype(np.int64)) >= 0).all()
    key = m["read"].astype(np.uint64) << np.uint64(32) | m["target"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()
    assert (m["count"] <= nh[m["read"]]).all() and (m["target"] < 65536).all()
    # every planted read reports its genome's user bin with the full count; algorithmic bytes >= the top-level visit
    pl = np.nonzero(wl.planted_genome >= 0)[0]
    assert len(pl) == (wl.n_reads + 1) // 2 and (np.diff(mo.astype(np.int64))[pl] >= 1).all()
    want = wl.genome_user_bin[wl.planted_genome[pl]]
    found = np.zeros(len(pl), dtype=bool)
    first, cnt = mo[pl].astype(np.int64), np.diff(mo.astype(np.int64))[pl]
    for off in range(int(min(cnt.max(), 6))):
        idx = np.minimum(first + off, len(m) - 1)
        found |= (off < cnt) & (m["target"][idx] == want) & (m["count"][idx] == nh[pl])
    assert found.mean() > 0.9999, found.mean()
    tm = st.timings()
    assert tm["algo_bytes"] >= int(nh.sum(dtype=np.uint64)) * 3 * 32
    # idempotent
    ck = bw.checksum_matches(m)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    assert bw.checksum_matches(st.fetch()[3]) == ck
    # oracle HIBF on the device's own bits for a random sample
    bw.download_hibf(flt, wl)
    ibfs = [oracle.Ibf(b, s, h, r) for (r, b, s, h) in wl.ibfs]
    hb = oracle.Hibf(ibfs, wl.next_ibf_id, wl.bin_to_user, wl.n_user_bins)
    rng = np.random.default_rng(5)
    algo = 0
    for r in np.unique(rng.integers(0, wl.n_reads, size=600)).tolist():
        seq = wl.bases[int(wl.off[r]):int(wl.off[r + 1])]
        hh = oracle.minimiser_hash(oracle.to_ranks(seq), wl.k, wl.w)
        thr = oracle.threshold_cutoff(len(hh), wl.rel_cutoff)
        counts = hb.bulk_count(hh, thr)
        exp = [(int(u), int(min(c, len(hh)))) for u, c in zip(np.nonzero(counts)[0], counts[np.nonzero(counts)[0]])]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[r]):int(mo[r + 1])]]
        assert nh[r] == len(hh) and got == exp, (r, got, exp)


def test_hibf_low_cutoff_at_full_size_runs_in_read_ranges(hibf_full, monkeypatch):
    """configs[2] at the binary's own --rel-cutoff 0.2 with the pre-pass `ganon classify` sets (--rel-filter 0.1 --fpr-query 1e-5):
    ~2700 chance pairs per read, 27 G raw pairs for the 10 M reads -- more than one radix sort counts and more than the device
    holds.  The batch is run in read ranges (gn_hibf.hip); whatever the range size, survivors and dropped totals are the same,
    and a sample of reads agrees with the oracle's counting agent + filter_matches."""
    import oracle
    hip, wl, flt, st, _ = hibf_full
    tfpr = np.full(65536, 0.05)
    runs = []
    for limit in (None, "600000000"):
        if limit:
            gu.SW.on(f"hibf_pair_limit={limit}")
        st.set_postfilter(0.1, 1e-5, tfpr)
        st.classify(wl.k, wl.w, 0.2)
        nh, status, mo, m = st.fetch()
        mx, d_fil, d_fpr = st.fetch_postfilter()
        runs.append((mo.copy(), bw.checksum_matches(m), len(m), d_fil, d_fpr, mx.copy(), m if limit is None else None))
    st.set_postfilter(None)
    a, b = runs
    assert np.array_equal(a[0], b[0]) and a[1:5] == b[1:5] and np.array_equal(a[5], b[5])
    raw_pairs = a[2] + a[3] + a[4]
    assert raw_pairs > 2_000 * (HIBF_READS // 1000) and raw_pairs > 2**31 or HIBF_READS < 10_000_000
    # oracle on a few reads: every pair that passed the cutoff, then filter_matches' two rules
    bw.download_hibf(flt, wl)
    hb = oracle.Hibf([oracle.Ibf(bn, s_, h, r) for (r, bn, s_, h) in wl.ibfs], wl.next_ibf_id, wl.bin_to_user, wl.n_user_bins)
    rng = np.random.default_rng(8)
    mo, m = a[0], a[6]
    for r in rng.integers(0, wl.n_reads, size=12).tolist():
        hh = oracle.minimiser_hash(oracle.to_ranks(wl.bases[int(wl.off[r]):int(wl.off[r + 1])]), wl.k, wl.w)
        counts = np.minimum(hb.bulk_count(hh, oracle.threshold_cutoff(len(hh), 0.2)).astype(np.int64), len(hh))
        nz = np.nonzero(counts)[0]
        mx_, mn_ = int(counts[nz].max()), int(min(len(hh), counts[nz].min()))
        thr = mx_ - int(np.ceil((mx_ - mn_) * 0.1))
        keep = {int(u) for u in nz if counts[u] >= thr}
        got = {int(x["target"]): int(x["count"]) & 0x7FFFFFFF for x in m[int(mo[r]):int(mo[r + 1])]}
        assert set(got) <= keep and all(got[u] == counts[u] for u in got), r      # survivors of --rel-filter, minus what --fpr-query took
        assert int(a[5][r]) == mx_


# ---------------------------------------------------------------------------------------------------------------
# The same 65 536 user bins in a layout like raptor's on log-normal user-bin sizes (bench.py: hibf64k_skew): split user bins in the
# top level, merged bins of different cardinality, children of 2 ... 1024 technical bins with different numbers of rows, three
# levels; a tenth of the reads descends into two children.  Every read of a sample == the oracle's counting agent (hibf.hpp:432-460).
# ---------------------------------------------------------------------------------------------------------------
def _check_against_oracle_hibf(wl, flt, nh, mo, m, sample, seed):
    bw.download_hibf(flt, wl)
    hb = oracle.Hibf([oracle.Ibf(b, s, h, r) for (r, b, s, h) in wl.ibfs], wl.next_ibf_id, wl.bin_to_user, wl.n_user_bins)
    rng = np.random.default_rng(seed)
    multi = 0
    for r in np.unique(rng.integers(0, wl.n_reads, size=sample)).tolist():
        seq = wl.bases[int(wl.off[r]):int(wl.off[r + 1])]
        hh = oracle.minimiser_hash(oracle.to_ranks(seq), wl.k, wl.w)
        thr = oracle.threshold_cutoff(len(hh), wl.rel_cutoff)
        counts = hb.bulk_count(hh, thr)
        exp = [(int(u), int(min(c, len(hh)))) for u, c in zip(np.nonzero(counts)[0], counts[np.nonzero(counts)[0]])]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[r]):int(mo[r + 1])]]
        assert nh[r] == len(hh) and got == exp, (r, got, exp)
        multi += len(exp) >= 2
    return multi


@pytest.mark.parametrize("rel_cutoff", [0.75, 0.3])
def test_hibf_skewed_layout_small_equals_oracle(rel_cutoff):
    import ganon_amd
    wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, "skew_small", 16384, 3, 40_000, rel_cutoff=rel_cutoff, seed=7, rows_scale=0.004)
    lay = wl.layout
    assert lay["depth"] == 3 and lay["top_split_technical_bins"] > lay["top_split_user_bins"] and lay["child_bins_max"] == 1024
    st = ganon_amd.HipStream(flt, wl.n_reads, wl.bases.size, wl.n_reads * 8)
    st.upload(wl.bases, wl.off, None)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh, status, mo, m = st.fetch()
    multi = _check_against_oracle_hibf(wl, flt, nh, mo, m, 3000, 11)
    assert multi > 30           # reads that matched in two user bins (two children) were among the sample
    # the same batch through the other kernel paths: no packed kernel, LDS kernel only, no sorting of a level's queue by IBF width
    for switch in ("hibf_pack", "hibf_reg", "hibf_one_pack"):
        gu.SW.on(switch)
        try:
            st.classify(wl.k, wl.w, wl.rel_cutoff)
            nh2, _, mo2, m2 = st.fetch()
        finally:
            gu.SW.off(switch)
        assert np.array_equal(nh, nh2) and np.array_equal(mo, mo2) and np.array_equal(m, m2), switch
    # dense user-bin counts of a few reads == the agent's result vector
    dense = st.dense_counts(0, 64, wl.n_user_bins)
    hb = oracle.Hibf([oracle.Ibf(b, s, h, r) for (r, b, s, h) in wl.ibfs], wl.next_ibf_id, wl.bin_to_user, wl.n_user_bins)
    for r in range(64):
        hh = oracle.minimiser_hash(oracle.to_ranks(wl.bases[int(wl.off[r]):int(wl.off[r + 1])]), wl.k, wl.w)
        assert np.array_equal(dense[r], hb.bulk_count(hh, oracle.threshold_cutoff(len(hh), wl.rel_cutoff)))
    st.destroy()
    flt.free()


def test_hibf_skewed_layout_fullsize():
    import ganon_amd
    n = int(os.environ.get("GANON_FULLSIZE_HIBF_READS", 10_000_000))
    wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, "hibf64k_skew", 65536, 3, n, seed=99)
    assert flt.info()["n_targets"] == 65536 and wl.layout["depth"] == 3
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2)
    st.upload(wl.bases, wl.off, None)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh, status, mo, m = st.fetch()
    cnt = np.diff(mo.astype(np.int64))
    pl = np.nonzero(wl.planted_genome >= 0)[0]
    # every planted read reports its genome's user bin(s); the genomes that live in two user bins give two matches
    assert (cnt[pl] >= 1).all()
    two = np.isin(wl.planted_genome[pl], np.fromiter(wl.genome_second_user_bin.keys(), dtype=np.int64))
    assert (cnt[pl][two] >= 2).mean() > 0.999 and 0.07 < two.mean() * len(pl) / n < 0.13
    levels = st.hibf_levels()
    assert len(levels) == 3 and all(lv["ms"] > 0 for lv in levels[:2])
    # the top level's rows are 64 bytes (512 bins): a line each; the lower levels mix widths: lines >= rows, a multiple of 128
    assert levels[0]["line_bytes"] == 2 * levels[0]["algo_bytes"] and all(lv["line_bytes"] >= lv["algo_bytes"] and lv["line_bytes"] % 128 == 0 for lv in levels)
    ck = bw.checksum_matches(m)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    assert bw.checksum_matches(st.fetch()[3]) == ck
    _check_against_oracle_hibf(wl, flt, nh, mo, m, 600, 5)
    st.destroy()
    flt.free()


# ---------------------------------------------------------------------------------------------------------------
# configs[2] at the REFERENCE'S HIBF defaults (bench.py: hibf64k_p001 / hibf64k_skew_p001): `ganon build --filter-type hibf` hands raptor
# --max-fp 0.001 and four hash functions (/root/reference/src/ganon/config.py:140-143,1258-1260, build_update.py:487-489) -> h = 4, bits
# Bernoulli(3/16) (0.1875^4 = 0.0012), both layouts, at --rel-cutoff 0.75 and at the binary's 0.2 under the wrapper's filter rules.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["uniform", "skew"])
def test_hibf_at_the_references_defaults_fullsize(layout):
    import ganon_amd
    n = int(os.environ.get("GANON_FULLSIZE_HIBF_READS", 10_000_000))
    if layout == "skew":
        wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, "hibf64k_skew_p001", 65536, 4, n, seed=99, fill=ganon_amd.FILL_3_OF_16)
        assert wl.layout["fill"] == "Bernoulli(3/16)"
    else:
        wl, flt = bw.make_hibf_device_workload(ganon_amd, "hibf64k_p001", 65536, 256, HIBF_ROWS, HIBF_ROWS, 4, n, seed=99, fill=ganon_amd.FILL_3_OF_16)
    assert flt.info()["n_targets"] == 65536
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2)
    st.upload(wl.bases, wl.off, None)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh, status, mo, m = st.fetch()
    cnt = np.diff(mo.astype(np.int64))
    pl = np.nonzero(wl.planted_genome >= 0)[0]
    assert (cnt[pl] >= 1).all()
    # at p^h = 0.0012 a random read matches nothing at 0.75: the matches are the planted ones (one or, for a genome in two user bins, two)
    not_pl = np.nonzero(wl.planted_genome < 0)[0]
    assert cnt[not_pl].sum() == 0 and cnt[pl].max() <= 3
    ck = bw.checksum_matches(m)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    assert bw.checksum_matches(st.fetch()[3]) == ck
    _check_against_oracle_hibf(wl, flt, nh, mo, m, 600, 5)
    # the binary's own cutoff 0.2 (T ~ 4 of 17.5 minimisers): plain, then with the device pre-pass of filter_matches under the wrapper's rules
    # (--rel-filter 0.1 --fpr-query 1e-5, user-bin fpr 0.001).  The raw pair count stays within a few per read -- at h = 3 / p^h = 0.05 it
    # was 2 700 per read -- so one pass sorts it; dropped + survivors = the plain run's matches; the switches change nothing.
    wl.rel_cutoff = 0.2
    st.classify(wl.k, wl.w, 0.2)
    nh2, _, mo2, m2 = st.fetch()
    assert len(m2) < 4 * n and len(m2) >= len(m)
    _check_against_oracle_hibf(wl, flt, nh2, mo2, m2, 600, 6)
    st.set_postfilter(0.1, 1e-5, np.full(65536, 0.001))
    res = []
    for sw in (None, "predrop", "hibf_pack"):
        if sw:
            gu.SW.on(sw)
        st.classify(wl.k, wl.w, 0.2)
        _, _, mo3, m3 = st.fetch()
        mx, d_fil, d_fpr = st.fetch_postfilter()
        res.append((bw.checksum_matches(m3), len(m3), d_fil, d_fpr, mo3.copy(), mx.copy()))
        if sw:
            gu.SW.off(sw)
    st.set_postfilter(None)
    assert res[0][1] + res[0][2] + res[0][3] == len(m2)
    for r in res[1:]:
        assert r[:4] == res[0][:4] and np.array_equal(r[4], res[0][4]) and np.array_equal(r[5], res[0][5])
    # per-level line bytes are reported (bench.py prints them against the gather roof)
    levels = st.hibf_levels()
    assert len(levels) >= 2 and all(lv["line_bytes"] >= lv["algo_bytes"] > 0 for lv in levels[:2])
    st.destroy()
    flt.free()
