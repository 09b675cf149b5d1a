"""BASELINE.json configs[3] and configs[4] at FULL size on one MI355X, through the C ABI:

  configs[3]  128 GiB flat IBF (32 768 technical bins = 4 KiB rows, 2^25 rows, h=4), 12.5 M pairs of 2x150 bp --
              one GPU's shard of the 100 M-pair job (every GPU of the 8 holds the same replica)
  configs[4]  one rank's 128 GiB column slice (bins 163 840..196 607 of 262 144) of a 1 TiB flat IBF, every pair
              classified against it, sparse matches sent to the read's owner through ganon_amd.partition over RCCL

Neither filter ever exists on the host: it is generated on the device (gn_filter_fill_random, seeded, position-keyed),
the planted genomes are emplaced on the device, and the oracle fetches exactly the rows its read sample touches
(gn_filter_download_row_list -> oracle.SampledIbf).  Checks: size-independent properties over the whole batch +
an oracle comparison on >= 3000 random pairs.  GANON_LARGE_ROWS / GANON_LARGE_PAIRS shrink it for a quick run."""
import os
import socket

import numpy as np

import gpu_util as gu
import pytest

import bench_workload as bw

pytestmark = pytest.mark.gpu

ROWS = int(os.environ.get("GANON_LARGE_ROWS", 1 << 25))
PAIRS = int(os.environ.get("GANON_LARGE_PAIRS", 12_500_000))
BINS = 32768
SLICES, SLICE = 8, 5


def _properties(wl, nh, status, mo, m, target_lo):
    assert (status == 0).all() and nh.min() >= 2 and nh.max() <= 127
    assert mo[0] == 0 and mo[-1] == len(m) and (np.diff(mo.astype(np.int64)) >= 0).all()
    key = m["read"].astype(np.uint64) << np.uint64(32) | m["target"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()          # grouped by read, ascending target, no duplicates
    assert (m["target"] >= target_lo).all() and (m["target"] < target_lo + wl.bins).all()
    assert (m["count"] <= nh[m["read"]]).all()                 # capped at n_hashes (GanonClassify.cpp:525-526)
    thr = np.maximum(1, np.ceil(nh[m["read"]].astype(np.float64) * wl.rel_cutoff)).astype(np.uint32)
    assert (m["count"] >= thr).all()                           # every reported match reaches the pair's cutoff
    # every pair cut from a planted genome reports the genome's bin with count == n_hashes (no false negatives)
    pl = np.nonzero(wl.planted_genome >= 0)[0]
    want = wl.genome_bins[wl.planted_genome[pl]].astype(np.int64) + target_lo
    found = np.zeros(len(pl), dtype=bool)
    first = mo[pl].astype(np.int64)
    cnt = mo[pl + 1].astype(np.int64) - first
    for off in range(int(min(cnt.max(), 6))):
        idx = np.minimum(first + off, len(m) - 1)
        found |= (off < cnt) & (m["target"][idx] == want) & (m["count"][idx] == nh[pl])
    assert found.mean() > 0.9999, found.mean()


def _oracle_sample(wl, flt, nh, mo, m, target_lo, n_sample=3000, seed=77):
    ibf = bw.sampled_oracle_ibf(flt, wl)
    rng = np.random.default_rng(seed)
    n_true = 0
    for r in np.unique(rng.integers(0, wl.n_reads, size=n_sample)).tolist():
        n_h, exp = bw.oracle_read_matches(ibf, wl, r)
        exp = [(t + target_lo, c) for t, c in exp]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[r]):int(mo[r + 1])]]
        assert nh[r] == n_h and got == exp, (r, got, exp)
        n_true += len(exp)
    assert n_true > n_sample // 4


# ----------------------------------------------------------------------------------------------- configs[3]
@pytest.fixture(scope="module")
def flat128g():
    import ganon_amd
    wl = bw.make_device_flat_workload("flat128g", BINS, ROWS, 4, PAIRS, paired=True, seed=4321)
    flt, _ = bw.device_filter(ganon_amd, wl)
    st = ganon_amd.HipStream(flt, PAIRS, wl.bases.size, PAIRS * 2)
    st.upload(wl.bases, wl.off, wl.off2)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    out = st.fetch()
    yield ganon_amd, wl, flt, st, out
    st.destroy()
    flt.free()


def test_config3_structure_and_planted_pairs(flat128g):
    hip, wl, flt, st, (nh, status, mo, m) = flat128g
    assert flt.info()["device_bytes"] == ROWS * 4096
    tm = st.timings()
    assert int(nh.sum(dtype=np.uint64)) == tm["n_hashes"] and tm["algo_bytes"] == tm["n_hashes"] * 4 * 4096
    assert tm["fetched_bytes"] <= tm["algo_bytes"]
    _properties(wl, nh, status, mo, m, 0)


def test_config3_idempotent_and_order_independent(flat128g):
    hip, wl, flt, st, (nh, status, mo, m) = flat128g
    ck = bw.checksum_matches(m)
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh2, status2, mo2, m2 = st.fetch()
    assert np.array_equal(nh, nh2) and np.array_equal(mo, mo2) and bw.checksum_matches(m2) == ck
    # a shuffled sub-batch gives the same per-pair answers (reads are independent, GanonClassify.cpp:676-831)
    rng = np.random.default_rng(3)
    pick = rng.choice(wl.n_reads, size=min(200_000, wl.n_reads), replace=False)
    L = wl.read_len
    m1 = wl.bases[: wl.n_reads * L].reshape(wl.n_reads, L)[pick]
    m2b = wl.bases[wl.n_reads * L:].reshape(wl.n_reads, L)[pick]
    sub = np.concatenate([m1.reshape(-1), m2b.reshape(-1)])
    o1 = np.arange(len(pick) + 1, dtype=np.uint64) * np.uint64(L)
    st2 = hip.HipStream(flt, len(pick), sub.size)
    st2.submit(sub, o1, o1 + np.uint64(len(pick) * L), wl.k, wl.w, wl.rel_cutoff)
    nh3, _, mo3, m3 = st2.fetch()
    assert np.array_equal(nh3, nh[pick])
    assert np.array_equal(np.diff(mo3.astype(np.int64)), np.diff(mo.astype(np.int64))[pick])
    for x in rng.integers(0, len(pick), size=2000).tolist():
        a = m3[int(mo3[x]):int(mo3[x + 1])]
        b = m[int(mo[pick[x]]):int(mo[pick[x] + 1])]
        assert np.array_equal(a["target"], b["target"]) and np.array_equal(a["count"], b["count"])
    st2.destroy()


def test_config3_early_exit_changes_nothing(flat128g, monkeypatch):
    hip, wl, flt, st, (nh, status, mo, m) = flat128g
    gu.SW.on("early_exit")
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh2, status2, mo2, m2 = st.fetch()
    tm = st.timings()
    assert tm["fetched_bytes"] == tm["algo_bytes"]
    assert np.array_equal(mo, mo2) and np.array_equal(m, m2)


def test_config3_sample_against_oracle(flat128g):
    hip, wl, flt, st, (nh, status, mo, m) = flat128g
    _oracle_sample(wl, flt, nh, mo, m, 0)


# ----------------------------------------------------------------------------------------------- configs[4]
def test_config4_column_slice_through_partition_and_rccl(flat128g):
    """Slice 5 of 8 of the 1 TiB filter.  Reuses the pairs of the fixture above; the first filter is released first
    (two 128 GiB matrices do not fit next to the batch buffers)."""
    import torch
    import torch.distributed as dist
    from ganon_amd import partition as gp
    hip, wl0, flt0, st0, _ = flat128g
    st0.destroy()
    flt0.free()
    W = BINS // 64
    wl = bw.make_device_flat_workload("slice1t", BINS, ROWS, 4, 1, paired=True, seed=4321, word_lo=SLICE * W, row_words_total=SLICES * W)
    # same pairs as configs[3] (generated once), the slice's own bits
    for a in ("bases", "off", "off2", "planted_genome", "n_reads"):
        setattr(wl, a, getattr(wl0, a))
    flt, _ = bw.device_filter(hip, wl)
    # the slice holds exactly the bits the unsliced filter has at those columns (position-keyed fill): spot rows
    rows = np.array([0, 1, ROWS // 3, ROWS - 1], dtype=np.uint64)
    twin = hip.fill_random_words(wl.seed, rows, W, 1, SLICE * W, SLICES * W)
    got = flt.download_row_list(rows, W)
    diff = got ^ twin
    assert (got & twin == twin).all() and np.count_nonzero(diff) < 64  # only emplaced genome bits may differ
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        target_lo = SLICE * BINS
        sl = gp.Slice(0, SLICE * W, (SLICE + 1) * W, BINS, np.arange(BINS, dtype=np.uint32),
                      np.arange(BINS, dtype=np.uint32) + np.uint32(target_lo))
        part = gp.PartitionedIbf(sl, 0, 1, gp.HipLocalFilter(flt, 0), comm_device="cuda")
        lo, hi, nh, status, mine = part.classify(wl.bases, wl.off, wl.off2, wl.k, wl.w, wl.rel_cutoff)
        assert (lo, hi) == (0, wl.n_reads)
        mo = np.searchsorted(mine["read"], np.arange(wl.n_reads + 1)).astype(np.uint64)
        _properties(wl, nh, status, mo, mine, target_lo)
        _oracle_sample(wl, flt, nh, mo, mine, target_lo, seed=78)
        # idempotent; and the host-staged exchange delivers the same records as the device-resident one
        ck = bw.checksum_matches(mine)
        part.local.device_records = lambda: None
        _, _, _, _, mine2 = part.classify(wl.bases, wl.off, wl.off2, wl.k, wl.w, wl.rel_cutoff)
        assert bw.checksum_matches(mine2) == ck and np.array_equal(mine2, mine)
        part.local.close()
    finally:
        dist.destroy_process_group()
        flt.free()


# ------------------------------------------------------------------- configs[3]'s filter as two bin-range parts
def test_config3_filter_as_two_64g_parts_gathered_equals_the_whole(flat128g, monkeypatch):
    """The 128 GiB filter of configs[3] cut at bin 16 384 into two 64 GiB column parts -- what ganon-classify places on two
    devices when the filter exceeds one (host/backend_hip.cpp) -- each classifying all 12.5 M pairs; the parts' matches,
    copied device to device and concatenated per pair by gn_gather, must be the whole filter's matches record for record."""
    import copy
    hip, wl0, flt0, st0, (nh, status, mo, m) = flat128g
    st0.destroy()
    flt0.free()                              # (idempotent: the slice test above may have released them already)
    W = BINS // 64
    parts, streams = [], []
    for g in range(2):
        wl = copy.copy(wl0)
        wl.bins = BINS // 2
        wl.word_lo, wl.row_words_total = g * (W // 2), W
        mine = (wl0.genome_bins // (BINS // 2)) == g
        wl.genomes = wl0.genomes[mine]
        wl.genome_bins = (wl0.genome_bins[mine] - g * (BINS // 2)).astype(np.uint32)
        flt, n_planted = bw.device_filter(hip, wl)
        assert flt.info()["device_bytes"] == ROWS * 2048 and n_planted > 0
        parts.append(flt)
    for flt in parts:
        st = hip.HipStream(flt, PAIRS, wl0.bases.size, PAIRS)
        st.upload(wl0.bases, wl0.off, wl0.off2)
        st.classify(wl0.k, wl0.w, wl0.rel_cutoff)
        streams.append(st)
    gu.SW.on("gather_copy")   # one GPU: take the device-to-device copy path all the same
    g = hip.HipGather(0, [None, np.arange(BINS // 2, dtype=np.uint32) + np.uint32(BINS // 2)])
    g.run(streams)
    mo2, m2 = g.fetch()
    assert g.peer_bytes() == 2 * (PAIRS + 1) * 8 + 12 * len(m)
    assert np.array_equal(mo2, mo) and np.array_equal(m2, m)
    nh2, status2 = streams[1].fetch_read_info()
    assert np.array_equal(nh2, nh) and np.array_equal(status2, status)
    g.destroy()
    for st in streams:
        st.destroy()
    for flt in parts:
        flt.free()
