"""GPU parity tests proper: every result comes through the C ABI (libganon_hip.so) and is compared
bit-exactly with the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import ganon_fixtures as gf
import gpu_util as gu
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import ganon_amd
    ganon_amd.load_library()
    assert ganon_amd.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return ganon_amd


def _classify(hip, flt, seqs1, seqs2, k, w, rel_cutoff):
    bases, off1, off2 = gu.pack_reads(seqs1, seqs2)
    st = hip.HipStream(flt, max(len(seqs1), 1), max(bases.size, 1))
    st.submit(bases, off1, off2, k, w, rel_cutoff)
    nh, status, mo, m = st.fetch()
    return st, nh, status, mo, m


def _tiny_filter(hip):
    ibf = gf.random_ibf(64, 257, 3, 0.3, 1)
    return hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs), ibf


# --------------------------------------------------------------------------------------------- minimiser
@pytest.mark.parametrize("k,w", [(19, 31), (4, 4), (10, 12), (32, 32), (1, 5), (21, 40), (4, 6), (31, 35), (8, 200)])
def test_minimiser_parity(hip, k, w):
    rng = np.random.default_rng(1000 * k + w)
    seqs = []
    for length in [0, 1, w - 1, w, w + 1, 64, 63 + w, 64 + w, 65 + w, 150, 151, 300, 1000, 5000]:
        if length < 0:
            continue
        seqs.append(gu.random_seq(rng, length))
    # low complexity / ties / IUPAC / lower case
    seqs += [b"A" * 150, b"ACACACACAC" * 20, b"AAAAACCCCC" * 15, gu.random_seq(rng, 200, b"AC"),
             gu.random_seq(rng, 200, b"ACGTNRYKMSWBDHVacgtnu"), gu.random_seq(rng, 333, b"AT")]
    seqs += [gu.random_seq(rng, int(rng.integers(0, 400))) for _ in range(300)]
    flt, _ = _tiny_filter(hip)
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, 0.5)
    ho, hs = st.fetch_hashes()
    exp = gu.oracle_hashes(seqs, None, k, w)
    for i, (es, eh) in enumerate(exp):
        assert status[i] == es, (i, len(seqs[i]))
        got = hs[int(ho[i]):int(ho[i + 1])]
        assert nh[i] == len(eh), (i, len(seqs[i]), nh[i], len(eh))
        assert np.array_equal(got, eh), (i, len(seqs[i]))


def test_minimiser_paired_parity(hip):
    k, w = 19, 31
    rng = np.random.default_rng(7)
    s1 = [gu.random_seq(rng, int(rng.integers(0, 300))) for _ in range(200)]
    s2 = [gu.random_seq(rng, int(rng.integers(0, 300))) for _ in range(200)]
    s1[0], s2[0] = b"ACGT" * 5, gu.random_seq(rng, 150)      # mate 1 shorter than w -> skipped regardless of mate 2
    s1[1], s2[1] = gu.random_seq(rng, 150), b"ACGT" * 5      # short mate 2 contributes nothing
    flt, _ = _tiny_filter(hip)
    st, nh, status, mo, m = _classify(hip, flt, s1, s2, k, w, 0.5)
    ho, hs = st.fetch_hashes()
    for i, (es, eh) in enumerate(gu.oracle_hashes(s1, s2, k, w)):
        assert status[i] == es
        assert np.array_equal(hs[int(ho[i]):int(ho[i + 1])], eh), i


def test_read_too_big(hip):
    # > 65535 minimisers -> GN_READ_BIG (GanonClassify.cpp:674,706); k == w emits every k-mer
    k = w = 8
    rng = np.random.default_rng(3)
    seqs = [gu.random_seq(rng, 65535 + 8), gu.random_seq(rng, 65535 + 7), gu.random_seq(rng, 100)]
    flt, _ = _tiny_filter(hip)
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, 1.0)
    assert list(status) == [hip.READ_BIG, hip.READ_OK, hip.READ_OK]
    assert list(nh) == [65536, 65535, 93]
    assert mo[1] - mo[0] == 0


def test_empty_batch(hip):
    flt, _ = _tiny_filter(hip)
    st, nh, status, mo, m = _classify(hip, flt, [], None, 19, 31, 0.2)
    assert len(nh) == 0 and len(m) == 0 and list(mo) == [0]


# --------------------------------------------------------------------------------------------- IBF counts
SHAPES = [  # bins, rows, h
    (1, 97, 1), (3, 211, 2), (64, 1000, 3), (100, 1531, 4), (128, 900, 5), (130, 777, 3), (500, 2048, 4),
    (4096, 4099, 4), (4100, 1200, 2), (8192, 1024, 3), (8256, 700, 4), (20000, 300, 2), (32768, 257, 4),
    (65536, 130, 3),
    (40000, 197, 3), (51264, 131, 2), (47168, 167, 4),    # rows of an odd number of words, 10 / 13 / 12 column slices: the 1024-thread shapes
]


@pytest.mark.parametrize("bins,rows,h", SHAPES)
def test_ibf_dense_counts_parity(hip, bins, rows, h):
    k, w = 19, 31
    rng = np.random.default_rng(bins * 31 + h)
    ibf = gf.random_ibf(bins, rows, h, 0.4, seed=bins + h)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    seqs = [gu.random_seq(rng, int(L)) for L in [31, 40, 150, 150, 150, 151, 299, 1000, 20, 4000]]
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, 0.3)
    ho, hs = st.fetch_hashes()
    dense = st.dense_counts(0, len(seqs), bins)
    b2t = np.arange(bins, dtype=np.uint32)
    for i in range(len(seqs)):
        hh = hs[int(ho[i]):int(ho[i + 1])]
        exp_m, exp_counts = gu.oracle_matches(ibf, b2t, bins, hh, 0.3) if status[i] == 0 else ([], np.zeros(bins, np.uint16))
        assert np.array_equal(dense[i], exp_counts), (i, np.nonzero(dense[i] != exp_counts)[0][:10])
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
        assert all(int(x["read"]) == i for x in m[int(mo[i]):int(mo[i + 1])])
        assert got == exp_m, (i, got[:5], exp_m[:5])


@pytest.mark.parametrize("rel_cutoff", [0.0, 0.2, 0.45, 0.75, 1.0])
def test_split_bins_and_cutoffs(hip, rel_cutoff):
    # targets own several (also non-contiguous) technical bins, some bins unassigned (GanonClassify.cpp:516-527)
    k, w = 19, 31
    bins, rows, h = 1000, 3001, 3
    rng = np.random.default_rng(11)
    ibf = gf.random_ibf(bins, rows, h, 0.35, seed=5)
    n_targets = 300
    b2t = rng.integers(0, n_targets, size=bins).astype(np.uint32)
    b2t[rng.integers(0, bins, size=50)] = 0xFFFFFFFF
    genomes = [gu.random_seq(rng, 2000) for _ in range(20)]
    for gi, g in enumerate(genomes):  # plant genomes so that real matches exist
        for hv in np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)):
            ibf.emplace(int(hv), gi * 7 % bins)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    seqs = []
    for i in range(200):
        if i % 2:
            g = genomes[i % 20]
            p = int(rng.integers(0, 1800))
            seqs.append(g[p:p + 150])
        else:
            seqs.append(gu.random_seq(rng, 150))
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, rel_cutoff)
    ho, hs = st.fetch_hashes()
    total = 0
    for i in range(len(seqs)):
        exp_m, _ = gu.oracle_matches(ibf, b2t, n_targets, hs[int(ho[i]):int(ho[i + 1])], rel_cutoff)
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
        assert got == exp_m, (i, got[:5], exp_m[:5])
        total += len(exp_m)
    assert total == len(m) and total > 0


def test_long_reads_many_flushes(hip):
    # thousands of minimisers per read: exercises the 15-iteration nibble flush many times
    k, w = 19, 23
    bins, rows, h = 4096, 2048, 4
    rng = np.random.default_rng(5)
    ibf = gf.random_ibf(bins, rows, h, 0.5, seed=9)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    seqs = [gu.random_seq(rng, 30000), gu.random_seq(rng, 12345), gu.random_seq(rng, 150)]
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, 0.1)
    ho, hs = st.fetch_hashes()
    dense = st.dense_counts(0, 3, bins)
    for i in range(3):
        assert nh[i] > (1000 if i < 2 else 10)
        assert np.array_equal(dense[i], ibf.bulk_count(hs[int(ho[i]):int(ho[i + 1])]))


def test_emplace_and_download(hip):
    bins, rows, h = 200, 1000, 3
    ibf = oracle.Ibf(bins, rows, h)
    flt = hip.HipFilter.ibf(None, bins, rows, h)
    rng = np.random.default_rng(2)
    hashes = rng.integers(0, 1 << 38, size=5000, dtype=np.uint64)
    bb = rng.integers(0, bins, size=5000).astype(np.uint32)
    flt.emplace(hashes, bb)
    for v, b in zip(hashes.tolist(), bb.tolist()):
        ibf.emplace(v, b)
    assert np.array_equal(flt.download_rows(0, rows, ibf.bin_words), ibf.data)


def test_kat_per_filter_matches(hip, kat):
    # reference KAT filters through the C ABI: GPU select_matches == oracle select_matches for every read
    for bname, b in kat["builds"].items():
        built = gf.build_ibf(b["targets"], b["k"], b["w"], max_fp=b["max_fp"])
        names = list(dict.fromkeys(t for _, t in built.bin_map))
        b2t = np.full(built.ibf.bins, 0xFFFFFFFF, dtype=np.uint32)
        for binno, t in built.bin_map:
            b2t[binno] = names.index(t)
        flt = hip.HipFilter.ibf(built.ibf.data, built.ibf.bins, built.ibf.bin_size, built.ibf.hash_funs, b2t, len(names))
        rnames = list(kat["reads"])
        seqs = [kat["reads"][r].replace("-", "A").encode() for r in rnames]
        for rc in (0.0, 0.2, 0.45, 0.6, 0.7, 1.0):
            st, nh, status, mo, m = _classify(hip, flt, seqs, None, b["k"], b["w"], rc)
            for i, s in enumerate(seqs):
                hh = oracle.minimiser_hash(oracle.to_ranks(s), b["k"], b["w"]) if len(s) >= b["w"] else np.zeros(0, np.uint64)
                exp_m, _ = gu.oracle_matches(built.ibf, b2t, len(names), hh, rc) if len(hh) else ([], None)
                got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
                assert got == exp_m, (bname, rnames[i], rc, got, exp_m)


# --------------------------------------------------------------------------------------------- HIBF
@pytest.mark.parametrize("n_ub,tmax,depth,rel_cutoff", [(40, 64, 1, 0.2), (300, 64, 2, 0.3), (1000, 64, 3, 0.1),
                                                         (500, 192, 2, 0.0), (200, 64, 2, 0.75), (2000, 128, 3, 0.5)])
def test_hibf_parity(hip, n_ub, tmax, depth, rel_cutoff):
    # counting_agent_type::bulk_count(values, T) (hibf.hpp:432-460,506-523) + select_matches (GanonClassify.cpp:543-577)
    k, w = 19, 31
    rng = np.random.default_rng(n_ub + tmax)
    genomes = {ub: gu.random_seq(rng, 1500) for ub in range(0, n_ub, 3)}
    uh = {ub: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in genomes.items()}
    hb = gf.random_hibf(n_ub, tmax, depth, seed=n_ub, density=0.3, hash_funs=3, user_hashes=uh)
    flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    assert flt.info()["is_hibf"] and flt.info()["n_targets"] == n_ub
    seqs = []
    keys = sorted(genomes)
    for i in range(300):
        if i % 3 == 0:
            seqs.append(gu.random_seq(rng, 150))
        elif i % 3 == 1:
            g = genomes[keys[i % len(keys)]]
            p = int(rng.integers(0, 1300))
            seqs.append(g[p:p + 150])
        else:
            seqs.append(gu.random_seq(rng, int(rng.integers(0, 60))))
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, rel_cutoff)
    ho, hs = st.fetch_hashes()
    dense = st.dense_counts(0, len(seqs), n_ub)
    n_true = 0
    algo = 0
    for i in range(len(seqs)):
        hh = hs[int(ho[i]):int(ho[i + 1])]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
        if status[i] != 0:
            assert got == [] and not dense[i].any()
            continue
        thr = oracle.threshold_cutoff(len(hh), rel_cutoff)
        exp_counts = hb.bulk_count(hh, thr)
        assert np.array_equal(dense[i], exp_counts), (i, np.nonzero(dense[i] != exp_counts)[0][:10])
        exp = [(int(u), int(min(c, len(hh)))) for u, c in enumerate(exp_counts) if c > 0]
        assert got == exp, (i, got[:5], exp[:5])
        n_true += len(exp)
        algo += hb.visited_bytes(hh, thr)
    assert n_true > 0
    assert st.timings()["algo_bytes"] == algo


@pytest.mark.parametrize("n_ub,tmax,depth,h,rel_cutoff", [(300, 64, 2, 2, 0.3), (700, 128, 3, 4, 0.6), (400, 320, 2, 5, 0.2),
                                                           (6000, 4480, 2, 3, 0.5), (150, 64, 2, 1, 0.9)])
def test_hibf_register_kernel_vs_lds_kernel_vs_oracle(hip, monkeypatch, n_ub, tmax, depth, h, rel_cutoff):
    # gn_hibf_pack_kernel (64/Gp items per wave; IBFs of the level's common lane width, single-bin runs) -> gn_hibf_reg_kernel
    # (one item per wave: other widths, split user bins) -> gn_hibf_level_kernel (everything; forced with the switches) and
    # the oracle: short, medium (n ~ 70) and long reads (n > 127: deferred to the LDS kernel inside a level), 1..5 hash
    # functions, an IBF wider than 64 words (tmax 4480 -> W = 70: deferred), split user bins, holes in the queues
    k, w = 19, 31
    rng = np.random.default_rng(n_ub * 7 + h)
    genomes = {ub: gu.random_seq(rng, 3000) for ub in range(0, n_ub, 5)}
    uh = {ub: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in genomes.items()}
    hb = gf.random_hibf(n_ub, tmax, depth, seed=n_ub + h, density=0.3, hash_funs=h, user_hashes=uh)
    flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    keys = sorted(genomes)
    seqs = []
    for i in range(400):
        L = (150, 150, 600, 2200)[i % 4] if i % 16 else 90
        if i % 2:
            g = genomes[keys[i % len(keys)]]
            p = int(rng.integers(0, 3000 - L))
            seqs.append(g[p:p + L])
        else:
            seqs.append(gu.random_seq(rng, L))
    st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, rel_cutoff)
    assert nh.max() > 127 and (nh[nh > 0] <= 127).any()
    tm = st.timings()
    for switch in ("hibf_pack", "hibf_reg", "hibf_one_pack"):  # per-item register kernel first / LDS kernel only / no sorting of a level by IBF width
        gu.SW.on(switch)
        st2, nh2, status2, mo2, m2 = _classify(hip, flt, seqs, None, k, w, rel_cutoff)
        gu.SW.off(switch)
        assert np.array_equal(nh, nh2) and np.array_equal(mo, mo2) and np.array_equal(m, m2), switch
        assert st2.timings()["algo_bytes"] == tm["algo_bytes"], switch
    ho, hs = st.fetch_hashes()
    n_true, algo = 0, 0
    for i in range(len(seqs)):
        hh = hs[int(ho[i]):int(ho[i + 1])]
        thr = oracle.threshold_cutoff(len(hh), rel_cutoff)
        exp_counts = hb.bulk_count(hh, thr)
        exp = [(int(u), int(min(c, len(hh)))) for u, c in enumerate(exp_counts) if c > 0]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
        assert got == exp, (i, len(hh), got[:5], exp[:5])
        n_true += len(exp)
        algo += hb.visited_bytes(hh, thr)
    assert n_true > 50 and tm["algo_bytes"] == algo


# --------------------------------------------------------------------------------------------- bin-range partition
def test_partition_slices_hip(hip):
    # config-5 style column slices on ONE GPU: the union of the per-slice results == the unpartitioned filter
    import dist_worker as dw
    from ganon_amd import partition as gp
    ibf, b2t, n_targets, seqs = dw.make_case(seed=9)
    bases, off1, _ = gu.pack_reads(seqs, None)
    full = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, n_targets)
    st, nh, status, mo, m_full = _classify(hip, full, seqs, None, dw.K, dw.W, 0.25)
    for world in (2, 4):
        parts = []
        for sl in gp.plan_partition(b2t, ibf.bins, world):
            rows = gp.slice_rows(ibf.data, ibf.bin_words, sl)
            loc = gp.HipLocalFilter.from_rows(rows, sl.bins_local, ibf.bin_size, ibf.hash_funs, sl.bin2target_local,
                                              max(1, len(sl.targets_global)))
            nh2, st2, mo2, m = loc.classify(bases, off1, None, dw.K, dw.W, 0.25).fetch()
            # the device-resident view of the same matches (what the RCCL exchange sends)
            rec = loc.device_records().cpu().numpy()
            assert np.array_equal(rec.view(np.uint32).reshape(-1, 3).view(hip.MATCH_DTYPE).reshape(-1), m)
            loc.close()
            assert np.array_equal(nh2, nh)
            g = np.zeros(len(m), dtype=hip.MATCH_DTYPE)
            g["read"], g["count"] = m["read"], m["count"]
            g["target"] = sl.targets_global[m["target"]] if len(m) else 0
            parts.append(g)
        got = np.concatenate(parts)
        got = got[np.lexsort((got["target"], got["read"]))]
        assert len(m_full) > 50 and np.array_equal(got, m_full), world


def test_partition_exchange_over_rccl_world1(hip):
    # The config-5 exchange step with device tensors over the nccl (= RCCL) backend.  One GPU means world size 1, so
    # the all-to-all is degenerate, but it goes through the same RCCL calls, tensor placement and split bookkeeping
    # that the N-GPU job uses (the N > 1 logic is covered by the gloo tests).
    import socket
    import torch
    import torch.distributed as dist
    import dist_worker as dw
    from ganon_amd import partition as gp
    ibf, b2t, n_targets, seqs = dw.make_case(seed=11)
    bases, off1, _ = gu.pack_reads(seqs, None)
    full = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, n_targets)
    st, nh, status, mo, m_full = _classify(hip, full, seqs, None, dw.K, dw.W, 0.25)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        part = gp.PartitionedIbf.from_host_rows(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, 0, 1,
                                                gp.HipLocalFilter.from_rows, comm_device="cuda")
        lo, hi, nh2, status2, mine = part.classify(bases, off1, None, dw.K, dw.W, 0.25)  # device-resident exchange
        part.local.device_records = lambda: None  # same collective, records staged through host memory
        lo3, hi3, nh3, status3, mine3 = part.classify(bases, off1, None, dw.K, dw.W, 0.25)
        assert np.array_equal(mine3, mine)
        part.local.close()
    finally:
        dist.destroy_process_group()
    assert (lo, hi) == (0, len(seqs)) and np.array_equal(nh2, nh)
    assert len(m_full) > 50 and np.array_equal(mine, m_full)


@pytest.mark.parametrize("bins,rows,h,paired", [(4096, 5003, 4, False), (4096, 5003, 4, True), (32768, 1201, 4, False),
                                                  (32768, 1201, 3, True), (8192, 2003, 5, False), (640, 3001, 2, True),
                                                  (20480, 1201, 4, False), (36864, 701, 3, True)])
def test_planted_matches_many_reads(hip, bins, rows, h, paired):
    # thousands of reads per launch (persistent waves loop over many reads), true matches with counts up to n
    # (> 15: the 4-bit first-level counters must spill correctly), single and paired (n ~ 35) reads
    k, w = 19, 31
    rng = np.random.default_rng(bins + h + paired)
    ibf = gf.random_ibf(bins, rows, h, 0.3, seed=bins + 7 * h)
    genomes = [gu.random_seq(rng, 1200) for _ in range(64)]
    gb = rng.integers(0, bins, size=64)
    for g, b in zip(genomes, gb):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), int(b))
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    n = 3000
    s1, s2 = [], []
    for i in range(n):
        if i % 3 == 2:
            s1.append(gu.random_seq(rng, 150))
            s2.append(gu.random_seq(rng, 150))
        else:
            g = genomes[i % 64]
            p = int(rng.integers(0, 900))
            s1.append(g[p:p + 150])
            s2.append(g[p + 100:p + 250])
    st, nh, status, mo, m = _classify(hip, flt, s1, s2 if paired else None, k, w, 0.75)
    ho, hs = st.fetch_hashes()
    b2t = np.arange(bins, dtype=np.uint32)
    tot = 0
    for i in range(n):
        exp_m, _ = gu.oracle_matches(ibf, b2t, bins, hs[int(ho[i]):int(ho[i + 1])], 0.75)
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
        assert got == exp_m, (i, nh[i], got[:4], exp_m[:4])
        tot += len(exp_m)
    assert tot >= n // 2 and tot == len(m)
    assert nh.max() > (30 if paired else 15)


def test_chunked_pipeline_parity(hip, monkeypatch):
    # switch chunk=N cuts the batch into chunks pipelined over two HIP streams (minimiser || count); results must
    # not depend on the chunking
    k, w = 19, 31
    bins, rows, h = 4096, 4099, 4
    rng = np.random.default_rng(21)
    ibf = gf.random_ibf(bins, rows, h, 0.3, seed=3)
    genomes = [gu.random_seq(rng, 1000) for _ in range(32)]
    for gi, g in enumerate(genomes):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), gi * 97 % bins)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    seqs = [genomes[i % 32][(i * 7) % 800:(i * 7) % 800 + 150] if i % 2 else gu.random_seq(rng, int(rng.integers(10, 900)))
            for i in range(5000)]
    outs = []
    for chunk in ("0", "1", "333", "4096"):
        gu.SW.on(f"chunk={chunk}")
        st, nh, status, mo, m = _classify(hip, flt, seqs, None, k, w, 0.6)
        outs.append((nh.copy(), status.copy(), mo.copy(), m.copy(), st.timings()["n_count_launches"]))
        st.destroy()
    assert outs[0][4] == 1 and outs[1][4] == 64 and outs[2][4] == 16 and outs[3][4] == 2  # capped at 64 launches
    for o in outs[1:]:
        for a, b in zip(outs[0][:4], o[:4]):
            assert np.array_equal(a, b)
    assert len(outs[0][3]) > 2000


@pytest.mark.gpu
@pytest.mark.parametrize("bins,rows,h", [(4096, 5003, 4), (4096, 3001, 2), (8192, 2003, 3), (32768, 1201, 4), (4032, 2003, 5),
                                         (4096, 3001, 1), (16384, 1501, 5), (20480, 1201, 4), (36864, 701, 4)])
def test_early_exit_is_exact(hip, monkeypatch, bins, rows, h):
    # Reads whose best count lands just below / at / above the cutoff (mutated copies of planted genomes), several
    # cutoffs: the fast kernel's early exit (stop fetching rows once no bin can still reach the cutoff) must not
    # change a single match -- against the oracle and against the same kernel with the exit disabled -- and must
    # actually skip rows when the cutoff leaves room for it.
    k, w = 19, 31
    rng = np.random.default_rng(bins * 3 + h)
    ibf = gf.random_ibf(bins, rows, h, 0.5, seed=bins + h)
    genomes = [gu.random_seq(rng, 900) for _ in range(48)]
    for gi, g in enumerate(genomes):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), (gi * 611 + 5) % bins)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    reads = []
    for i in range(2400):
        g = genomes[i % 48]
        p = int(rng.integers(0, 700))
        s = bytearray(g[p:p + 150])
        for _ in range(i % 8):  # 0..7 substitutions: minimiser hits range from all to about a third
            q = int(rng.integers(0, 150))
            s[q] = b"ACGT"[(b"ACGT".index(s[q]) + 1 + int(rng.integers(0, 3))) % 4]
        reads.append(bytes(s) if i % 5 else gu.random_seq(rng, 150))
    b2t = np.arange(bins, dtype=np.uint32)
    for cutoff in (0.0, 0.3, 0.55, 0.75, 0.9, 1.0):
        gu.SW.off("early_exit")
        st, nh, status, mo, m = _classify(hip, flt, reads, None, k, w, cutoff)
        tm = st.timings()
        ho, hs = st.fetch_hashes()
        gu.SW.on("early_exit")
        st2, nh2, status2, mo2, m2 = _classify(hip, flt, reads, None, k, w, cutoff)
        tm2 = st2.timings()
        assert np.array_equal(mo, mo2) and np.array_equal(m, m2), cutoff
        assert tm2["fetched_bytes"] == tm2["algo_bytes"] == tm["algo_bytes"]
        assert tm["fetched_bytes"] <= tm["algo_bytes"]
        if cutoff == 0.0:
            assert tm["fetched_bytes"] == tm["algo_bytes"]     # T = 1: nothing can be ruled out early
        if cutoff >= 0.75 and bins % 4096 == 0 and h >= 3:  # (h = 2 at 50 % fill: random bins stay in the race)
            assert tm["fetched_bytes"] < tm["algo_bytes"]      # rows were skipped
        near = 0
        for i in range(0, len(reads), 3):
            hh = hs[int(ho[i]):int(ho[i + 1])]
            exp_m, _ = gu.oracle_matches(ibf, b2t, bins, hh, cutoff)
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp_m, (cutoff, i, got[:3], exp_m[:3])
            thr = oracle.threshold_cutoff(len(hh), cutoff)
            near += any(abs(c - thr) <= 1 for _, c in exp_m)
        if 0.3 <= cutoff <= 0.9:
            assert near > 10   # the boundary is exercised
        st.destroy()
        st2.destroy()
    flt.free()


@pytest.mark.gpu
@pytest.mark.parametrize("bins,rows,h,contiguous", [(4096, 4001, 4, True), (4096, 4001, 3, False), (1024, 9001, 4, False),
                                                    (16384, 1501, 4, True), (16384, 1501, 2, False), (704, 9001, 5, False),
                                                    (4096, 2001, 2, False), (36864, 701, 4, True), (36864, 701, 3, False),
                                                    (20480, 1201, 4, False), (51264, 401, 3, True), (40000, 401, 4, False)])
def test_candidate_select_matches_target_scan(hip, monkeypatch, bins, rows, h, contiguous):
    # Split-bin maps of every kind (targets of 1..300 bins, contiguous runs or scattered bins, bins of no target, rows
    # of one to several column slices and rows narrower than a wave): the generic kernel's candidate-driven select
    # (count*nb >= T prefilter, lowest-candidate-bin rule, staged and direct output) must give what the plain scan
    # over every target gives, and what the oracle gives -- including reads with more hits than the staging list.
    k, w = 19, 31
    rng = np.random.default_rng(bins + 13 * h + contiguous)
    sizes = []
    left = bins - bins // 16          # a sixteenth of the bins belongs to no target
    while left > 0:
        s = int(rng.choice([1, 1, 2] if rows == 2001 else [1, 1, 1, 1, 2, 2, 2, 3, 4, 5, 9, 40, 300]))
        s = min(s, left)
        sizes.append(s)
        left -= s
    n_targets = len(sizes)
    order = np.arange(bins) if contiguous else rng.permutation(bins)
    b2t = np.full(bins, 0xFFFFFFFF, dtype=np.uint32)
    pos = 0
    tbins = []
    for t, s in enumerate(sizes):
        b2t[order[pos:pos + s]] = t
        tbins.append(order[pos:pos + s])
        pos += s
    ibf = gf.random_ibf(bins, rows, h, 0.45, seed=bins + h)
    genomes = []
    for gi in range(60):
        t = int(rng.integers(0, n_targets))
        g = gu.random_seq(rng, 1500)
        hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w))
        parts = np.array_split(hs, len(tbins[t])) if len(tbins[t]) <= 8 else np.array_split(hs, 8)
        for pi, part in enumerate(parts):     # the genome's hashes are spread over the target's bins, like ganon-build
            if len(part):
                ibf.emplace_many(part, int(tbins[t][pi]))
        if gi % 5 == 0:                        # ... and some are in every bin of their target: one hash then hits several
            for b in tbins[t][:8]:             # bins of the target, the sum passes n and is capped (:525-526)
                ibf.emplace_many(hs, int(b))
        genomes.append(g)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    reads = []
    for i in range(1500):
        g = genomes[i % 60]
        p = int(rng.integers(0, 1300))
        s = bytearray(g[p:p + 150])
        for _ in range(i % 6):
            q = int(rng.integers(0, 150))
            s[q] = b"ACGT"[(b"ACGT".index(s[q]) + 1 + int(rng.integers(0, 3))) % 4]
        reads.append(bytes(s) if i % 4 else gu.random_seq(rng, 150))
    reads.append(g[:640])                      # a long read (n > 64: several row-table chunks)
    for cutoff in (0.05, 0.1, 0.3, 0.75, 1.0):
        gu.SW.off("cand_select")
        st, nh, status, mo, m = _classify(hip, flt, reads, None, k, w, cutoff)
        ho, hs = st.fetch_hashes()
        # default = split kernel (register counters, byte image) for reads of up to 127 minimisers; without it the
        # generic kernel's candidate select; without that the plain scan over every target
        gu.SW.on("split_kernel")
        st3, nh3, status3, mo3, m3 = _classify(hip, flt, reads, None, k, w, cutoff)
        assert np.array_equal(mo, mo3) and np.array_equal(m, m3), cutoff
        st3.destroy()
        gu.SW.on("cand_select")
        st2, nh2, status2, mo2, m2 = _classify(hip, flt, reads, None, k, w, cutoff)
        gu.SW.off("split_kernel")
        assert np.array_equal(mo, mo2) and np.array_equal(m, m2), cutoff
        per_read = np.diff(mo.astype(np.int64))
        assert (m["count"] <= nh[m["read"]]).all()
        if cutoff == 1.0:
            assert (m["count"] == nh[m["read"]]).all() and len(m) > 100   # capped sums are reported as n
        if cutoff == 0.1 and h == 2 and bins == 4096:
            assert per_read.max() > 128        # the direct (count-then-write) pass of the candidate select ran
        for i in range(0, len(reads), 4):
            exp_m, _ = gu.oracle_matches(ibf, b2t, n_targets, hs[int(ho[i]):int(ho[i + 1])], cutoff)
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp_m, (cutoff, i, got[:3], exp_m[:3])
        st.destroy()
        st2.destroy()
    flt.free()


# --------------------------------------------------------------------------------------------- device fill / streaming load
@pytest.mark.gpu
def test_fill_random_is_position_keyed_and_matches_numpy_twin(hip):
    # gn_filter_fill_random == ganon_amd.fill_random_words bit for bit; a column slice holds the unsliced filter's bits
    S, bins_total, h = 777, 1000, 3
    Wt = (bins_total + 63) >> 6
    full = hip.HipFilter.ibf(None, bins_total, S, h)
    full.fill_random(99, 1)
    got = full.download_rows(0, S, Wt)
    twin = hip.fill_random_words(99, np.arange(S), Wt, 1, 0, Wt, bins=bins_total)
    assert np.array_equal(got, twin)
    assert abs(np.unpackbits(got[:, :-1].view(np.uint8)).mean() - 0.5) < 0.01 and (got[:, -1] >> np.uint64(bins_total & 63) == 0).all()
    for word_lo, bins_local in ((0, 320), (5, 384), (10, 1000 - 640)):
        Wl = (bins_local + 63) >> 6
        sl = hip.HipFilter.ibf(None, bins_local, S, h)
        sl.fill_random(99, 1, word_lo, Wt)
        exp = got[:, word_lo:word_lo + Wl].copy()
        if bins_local & 63:
            exp[:, -1] &= np.uint64((1 << (bins_local & 63)) - 1)
        assert np.array_equal(sl.download_rows(0, S, Wl), exp)
        # row gather agrees with the range download
        pick = np.array([0, 5, 776, 5, 300], dtype=np.uint64)
        assert np.array_equal(sl.download_row_list(pick, Wl), exp[pick.astype(np.int64)])
        sl.free()
    quarter = hip.HipFilter.ibf(None, bins_total, S, h)
    quarter.fill_random(7, 2)
    q = quarter.download_rows(0, S, Wt)
    assert np.array_equal(q, hip.fill_random_words(7, np.arange(S), Wt, 2, 0, Wt, bins=bins_total))
    assert abs(np.unpackbits(q[:, :-1].view(np.uint8)).mean() - 0.25) < 0.01
    quarter.free()
    full.free()


@pytest.mark.gpu
def test_streaming_write_rows_equals_one_shot_upload(hip):
    # gn_filter_write_rows in chunks (whole rows, and one column slice out of wider source rows) == gn_filter_upload_ibf
    rng = np.random.default_rng(5)
    S, bins, h = 3001, 700, 3
    ibf = gf.random_ibf(bins, S, h, 0.3, seed=3)
    seqs = [gu.random_seq(rng, 150) for _ in range(300)]
    for gi in range(20):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(seqs[gi]), 19, 31)), gi * 31 % bins)
    ref = hip.HipFilter.ibf(ibf.data, bins, S, h)
    st1, nh, status, mo, m = _classify(hip, ref, seqs, None, 19, 31, 0.3)
    streamed = hip.HipFilter.ibf(None, bins, S, h)
    for r0 in range(0, S, 500):
        streamed.write_rows(r0, ibf.data[r0:r0 + 500])
    streamed.finalize()
    assert np.array_equal(streamed.download_rows(0, S, ibf.bin_words), ref.download_rows(0, S, ibf.bin_words))
    st2, nh2, status2, mo2, m2 = _classify(hip, streamed, seqs, None, 19, 31, 0.3)
    assert len(m) > 15 and np.array_equal(m, m2) and np.array_equal(nh, nh2)
    st1.destroy()
    st2.destroy()
    # column slice: words [3, 8) of every 11-word source row, local bins 5*64 - 20 (padding cleared by finalize)
    lo, Wl, bl = 3, 5, 300
    sl = hip.HipFilter.ibf(None, bl, S, h)
    for r0 in range(0, S, 777):
        sl.write_rows(r0, ibf.data[r0:r0 + 777], word_lo=lo)
    sl.finalize()
    exp = ibf.data[:, lo:lo + Wl].copy()
    exp[:, -1] &= np.uint64((1 << (bl & 63)) - 1)
    assert np.array_equal(sl.download_rows(0, S, Wl), exp)
    for f in (ref, streamed, sl):
        f.free()


def _exact_filter_matches(m_read, n_hashes, rel_filter, fpr_query, tfpr):
    """filter_matches (GanonClassify.cpp:579-613, threshold :755-761) on one read's (target, count) list through the oracle;
    -> (kept list, n dropped by rel_filter, n dropped by fpr_query, max_count)"""
    import ctypes as C
    import math
    if len(m_read) == 0:
        return [], 0, 0, 0
    counts = np.array([c for _, c in m_read], dtype=np.uint64)
    fprs = np.array([tfpr[t] for t, _ in m_read], dtype=np.float64)
    mx, mn = int(counts.max()), min(int(n_hashes), int(counts.min()))
    thr = mx - int(math.ceil(float(mx - mn) * rel_filter))
    keep = np.zeros(len(m_read), dtype=np.uint8)
    L = oracle.lib()
    L.gno_filter_matches.restype = C.c_size_t
    L.gno_filter_matches(counts.ctypes.data_as(C.c_void_p), fprs.ctypes.data_as(C.c_void_p), C.c_size_t(len(m_read)),
                         C.c_uint64(int(n_hashes)), C.c_uint64(thr), C.c_double(fpr_query), keep.ctypes.data_as(C.c_void_p))
    kept = [m_read[i] for i in range(len(m_read)) if keep[i] == 1]
    return kept, int((keep == 2).sum()), int((keep == 3).sum()), mx


@pytest.mark.parametrize("shape", ["identity4096", "split1000", "hibf"])
@pytest.mark.parametrize("rel_filter,fpr_query", [(0.1, 1e-5), (0.0, 1.0), (1.0, 0.01), (0.5, 1.0), (0.25, 1e-9)])
def test_device_postfilter_is_a_safe_prepass_of_filter_matches(hip, shape, rel_filter, fpr_query):
    # gn_stream_set_postfilter: the --rel-filter rule is applied exactly, the --fpr-query rule conservatively; applying the
    # exact rule to the survivors gives exactly what filter_matches gives on the unfiltered matches
    k, w = 19, 31
    rng = np.random.default_rng(77)
    genomes = [gu.random_seq(rng, 3000) for _ in range(24)]
    if shape == "hibf":
        uh = {ub * 9: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in enumerate(genomes)}
        hb = gf.random_hibf(600, 128, 2, seed=31, density=0.42, hash_funs=3, user_hashes=uh)
        flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
        n_targets = 600
    else:
        bins, rows, h = (4096, 1531, 4) if shape == "identity4096" else (1000, 3001, 3)
        ibf = gf.random_ibf(bins, rows, h, 0.42, seed=5)
        n_targets = bins if shape == "identity4096" else 300
        b2t = None if shape == "identity4096" else rng.integers(0, n_targets, size=bins).astype(np.uint32)
        for gi, g in enumerate(genomes):
            for hv in np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)):
                ibf.emplace(int(hv), gi * 41 % bins)
        flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    tfpr = rng.choice([1e-4, 0.003, 0.02, 0.05, 0.11, 0.3], size=n_targets)
    seqs = []
    for i in range(600):
        L = int(rng.choice([100, 150, 150, 250, 600]))
        if i % 3:
            g = genomes[i % 24]
            p = int(rng.integers(0, 3000 - L))
            seqs.append(g[p:p + L])
        else:
            seqs.append(gu.random_seq(rng, L))
    bases, off1, off2 = gu.pack_reads(seqs, None)
    st = hip.HipStream(flt, len(seqs), bases.size)
    st.submit(bases, off1, off2, k, w, 0.15)
    nh, status, mo, m = st.fetch()
    raw = [[(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]] for i in range(len(seqs))]
    assert max(len(r) for r in raw) > 40  # the cooperative path is exercised
    st.set_postfilter(rel_filter, fpr_query, tfpr)
    st.submit(bases, off1, off2, k, w, 0.15)
    nh2, status2, mo2, m2 = st.fetch()
    mx, d_fil, d_fpr = st.fetch_postfilter()
    assert np.array_equal(nh, nh2) and np.array_equal(status, status2)
    e_fil = e_fpr = on_device = n_marked = n_kept = 0
    for i in range(len(seqs)):
        kept, nf, nq, emx = _exact_filter_matches(raw[i], nh[i], rel_filter, fpr_query, tfpr)
        got = [(int(x["target"]), int(x["count"]) & 0x7FFFFFFF) for x in m2[int(mo2[i]):int(mo2[i + 1])]]
        marked = [(int(x["target"]), int(x["count"]) & 0x7FFFFFFF) for x in m2[int(mo2[i]):int(mo2[i + 1])] if int(x["count"]) >> 31]
        assert set(marked) <= set(kept)  # GN_MATCH_FPR_OK only on matches the exact rule keeps
        n_marked += len(marked)
        n_kept += len(kept)
        assert all(int(x["read"]) == i for x in m2[int(mo2[i]):int(mo2[i + 1])])
        assert int(mx[i]) == emx
        assert set(kept) <= set(got) <= set(raw[i]), i
        assert got == [x for x in raw[i] if x in set(got)]  # original order
        # the host applies the exact --fpr-query rule to the survivors: what it keeps is what filter_matches keeps
        exact_drop = set(raw[i]) - set(kept)
        assert [x for x in got if x not in exact_drop] == kept, i
        assert len(raw[i]) - len(got) >= nf
        e_fil += nf
        e_fpr += nq
        on_device += len(raw[i]) - len(got) - nf
    assert d_fil == e_fil
    assert d_fpr == on_device and d_fpr <= e_fpr
    if fpr_query < 1.0 and e_fpr > 50:
        assert d_fpr >= 0.9 * e_fpr  # the margin leaves only borderline cases to the host
    if fpr_query >= 1.0:
        assert n_marked == 0
    elif fpr_query > 1e-8 and n_kept > 50:
        assert n_marked >= 0.9 * n_kept
    st.set_postfilter(None)
    st.submit(bases, off1, off2, k, w, 0.15)
    _, _, mo3, m3 = st.fetch()
    assert np.array_equal(mo3, mo) and np.array_equal(m3, m)
    st.destroy()
    flt.free()


def test_postfilter_survives_match_buffer_regrow_and_empty_batches(hip):
    # a match buffer far too small for the unfiltered matches: the batch is re-run after the buffer grew, the pre-pass with it
    k, w = 19, 31
    rng = np.random.default_rng(123)
    ibf = gf.random_ibf(2048, 1201, 3, 0.45, seed=8)
    flt = hip.HipFilter.ibf(ibf.data, 2048, 1201, 3)
    seqs = [gu.random_seq(rng, 150) for _ in range(400)]
    bases, off1, off2 = gu.pack_reads(seqs, None)
    tfpr = np.full(2048, 0.09)
    ref = hip.HipStream(flt, len(seqs), bases.size)
    ref.set_postfilter(0.2, 1e-3, tfpr)
    ref.submit(bases, off1, off2, k, w, 0.1)
    nh, status, mo, m = ref.fetch()
    mx, a, b = ref.fetch_postfilter()
    assert a + b > 20000 and len(m) > 0
    small = hip.HipStream(flt, len(seqs), bases.size, max_matches=64)
    small.set_postfilter(0.2, 1e-3, tfpr)
    for _ in range(2):  # (the second batch finds the buffer already grown)
        small.submit(bases, off1, off2, k, w, 0.1)
        nh2, status2, mo2, m2 = small.fetch()
        mx2, a2, b2 = small.fetch_postfilter()
        assert np.array_equal(mo, mo2) and np.array_equal(m, m2) and np.array_equal(mx, mx2) and (a, b) == (a2, b2)
    # reads too short for a window / no reads at all
    e_bases, e_off, _ = gu.pack_reads([b"ACGT", b""], None)
    small.submit(e_bases, e_off, None, k, w, 0.1)
    nh3, status3, mo3, m3 = small.fetch()
    mx3, a3, b3 = small.fetch_postfilter()
    assert list(status3) == [1, 1] and len(m3) == 0 and list(mo3) == [0, 0, 0] and list(mx3) == [0, 0] and (a3, b3) == (0, 0)
    for s in (ref, small):
        s.destroy()
    flt.free()


@pytest.mark.parametrize("split_map", [False, True])
def test_long_reads_option_classifies_what_the_default_skips(hip, split_map):
    # reads with more than 65535 minimisers: GN_READ_BIG by default (TIntCount = uint16_t, GanonClassify.cpp:45-49,674);
    # with gn_stream_set_long_reads they are counted with 32-bit counters like the reference's -DLONGREADS build
    k, w = 19, 31
    rng = np.random.default_rng(31)
    bins, rows, h = 300, 40009, 3
    ibf = gf.random_ibf(bins, rows, h, 0.2, seed=4)
    genome = gu.random_seq(rng, 620_000)
    gh = np.unique(oracle.minimiser_hash(oracle.to_ranks(genome), k, w))
    ibf.emplace_many(gh[: len(gh) // 2], 17)
    ibf.emplace_many(gh[len(gh) // 2:], 18)
    n_targets = 120 if split_map else bins
    b2t = None
    if split_map:
        b2t = rng.integers(0, n_targets, size=bins).astype(np.uint32)
        b2t[17] = b2t[18] = 5  # the genome's two bins belong to one target
        b2t[rng.integers(0, bins, size=10)] = 0xFFFFFFFF
        b2t[17] = b2t[18] = 5
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    s1 = [genome[1000:601_000], gu.random_seq(rng, 560_000), genome[:300_000], gu.random_seq(rng, 150), genome[5000:5150]]
    s2 = [b"", b"", genome[300_000:600_000], b"", b""]
    for paired in (False, True):
        bases, off1, off2 = gu.pack_reads(s1, s2 if paired else None)
        st = hip.HipStream(flt, len(s1), bases.size)
        for cutoff in (0.0, 0.3, 0.9):
            st.set_long_reads(False)
            st.submit(bases, off1, off2, k, w, cutoff)
            nh0, status0, mo0, m0 = st.fetch()
            st.set_long_reads(True)
            st.submit(bases, off1, off2, k, w, cutoff)
            nh, status, mo, m = st.fetch()
            assert np.array_equal(nh, nh0)
            big = [i for i in range(len(s1)) if nh[i] > 65535]
            assert big == ([0, 1, 2] if paired else [0, 1])
            for i in range(len(s1)):
                got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
                if i not in big:  # untouched by the option
                    assert status[i] == status0[i]
                    assert got == [(int(x["target"]), int(x["count"])) for x in m0[int(mo0[i]):int(mo0[i + 1])]]
                    continue
                assert status0[i] == 2 and mo0[i] == mo0[i + 1] and status[i] == 0
                hh = oracle.minimiser_hash(oracle.to_ranks(s1[i]), k, w)
                if paired and len(s2[i]) >= w:
                    hh = np.concatenate([hh, oracle.minimiser_hash(oracle.to_ranks(s2[i]), k, w)])
                assert len(hh) == nh[i]
                counts = np.zeros(bins, dtype=np.uint64)
                for a in range(0, len(hh), 60000):  # uint16 counters cannot wrap within 60000 hashes
                    counts += ibf.bulk_count(hh[a:a + 60000]).astype(np.uint64)
                T = max(1, oracle.threshold_rel(len(hh), cutoff))
                exp = []
                for t in range(n_targets):
                    c = int(counts[t]) if b2t is None else int(counts[b2t == t].sum())
                    c = min(c, len(hh))
                    if c >= T:
                        exp.append((t, c))
                assert got == exp, (paired, cutoff, i, got[:4], exp[:4])
                if i != 1:
                    assert any(c > 65535 for _, c in got)  # the planted genome: a count no uint16 holds
        st.destroy()
    flt.free()


def _hibf_bulk_count_wide(hb, hashes, threshold):
    """the C oracle's agent with value_t = uint32_t (hierarchical_interleaved_bloom_filter.hpp:432-460 as the reference's
    -DLONGREADS build instantiates it): {user bin: sum} of the sums that reached the threshold"""
    counts = hb.bulk_count_longreads(hashes, threshold)
    return {int(u): int(counts[u]) for u in np.nonzero(counts)[0]}


def test_long_reads_option_for_an_hibf(hip):
    # an HIBF with gn_stream_set_long_reads: reads of more than 65535 minimisers are counted (through the LDS-counter level
    # kernel) instead of skipped, and no per-user-bin sum wraps at 2^16 -- for ANY read, like the reference's uint32 build
    k, w = 19, 31
    rng = np.random.default_rng(9)
    genome = gu.random_seq(rng, 640_000)
    other = gu.random_seq(rng, 40_000)
    uh = {7: np.unique(oracle.minimiser_hash(oracle.to_ranks(genome), k, w)), 33: np.unique(oracle.minimiser_hash(oracle.to_ranks(other), k, w))}
    hb = gf.random_hibf(90, 32, 2, seed=12, density=0.15, hash_funs=2, rows=(60000, 90000), user_hashes=uh)
    flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    seqs = [genome[1000:631_000], other[100:39_000], genome[200_000:200_150], gu.random_seq(rng, 600_000), other[5000:5250]]
    bases, off1, off2 = gu.pack_reads(seqs, None)
    st = hip.HipStream(flt, len(seqs), bases.size)
    for cutoff in (0.1, 0.6):
        st.set_long_reads(False)
        st.submit(bases, off1, off2, k, w, cutoff)
        nh0, status0, mo0, m0 = st.fetch()
        st.set_long_reads(True)
        st.submit(bases, off1, off2, k, w, cutoff)
        nh, status, mo, m = st.fetch()
        assert np.array_equal(nh, nh0) and [int(x) for x in status0] == [2, 0, 0, 2, 0] and (status == 0).all()
        for i, sq in enumerate(seqs):
            hh = oracle.minimiser_hash(oracle.to_ranks(sq), k, w)
            assert len(hh) == nh[i]
            T = max(1, oracle.threshold_rel(len(hh), cutoff))
            exp = sorted((u, min(c, len(hh))) for u, c in _hibf_bulk_count_wide(hb, hh, T).items())
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp, (cutoff, i, got[:4], exp[:4])
            if status0[i] == 2:
                assert mo0[i] == mo0[i + 1]
        got0 = [(int(x["target"]), int(x["count"])) for x in m[int(mo[0]):int(mo[1])]]
        assert any(u == 7 and c > 65535 for u, c in got0)  # the planted genome: a count no uint16 holds
    st.destroy()
    flt.free()


def test_hibf_sums_wrap_at_16_bits_in_the_default_build_only(hip):
    # a user bin split over several technical bins that ALL hold a read's minimisers: the per-user-bin sum passes 65535 although
    # the read has fewer minimisers than that.  Default build: value_t = uint16_t wraps (hibf.hpp:438,442) -- the oracle's C
    # restatement does the same; with the long-reads switch (uint32 build) the sum is exact and then capped at the read's
    # minimiser count (GanonClassify.cpp:561-564)
    k, w = 19, 31
    rng = np.random.default_rng(2)
    hb = gf.random_hibf(60, 32, 2, seed=21, density=0.05, hash_funs=2, rows=(60000, 90000))
    where = {}
    for i, b2u in enumerate(hb.bin_to_user):
        for b, u in enumerate(b2u):
            if u >= 0:
                where.setdefault(int(u), []).append((i, b))
    ub, cells = max(where.items(), key=lambda kv: len(kv[1]))
    assert len(cells) >= 2
    genome = gu.random_seq(rng, 330_000)
    hh = oracle.minimiser_hash(oracle.to_ranks(genome), k, w)
    assert 32768 < len(hh) <= 65535
    uniq = np.unique(hh)
    for i, b in cells:  # every technical bin of the user bin, and the merged bins on the way down to it
        hb.ibfs[i].emplace_many(uniq, b)
    leaf = cells[0][0]
    while leaf != 0:
        pi, pb = next((pi, pb) for pi, nx in enumerate(hb.next_ibf_id) for pb, c in enumerate(nx) if c == leaf and hb.bin_to_user[pi][pb] < 0)
        hb.ibfs[pi].emplace_many(uniq, pb)
        leaf = pi
    hb = oracle.Hibf(hb.ibfs, hb.next_ibf_id, hb.bin_to_user, hb.n_user_bins)  # (the C view of the changed matrices)
    flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    bases, off1, off2 = gu.pack_reads([genome], None)
    st = hip.HipStream(flt, 1, bases.size)
    T = max(1, oracle.threshold_rel(len(hh), 0.25))
    raw = len(cells) * len(hh)
    assert raw > 65535
    for wide in (False, True):
        st.set_long_reads(wide)
        st.submit(bases, off1, off2, k, w, 0.25)
        nh, status, mo, m = st.fetch()
        got = {int(x["target"]): int(x["count"]) for x in m}
        if wide:
            exp = {u: min(c, len(hh)) for u, c in _hibf_bulk_count_wide(hb, hh, T).items()}
            assert exp[ub] == len(hh)
        else:
            cnt = hb.bulk_count(hh, T)
            exp = {int(u): min(int(cnt[u]), len(hh)) for u in np.nonzero(cnt)[0]}
            assert exp.get(ub, 0) == ((raw & 0xFFFF) if (raw & 0xFFFF) >= T else 0) or len(cells) > 2
        assert nh[0] == len(hh) and status[0] == 0 and got == exp, (wide, got, exp)
    st.destroy()
    flt.free()


@pytest.mark.parametrize("shape", ["identity", "split", "hibf"])
def test_matches_against_an_oracle_that_never_sees_device_hashes(hip, shape):
    # the whole path on both sides: oracle minimisers -> oracle counts -> oracle selection, nothing taken from the device
    # (most other match tests hand the device's hashes to the oracle's counting, which test_minimiser_parity justifies)
    k, w = 19, 31
    rng = np.random.default_rng(404)
    genomes = [gu.random_seq(rng, 5000) for _ in range(40)]
    if shape == "hibf":
        uh = {ub * 13: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in enumerate(genomes)}
        hb = gf.random_hibf(700, 128, 2, seed=77, density=0.3, hash_funs=3, user_hashes=uh)
        flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    else:
        bins, rows, h = (4096, 3001, 4) if shape == "identity" else (1500, 4099, 3)
        ibf = gf.random_ibf(bins, rows, h, 0.3, seed=6)
        n_targets = bins if shape == "identity" else 400
        b2t = None if shape == "identity" else rng.integers(0, n_targets, size=bins).astype(np.uint32)
        for gi, g in enumerate(genomes):
            ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), gi * 37 % bins)
        flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    iupac = b"ACGTNRYKMSWBDHVacgtn"
    s1, s2 = [], []
    for i in range(4000):
        L = int(rng.choice([25, 31, 75, 100, 150, 150, 151, 250, 400]))
        if i % 3:
            g = genomes[i % len(genomes)]
            p = int(rng.integers(0, len(g) - 2 * L))
            a, b = bytearray(g[p:p + L]), bytearray(g[p + L:p + 2 * L])
        else:
            a, b = bytearray(gu.random_seq(rng, L)), bytearray(gu.random_seq(rng, int(rng.choice([10, L]))))
        for _ in range(int(rng.integers(0, 4))):
            a[int(rng.integers(0, len(a)))] = iupac[int(rng.integers(0, len(iupac)))]
        s1.append(bytes(a))
        s2.append(bytes(b))
    for paired in (False, True):
        for cutoff in (0.1, 0.6):
            bases, off1, off2 = gu.pack_reads(s1, s2 if paired else None)
            st = hip.HipStream(flt, len(s1), bases.size)
            st.submit(bases, off1, off2, k, w, cutoff)
            nh, status, mo, m = st.fetch()
            n_match = 0
            for i in range(len(s1)):
                if len(s1[i]) < w:
                    assert status[i] == 1 and mo[i] == mo[i + 1]
                    continue
                hh = oracle.minimiser_hash(oracle.to_ranks(s1[i]), k, w)
                if paired and len(s2[i]) >= w:
                    hh = np.concatenate([hh, oracle.minimiser_hash(oracle.to_ranks(s2[i]), k, w)])
                assert status[i] == 0 and nh[i] == len(hh), i
                thr = oracle.threshold_cutoff(len(hh), cutoff)
                if shape == "hibf":
                    ec = hb.bulk_count(hh, thr)
                    exp = [(int(u), int(min(c, len(hh)))) for u, c in enumerate(ec) if c > 0]
                else:
                    exp, _ = gu.oracle_matches(ibf, b2t if b2t is not None else np.arange(bins, dtype=np.uint32), n_targets, hh, cutoff)
                got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
                assert got == exp, (shape, paired, cutoff, i, got[:3], exp[:3])
                n_match += len(exp)
            assert n_match > 1000
            st.destroy()
    flt.free()


@pytest.mark.parametrize("rel_filter,fpr_query", [(0.1, 1e-5), (0.0, 1.0), (0.5, 1e-2), (1.0, 1.0)])
def test_joint_postfilter_over_the_filters_of_a_level(hip, rel_filter, fpr_query):
    # three filters with disjoint targets classify the same batch; the pre-pass uses the LEVEL's max/min per read
    # (GanonClassify.cpp:716-735,755-761), so its survivors are those of filter_matches on the union of the matches
    k, w = 19, 31
    rng = np.random.default_rng(99)
    genomes = [gu.random_seq(rng, 3000) for _ in range(30)]
    shapes = [("flat", 2048, 2003, 4), ("flat", 700, 3001, 3), ("hibf", 300, None, 3)]
    flts, n_targets, tfprs = [], [], []
    for fi, (kind, bins, rows, h) in enumerate(shapes):
        mine = [g for gi, g in enumerate(genomes) if gi % 3 == fi]
        if kind == "hibf":
            uh = {ub * 7: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in enumerate(mine)}
            hb = gf.random_hibf(bins, 64, 2, seed=11, density=0.4, hash_funs=h, user_hashes=uh)
            flts.append(hip.HipFilter.hibf(*gf.hibf_upload_args(hb)))
        else:
            ibf = gf.random_ibf(bins, rows, h, 0.4, seed=20 + fi)
            for gi, g in enumerate(mine):
                ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), gi * 17 % bins)
            flts.append(hip.HipFilter.ibf(ibf.data, bins, rows, h))
        n_targets.append(bins)
        tfprs.append(rng.choice([1e-4, 0.01, 0.05, 0.2], size=bins))
    seqs = []
    for i in range(500):
        L = int(rng.choice([100, 150, 250]))
        if i % 4:
            g = genomes[i % 30]
            p = int(rng.integers(0, 3000 - L))
            seqs.append(g[p:p + L])
        else:
            seqs.append(gu.random_seq(rng, L))
    bases, off1, off2 = gu.pack_reads(seqs, None)
    sts = [hip.HipStream(f, len(seqs), bases.size) for f in flts]
    raw = []
    for st in sts:  # unfiltered matches of every filter
        st.submit(bases, off1, off2, k, w, 0.15)
        nh, status, mo, m = st.fetch()
        raw.append([[(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]] for i in range(len(seqs))])
    for st, tf in zip(sts, tfprs):
        st.set_postfilter(rel_filter, fpr_query, tf, joint=True)
        st.submit(bases, off1, off2, k, w, 0.15)
    hip.HipStream.postfilter_joint(sts)
    got, mxs, drops = [], [], []
    for st in sts:
        nh2, _, mo2, m2 = st.fetch()
        mx, a, b = st.fetch_postfilter()
        got.append([[(int(x["target"]), int(x["count"]) & 0x7FFFFFFF, int(x["count"]) >> 31) for x in m2[int(mo2[i]):int(mo2[i + 1])]]
                    for i in range(len(seqs))])
        mxs.append(mx)
        drops.append((a, b))
    base = np.concatenate([[0], np.cumsum(n_targets)])
    all_fpr = np.concatenate(tfprs)
    e_fil = e_fpr = dev_fpr = 0
    for i in range(len(seqs)):
        union = [(int(base[f]) + t, c) for f in range(3) for t, c in raw[f][i]]
        kept, nf, nq, emx = _exact_filter_matches(union, nh[i], rel_filter, fpr_query, all_fpr)
        kept = set(kept)
        for f in range(3):
            assert int(mxs[f][i]) == emx, (i, f)
            g = [(int(base[f]) + t, c) for t, c, _ in got[f][i]]
            mine = {u for u in union if base[f] <= u[0] < base[f + 1]}
            assert kept & mine <= set(g) <= mine
            assert {(int(base[f]) + t, c) for t, c, ok in got[f][i] if ok} <= kept
            dev_fpr += len([u for u in mine if u not in set(g)])
        e_fil += nf
        e_fpr += nq
    assert sum(a for a, _ in drops) == e_fil
    assert sum(b for _, b in drops) == dev_fpr - e_fil <= e_fpr
    assert any(max(len(r) for r in raw[f]) > 20 for f in range(3))
    for st in sts:
        st.destroy()
    for f in flts:
        f.free()


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("rel_filter,fpr_query", [(0.1, 1e-5), (0.0, 1.0), (0.5, 1e-2), (1.0, 1.0)])
def test_merging_postfilter_over_filters_that_share_targets(hip, rel_filter, fpr_query, dense):
    # three filters report overlapping sets of level-wide targets; the device replays the reference's merge per read
    # (GanonClassify.cpp:531-537: larger count wins, max/min follow the entries that got in) and applies filter_matches to the
    # winners -- checked against a replay of the same rule on the unfiltered matches and the oracle's filter_matches
    import ctypes as C
    import math
    k, w = 19, 31
    rng = np.random.default_rng(7)
    genomes = [gu.random_seq(rng, 3000) for _ in range(30)]
    n_gid = 900
    # sparse filters: a read has a handful of matches (a wave merges it); dense ones: a read matches most bins of every filter,
    # 1000-4600 matches over the level (a block merges it, or -- past 4096 -- it is left to the caller)
    shapes = [(2100, 2003, 2), (2200, 3001, 2), (300, 1201, 2)] if dense else [(512, 2003, 4), (700, 3001, 3), (300, 1201, 2)]
    flts, gids, tfprs, ibfs = [], [], [], []
    n_gid = 5000 if dense else n_gid
    for fi, (bins, rows, h) in enumerate(shapes):
        ibf = gf.random_ibf(bins, rows, h, 0.5 if dense else 0.45, seed=40 + fi)
        for gi, g in enumerate(genomes):
            if (gi + fi) % 2 == 0:  # every genome sits in two of the three filters
                ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), gi)
        gid = (30 + rng.permutation(n_gid)[:bins]).astype(np.uint32)  # (a filter names every target once)
        gid[:30] = np.arange(30)  # bin gi of every filter is the level's target gi: shared
        flts.append(hip.HipFilter.ibf(ibf.data, bins, rows, h))
        gids.append(gid)
        tfprs.append(rng.choice([1e-4, 0.01, 0.05, 0.2], size=bins))
    seqs = []
    for i in range(400):
        L = int(rng.choice([100, 150, 250]))
        if i % 4:
            g = genomes[i % 30]
            p = int(rng.integers(0, 3000 - L))
            seqs.append(g[p:p + L])
        else:
            seqs.append(gu.random_seq(rng, L))
    bases, off1, off2 = gu.pack_reads(seqs, None)
    sts = [hip.HipStream(f, len(seqs), bases.size) for f in flts]
    raw = []
    for st in sts:
        st.submit(bases, off1, off2, k, w, 0.12)
        nh, status, mo, m = st.fetch()
        raw.append([[(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]] for i in range(len(seqs))])
    for st, tf, gid in zip(sts, tfprs, gids):
        st.set_postfilter(rel_filter, fpr_query, tf, target_gid=gid)
        st.submit(bases, off1, off2, k, w, 0.12)
    hip.HipStream.postfilter_joint(sts)
    got, mxs, drops = [], [], []
    for st in sts:
        _, _, mo2, m2 = st.fetch()
        mx, a, b = st.fetch_postfilter()
        got.append([{int(x["target"]): (int(x["count"]) & 0x3FFFFFFF, int(x["count"]) >> 31) for x in m2[int(mo2[i]):int(mo2[i + 1])]}
                    for i in range(len(seqs))])
        mxs.append(mx)
        drops.append((a, b))
    L_ = oracle.lib()
    L_.gno_filter_matches.restype = C.c_size_t
    e_fil = e_fpr = d_fpr = n_big = n_block = n_shared = 0
    for i in range(len(seqs)):
        n = int(nh[i])
        merged, mxc, mnc = {}, 0, n
        for f in range(3):  # the reference's loop over the filters of the level
            for t, c in raw[f][i]:
                g = int(gids[f][t])
                if c > merged.get(g, (0,))[0]:
                    n_shared += g in merged
                    merged[g] = (c, f, t)
                    mxc, mnc = max(mxc, c), min(mnc, c)
        total = sum(len(raw[f][i]) for f in range(3))
        n_block += 512 < total <= 4096
        if total > 4096:  # left to the caller
            n_big += 1
            assert all(int(mxs[f][i]) == 0x80000000 for f in range(3))
            assert all(got[f][i] == {t: (c, 0) for t, c in raw[f][i]} for f in range(3))
            continue
        assert all(int(mxs[f][i]) == mxc for f in range(3)), i
        if not merged:
            assert all(not got[f][i] for f in range(3))
            continue
        thr = mxc - int(math.ceil(float(mxc - mnc) * rel_filter))
        items = sorted(merged.items())
        counts = np.array([v[0] for _, v in items], dtype=np.uint64)
        fprs = np.array([tfprs[v[1]][v[2]] for _, v in items], dtype=np.float64)
        keep = np.zeros(len(items), dtype=np.uint8)
        L_.gno_filter_matches(counts.ctypes.data_as(C.c_void_p), fprs.ctypes.data_as(C.c_void_p), C.c_size_t(len(items)), C.c_uint64(n),
                              C.c_uint64(thr), C.c_double(fpr_query), keep.ctypes.data_as(C.c_void_p))
        e_fil += int((keep == 2).sum())
        e_fpr += int((keep == 3).sum())
        for (g, (c, f, t)), kflag in zip(items, keep):
            present = t in got[f][i]
            if kflag == 1:
                assert present and got[f][i][t][0] == c, (i, g)
            elif kflag == 2:
                assert not present
            else:
                d_fpr += not present  # dropped on the device, or left for the caller's exact check
            if present and got[f][i][t][1]:
                assert kflag == 1  # marked GN_MATCH_FPR_OK only if the exact rule keeps it
        winners = {(f, t) for _, (c, f, t) in items}
        for f in range(3):  # nothing but winners survives
            assert all((f, t) in winners for t in got[f][i])
    assert (n_big > 20 and n_block > 20) if dense else n_big == 0, (n_big, n_block)
    assert n_shared > 100 and sum(a for a, _ in drops) == e_fil and sum(b for _, b in drops) == d_fpr <= e_fpr
    for st in sts:
        st.destroy()
    for f in flts:
        f.free()


@pytest.mark.parametrize("bins,joint,split", [(4096, False, False), (9000, False, False), (4096, True, False), (4096, False, True),
                                              (9000, True, True), (32768, False, True)])
def test_count_kernel_leaves_unwritten_only_what_the_prepass_would_drop(hip, monkeypatch, bins, joint, split):
    # with a filter_matches pre-pass on the stream the fast kernel does not write bins that the --rel-filter rule is bound to
    # drop (threshold from the unit's own maximum and a lower bound of the read's minimum); switching that off
    # (switch predrop) must change nothing: survivors, their order and marks, every read's maximum, both totals.
    # 9000 bins = three column slices per read: a slice only knows its own maximum.  joint: the minimum bound is 0.
    # split: targets own one to four bins (the split-bin kernel: the waves of a read share their maxima).
    k, w = 19, 31
    rng = np.random.default_rng(123)
    genomes = [gu.random_seq(rng, 2500) for _ in range(16)]
    ibf = gf.random_ibf(bins, 1531, 3, 0.5, seed=15)  # every minimiser hits an eighth of the bins by chance
    for gi, g in enumerate(genomes):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), (gi * 577) % bins)
    n_targets, b2t = bins, None
    if split:
        b2t = np.cumsum(rng.random(bins) < 0.45).astype(np.uint32)  # consecutive bins share a target now and then
        b2t -= b2t[0]
        n_targets = int(b2t[-1]) + 1
    flt = hip.HipFilter.ibf(ibf.data, bins, 1531, 3, b2t, n_targets)
    tfpr = rng.choice([1e-4, 0.02, 0.11, 0.3], size=n_targets)
    seqs = []
    for i in range(500):
        L = int(rng.choice([60, 100, 150, 250]))
        g = genomes[i % 16]
        p = int(rng.integers(0, 2500 - L))
        seqs.append(g[p:p + L] if i % 4 else gu.random_seq(rng, L))
    bases, off1, off2 = gu.pack_reads(seqs, None)
    st = hip.HipStream(flt, len(seqs), bases.size)
    st.submit(bases, off1, off2, k, w, 0.1)
    nh, _, mo, m = st.fetch()
    assert int(mo[-1]) > 200 * len(seqs)  # hundreds of chance matches per read
    res = {}
    for rel_filter, fpr_query in ((0.1, 1e-5), (0.5, 1.0), (0.0, 1.0), (0.99, 0.5)):
        for tag in ("predrop", "plain", "bin_lists"):
            if tag == "plain":
                gu.SW.on("predrop")
            else:
                gu.SW.off("predrop")
            if tag == "bin_lists":  # split maps whose targets own consecutive bins: without the shortcut that needs no bin lists
                gu.SW.on("csr_identity")
            else:
                gu.SW.off("csr_identity")
            st.set_postfilter(rel_filter, fpr_query, tfpr, joint=joint)
            st.submit(bases, off1, off2, k, w, 0.1)
            if joint:
                hip.HipStream.postfilter_joint([st])
            _, _, mo2, m2 = st.fetch()
            mx, a, b = st.fetch_postfilter()
            res[tag] = (mo2.copy(), m2.copy(), mx.copy(), a, b)
        gu.SW.off("predrop")
        gu.SW.off("csr_identity")
        assert np.array_equal(res["predrop"][0], res["bin_lists"][0]) and np.array_equal(res["predrop"][1], res["bin_lists"][1])
        assert np.array_equal(res["predrop"][0], res["plain"][0]) and np.array_equal(res["predrop"][1], res["plain"][1])
        assert np.array_equal(res["predrop"][2], res["plain"][2]) and res["predrop"][3:] == res["plain"][3:]
        # ... and it is the reference's rule: every read against _exact_filter_matches on the unfiltered matches
        tot = 0
        for i in range(0, len(seqs), 7 if bins < 20000 else 31):   # (32 768 bins: 4 000 chance matches a read through Python lists)
            raw = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            kept, nf, nq, emx = _exact_filter_matches(raw, nh[i], rel_filter, 1.0, tfpr)
            got = [(int(x["target"]), int(x["count"]) & 0x7FFFFFFF) for x in res["predrop"][1][int(res["predrop"][0][i]):int(res["predrop"][0][i + 1])]]
            assert int(res["predrop"][2][i]) == emx and set(got) <= set(kept), i
            if fpr_query >= 1.0:
                assert got == kept, i
            tot += nf
        assert tot > 0 or rel_filter in (0.0, 0.99)
    st.destroy()
    flt.free()


@pytest.mark.parametrize("prepass", [False, True])
@pytest.mark.parametrize("limit", [200, 3000, 40000])
def test_hibf_batch_in_read_ranges_equals_one_pass(hip, monkeypatch, limit, prepass):
    # A batch with more raw (read, user bin) pairs than one radix sort (2^31 items) or the device takes is run in read ranges,
    # each with its own levels / pre-drop / sort / finish, appended in read order.  switch hibf_pair_limit=N makes that happen
    # at toy size: the result must be the one-pass result record for record -- and the oracle's.
    k, w = 19, 31
    rng = np.random.default_rng(41)
    genomes = [gu.random_seq(rng, 4000) for _ in range(50)]
    uh = {ub * 5: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in enumerate(genomes)}
    hb = gf.random_hibf(400, 64, 2, seed=31, density=0.35, hash_funs=3, user_hashes=uh)
    flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    seqs = []
    for i in range(700):
        L = int(rng.choice([60, 150, 250]))
        if i % 3:
            g = genomes[i % 50]
            p = int(rng.integers(0, 4000 - L))
            seqs.append(g[p:p + L])
        else:
            seqs.append(gu.random_seq(rng, L))
    seqs[5] = b"ACGT"
    bases, off1, off2 = gu.pack_reads(seqs, None)
    tfpr = rng.choice([1e-4, 0.01, 0.05, 0.2], size=400)

    def run():
        st = hip.HipStream(flt, len(seqs), bases.size, max_matches=64)   # tiny buffers: the regrow path is part of it
        if prepass:
            st.set_postfilter(0.1, 1e-3, tfpr)
        out = []
        for cutoff in (0.1, 0.5):
            st.submit(bases, off1, off2, k, w, cutoff)
            nh, status, mo, m = st.fetch()
            extra = st.fetch_postfilter() if prepass else None
            out.append((nh.copy(), status.copy(), mo.copy(), m.copy(), extra))
            # a second batch on the same stream (the range size of the last batch is remembered)
            st.submit(bases, off1, off2, k, w, cutoff)
            nh2, status2, mo2, m2 = st.fetch()
            assert np.array_equal(mo2, mo) and np.array_equal(m2, m)
        st.destroy()
        return out

    whole = run()
    gu.SW.on(f"hibf_pair_limit={limit}")
    split = run()
    for (nh, status, mo, m, ex), (nh2, status2, mo2, m2, ex2) in zip(whole, split):
        assert np.array_equal(nh, nh2) and np.array_equal(status, status2) and np.array_equal(mo, mo2) and np.array_equal(m, m2)
        if prepass:
            assert np.array_equal(ex[0], ex2[0]) and ex[1:] == ex2[1:]
    if not prepass:
        nh, status, mo, m, _ = whole[0]
        for r, sq in enumerate(seqs):
            if len(sq) < w:
                continue
            hh = oracle.minimiser_hash(oracle.to_ranks(sq), k, w)
            counts = hb.bulk_count(hh, oracle.threshold_cutoff(len(hh), 0.1))
            exp = [(int(u), int(min(int(counts[u]), len(hh)))) for u in np.nonzero(counts)[0]]
            assert [(int(x["target"]), int(x["count"])) for x in m[int(mo[r]):int(mo[r + 1])]] == exp, r
    assert len(whole[0][3]) > 2000
    flt.free()


@pytest.mark.parametrize("bins,rows,h,nb", [(4096, 4001, 4, 2), (8192, 3001, 3, 4), (32768, 1201, 4, 2), (16384, 1501, 2, 4), (36864, 701, 4, 2), (51264, 401, 3, 2)])
def test_packed_select_for_uniform_power_of_two_targets(hip, monkeypatch, bins, rows, h, nb):
    # every target owns the same 2 or 4 consecutive bins (what a database of equally sized, over-sized targets looks like): at low
    # cutoffs the split kernel judges them with packed 16-bit arithmetic over its bin-ordered counters.  Same matches as the general
    # scan (switch uniform_select), as the oracle, and -- with the pre-pass -- the same survivors and dropped totals.
    k, w = 19, 31
    rng = np.random.default_rng(bins + nb)
    n_targets = bins // nb
    b2t = (np.arange(bins, dtype=np.uint32) // nb).astype(np.uint32)
    ibf = gf.random_ibf(bins, rows, h, 0.5, seed=bins + h)
    genomes = []
    for gi in range(40):
        t = int(rng.integers(0, n_targets))
        g = gu.random_seq(rng, 1500)
        hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w))
        for pi, part in enumerate(np.array_split(hs, nb)):
            ibf.emplace_many(part, t * nb + pi)
        if gi % 4 == 0:
            for x in range(nb):             # all of it in every bin of the target: the sum passes n and is capped
                ibf.emplace_many(hs, t * nb + x)
        genomes.append(g)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    reads = []
    for i in range(800):
        g = genomes[i % 40]
        p = int(rng.integers(0, 1300))
        reads.append(g[p:p + 150] if i % 3 else gu.random_seq(rng, int(rng.choice([60, 150]))))
    tfpr = rng.choice([1e-4, 0.01, 0.05, 0.2], size=n_targets)
    for cutoff in (0.05, 0.15, 0.3, 0.8):
        st, nh, status, mo, m = _classify(hip, flt, reads, None, k, w, cutoff)
        ho, hs = st.fetch_hashes()
        gu.SW.on("uniform_select")
        st2, nh2, status2, mo2, m2 = _classify(hip, flt, reads, None, k, w, cutoff)
        gu.SW.off("uniform_select")
        assert np.array_equal(mo, mo2) and np.array_equal(m, m2), cutoff
        for i in range(0, len(reads), 5 if bins < 50000 else 13):
            exp_m, _ = gu.oracle_matches(ibf, b2t, n_targets, hs[int(ho[i]):int(ho[i + 1])], cutoff)
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp_m, (cutoff, i, got[:3], exp_m[:3])
        if cutoff == 0.05:
            assert np.diff(mo.astype(np.int64)).max() > 128     # reads with more hits than a wave's staging list
        st.destroy()
        st2.destroy()
        # with the filter_matches pre-pass (the select leaves unwritten what the rule is bound to drop)
        res = []
        for off in (False, True):
            if off:
                gu.SW.on("uniform_select")
            bases, off1, _ = gu.pack_reads(reads, None)
            sp = hip.HipStream(flt, len(reads), bases.size)
            sp.set_postfilter(0.1, 1e-3, tfpr)
            sp.submit(bases, off1, None, k, w, cutoff)
            r = sp.fetch()
            res.append((r[2].copy(), r[3].copy(), sp.fetch_postfilter()))
            sp.destroy()
            gu.SW.off("uniform_select")
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert np.array_equal(res[0][2][0], res[1][2][0]) and res[0][2][1:] == res[1][2][1:]
    flt.free()


@pytest.mark.parametrize("bins,rows,h,big", [(4096, 4001, 4, False), (4160, 3001, 3, True), (8192, 3001, 2, False), (32768, 1201, 4, True),
                                             (36864, 701, 5, False), (65536, 601, 3, True), (8192, 2001, 4, True), (51264, 401, 3, True)])
def test_run_select_for_targets_of_mixed_widths(hip, monkeypatch, bins, rows, h, big):
    # every bin has a target, targets own one to four consecutive bins (what ganon-build makes of targets of different sizes above
    # max_hashes_bin): at low cutoffs the split kernel judges them with a running sum over each lane's own bins (targets straddle
    # lanes, dwords and column slices).  Same matches as the general scan (switch run_select), as the oracle, and -- with the
    # pre-pass -- the same survivors and dropped totals.
    k, w = 19, 31
    rng = np.random.default_rng(bins + h)
    # (big: a few targets of more than four bins among them -- up to several lanes long; those are judged from a list)
    sizes = []
    while sum(sizes) < bins:
        pick = int(rng.choice([5, 9, 40, 70, 200])) if big and rng.random() < 0.02 else int(rng.choice([1, 1, 1, 2, 2, 3, 4]))
        sizes.append(min(pick, bins - sum(sizes)))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_targets = len(sizes)
    b2t = np.repeat(np.arange(n_targets, dtype=np.uint32), sizes)
    ibf = gf.random_ibf(bins, rows, h, 0.5, seed=bins + h)
    genomes = []
    for gi in range(48):
        # (some at the very ends of the map and around the borders of lanes and slices)
        t = [0, n_targets - 1, int(b2t[63]), int(b2t[64]), int(b2t[min(bins - 1, 64 * 64 - 1)]), int(b2t[min(bins - 1, 64 * 64)])][gi] if gi < 6 else int(rng.integers(0, n_targets))
        if big and gi in (6, 7, 8, 9):       # some of the long targets too
            t = int(np.flatnonzero(np.asarray(sizes) > 4)[gi - 6])
        g = gu.random_seq(rng, 1500)
        hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w))
        for pi, part in enumerate(np.array_split(hs, min(sizes[t], 8))):
            ibf.emplace_many(part, int(off[t]) + pi)
        if gi % 4 == 0:
            for x in range(min(sizes[t], 8)):  # all of it in every bin of the target: the sum passes n and is capped
                ibf.emplace_many(hs, int(off[t]) + x)
        genomes.append(g)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    reads = []
    for i in range(800):
        g = genomes[i % 48]
        p = int(rng.integers(0, 1300))
        reads.append(g[p:p + 150] if i % 3 else gu.random_seq(rng, int(rng.choice([60, 150]))))
    tfpr = rng.choice([1e-4, 0.01, 0.05, 0.2], size=n_targets)
    for cutoff in (0.05, 0.15, 0.3, 0.8):
        st, nh, status, mo, m = _classify(hip, flt, reads, None, k, w, cutoff)
        ho, hs = st.fetch_hashes()
        gu.SW.on("run_select")
        st2, nh2, status2, mo2, m2 = _classify(hip, flt, reads, None, k, w, cutoff)
        gu.SW.off("run_select")
        assert np.array_equal(mo, mo2) and np.array_equal(m, m2), cutoff
        for i in range(0, len(reads), 5):
            exp_m, _ = gu.oracle_matches(ibf, b2t, n_targets, hs[int(ho[i]):int(ho[i + 1])], cutoff)
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp_m, (cutoff, i, got[:3], exp_m[:3])
        if cutoff == 0.05:
            assert np.diff(mo.astype(np.int64)).max() > 128     # reads with more hits than a wave's staging list
        st.destroy()
        st2.destroy()
        # with the filter_matches pre-pass (the select leaves unwritten what the rule is bound to drop)
        res = []
        for off_ in (False, True):
            if off_:
                gu.SW.on("run_select")
            bases, off1, _ = gu.pack_reads(reads, None)
            sp = hip.HipStream(flt, len(reads), bases.size)
            sp.set_postfilter(0.1, 1e-3, tfpr)
            sp.submit(bases, off1, None, k, w, cutoff)
            r = sp.fetch()
            res.append((r[2].copy(), r[3].copy(), sp.fetch_postfilter()))
            sp.destroy()
            gu.SW.off("run_select")
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert np.array_equal(res[0][2][0], res[1][2][0]) and res[0][2][1:] == res[1][2][1:]
    flt.free()


def test_target_id_without_bins_does_not_shift_later_targets(hip):
    # bin2target [0,0,2,2,2,...] with n_targets covering an id (1, and a few more) that owns no bin: accepted by gn_filter_upload_ibf
    # (every bin has a target, bins in target order), but the run select numbers targets by counting target ends, so an empty id
    # would shift every later one -- such a map takes the general scan instead (gn_capi.hip run_ok)
    k, w, bins, rows, h = 19, 31, 4096, 1201, 4
    rng = np.random.default_rng(99)
    sizes, ids, nxt = [], [], 0
    while sum(sizes) < bins:
        if rng.random() < 0.1 or nxt == 1:
            nxt += 1                                  # this id gets no bin
        sizes.append(min(int(rng.choice([1, 2, 3, 4])), bins - sum(sizes)))
        ids.append(nxt)
        nxt += 1
    n_targets = nxt + 2                               # two unused ids at the very end as well
    b2t = np.repeat(np.asarray(ids, dtype=np.uint32), sizes)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ibf = gf.random_ibf(bins, rows, h, 0.5, seed=7)
    genomes = []
    for gi in range(32):
        t = int(rng.integers(0, len(sizes)))
        g = gu.random_seq(rng, 1500)
        hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w))
        for pi, part in enumerate(np.array_split(hs, sizes[t])):
            ibf.emplace_many(part, int(off[t]) + pi)
        genomes.append(g)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    reads = [genomes[i % 32][(i * 31) % 1300:(i * 31) % 1300 + 150] if i % 3 else gu.random_seq(rng, 150) for i in range(600)]
    for cutoff in (0.05, 0.3, 0.8):
        st, nh, status, mo, m = _classify(hip, flt, reads, None, k, w, cutoff)
        ho, hs = st.fetch_hashes()
        n_match = 0
        for i in range(0, len(reads), 2):
            exp_m, _ = gu.oracle_matches(ibf, b2t, n_targets, hs[int(ho[i]):int(ho[i + 1])], cutoff)
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp_m, (cutoff, i, got[:3], exp_m[:3])
            n_match += len(exp_m)
        assert n_match > 100
        st.destroy()
    flt.free()


@pytest.mark.parametrize("kind", ["uniform2", "uniform4", "mixed", "mixed_long"])
def test_split_kernel_switches_give_the_same_survivors(hip, monkeypatch, kind):
    # the A/B switches of the split kernel's selects (DESIGN "Switches"): the round-3 instantiation for uniform maps, no maximum before
    # the first select, the scan instead of the packed / running-sum select -- with a filter_matches pre-pass to follow every one of
    # them leaves the same matches, the same survivors and the same dropped totals
    k, w, bins, rows, h = 19, 31, 16384, 1801, 3
    rng = np.random.default_rng(len(kind))
    if kind.startswith("uniform"):
        nb = int(kind[-1])
        sizes = [nb] * (bins // nb)
    else:
        sizes = []
        while sum(sizes) < bins:
            pick = int(rng.choice([6, 30, 150])) if kind == "mixed_long" and rng.random() < 0.02 else int(rng.choice([1, 1, 2, 3, 4]))
            sizes.append(min(pick, bins - sum(sizes)))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n_targets = len(sizes)
    b2t = np.repeat(np.arange(n_targets, dtype=np.uint32), sizes)
    ibf = gf.random_ibf(bins, rows, h, 0.5, seed=77)
    genomes = []
    for gi in range(32):
        t = int(rng.integers(0, n_targets))
        g = gu.random_seq(rng, 1500)
        hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w))
        for pi, part in enumerate(np.array_split(hs, min(sizes[t], 6))):
            ibf.emplace_many(part, int(off[t]) + pi)
        genomes.append(g)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
    reads = [genomes[i % 32][(i * 37) % 1300:(i * 37) % 1300 + 150] if i % 3 else gu.random_seq(rng, 150) for i in range(600)]
    bases, off1, _ = gu.pack_reads(reads, None)
    tfpr = rng.choice([1e-4, 0.01, 0.05, 0.2], size=n_targets)
    switches = [None, "max_first", "const_nb", "uniform_select", "run_select", "split_kernel"]
    for cutoff in (0.1, 0.25, 0.75):
        res = []
        for sw in switches:
            for e in switches[1:]:
                gu.SW.off(e)
            if sw:
                gu.SW.on(sw)
            sp = hip.HipStream(flt, len(reads), bases.size)
            sp.set_postfilter(0.1, 1e-3, tfpr)
            sp.submit(bases, off1, None, k, w, cutoff)
            r = sp.fetch()
            res.append((r[2].copy(), r[3].copy(), sp.fetch_postfilter()))
            sp.destroy()
        for e in switches[1:]:
            gu.SW.off(e)
        for sw, x in zip(switches[1:], res[1:]):
            assert np.array_equal(res[0][0], x[0]) and np.array_equal(res[0][1], x[1]), (kind, cutoff, sw)
            assert np.array_equal(res[0][2][0], x[2][0]) and res[0][2][1:] == x[2][1:], (kind, cutoff, sw)
        if cutoff == 0.1:
            assert len(res[0][1]) > 0
    flt.free()


def test_deferred_launches_sized_by_the_previous_batch(hip, monkeypatch):
    # The launches that take what the fast kernels defer (reads with more than 127 minimisers, reads longer than 640 letters) size
    # their persistent grids by what the stream's PREVIOUS batch deferred.  A batch of short reads (nothing deferred) followed by a
    # batch of thousands of long reads (everything deferred) must come out like the same batches on fresh streams and with the
    # full grids -- and so must the short batch after the long one.
    k, w = 19, 31
    rng = np.random.default_rng(77)
    ibf = gf.random_ibf(256, 8191, 3, 0.4, 5)
    genome = gu.random_seq(rng, 60000)
    for b in range(256):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(genome[b * 200:b * 200 + 2200]), k, w)), b)
    flt = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs)
    short = [genome[p:p + 150] for p in rng.integers(0, 59000, size=3000)]
    long_ = [genome[p:p + int(L)] for p, L in zip(rng.integers(0, 40000, size=4000), rng.integers(1500, 4000, size=4000))]

    def run(st, reads):
        bases, off1, _ = gu.pack_reads(reads, None)
        st.submit(bases, off1, None, k, w, 0.3)
        nh, status, mo, m = st.fetch()
        ho, hs = st.fetch_hashes()
        return nh.copy(), status.copy(), mo.copy(), m.copy(), ho.copy(), hs.copy()

    cap_r, cap_b = 4000, sum(len(x) for x in long_) + 64
    reused = hip.HipStream(flt, cap_r, cap_b)
    got = [run(reused, short), run(reused, long_), run(reused, short), run(reused, long_)]
    gu.SW.on("deferred_grids")
    want = []
    for reads in (short, long_):
        fresh = hip.HipStream(flt, cap_r, cap_b)
        want.append(run(fresh, reads))
        fresh.destroy()
    gu.SW.off("deferred_grids")
    for g, wnt in zip(got, [want[0], want[1], want[0], want[1]]):
        for a, b in zip(g, wnt):
            assert np.array_equal(a, b)
    assert len(got[1][3]) > 1000 and (got[1][0] > 127).all()
    # ... and against the oracle on a sample of the long reads
    for i in range(0, len(long_), 400):
        exp_m, _ = gu.oracle_matches(ibf, np.arange(256, dtype=np.uint32), 256, oracle.minimiser_hash(oracle.to_ranks(long_[i]), k, w), 0.3)
        mo, m = got[3][2], got[3][3]
        assert [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]] == exp_m
    reused.destroy()
    flt.free()
