"""Bin-range partitioned flat IBF through the C ABI (gn_gather, gn_streams_postfilter_joint over several devices):
the column parts of a filter, each a flat IBF of its own, classify the same batch; put back together on the owner
device they must give what the unpartitioned filter gives -- and that is checked against the oracle as well.

One GPU here, so the parts share it; $switch gather_copy / $switch joint_apart make the library treat them as if
they sat on different devices (device-to-device copies of offsets, matches and per-read max/min through the same
hipMemcpyPeerAsync calls the multi-GPU placement uses)."""
import numpy as np
import pytest

import ganon_fixtures as gf
import gpu_util as gu
import oracle

pytestmark = pytest.mark.gpu

K, W = 19, 31


@pytest.fixture(scope="module")
def hip():
    import ganon_amd
    ganon_amd.load_library()
    assert ganon_amd.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return ganon_amd


def _case(seed, bins=9000, rows=2503, h=3, density=0.35, n_reads=400):
    rng = np.random.default_rng(seed)
    ibf = gf.random_ibf(bins, rows, h, density, seed=seed)
    b2t = np.full(bins, 0xFFFFFFFF, dtype=np.uint32)
    b, t = 0, 0
    while b < bins:                       # target t owns a run of 1..3 bins, a few bins belong to nobody
        if rng.random() < 0.03:
            b += 1
            continue
        run = min(int(rng.choice([1, 1, 1, 2, 3])), bins - b)
        b2t[b:b + run] = t
        b += run
        t += 1
    genomes = [gu.random_seq(rng, 2000) for _ in range(40)]
    for gi, g in enumerate(genomes):
        hv = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), K, W))
        tb = np.nonzero(b2t == (gi * 131) % t)[0]
        ibf.emplace_many(hv, int(tb[0]))
    seqs = []
    for i in range(n_reads):
        L = int(rng.choice([40, 100, 150, 250]))
        if i % 3:
            g = genomes[i % len(genomes)]
            p = int(rng.integers(0, 2000 - L))
            seqs.append(g[p:p + L])
        else:
            seqs.append(gu.random_seq(rng, L))
    seqs[7] = b"ACGT"                      # shorter than the window
    return ibf, b2t, t, seqs


def _parts(hip, ibf, b2t, world):
    from ganon_amd import partition as gp
    out = []
    for sl in gp.plan_partition(b2t, ibf.bins, world):
        rows = gp.slice_rows(ibf.data, ibf.bin_words, sl)
        flt = hip.HipFilter.ibf(rows.reshape(-1), sl.bins_local, ibf.bin_size, ibf.hash_funs, sl.bin2target_local,
                                max(1, len(sl.targets_global)))
        out.append((flt, sl))
    return out


def _oracle_matches(ibf, b2t, n_targets, seqs, rel_cutoff):
    exp = []
    for r, s in enumerate(seqs):
        if len(s) < W:
            continue
        hv = oracle.minimiser_hash(oracle.to_ranks(s), K, W)
        m, _ = gu.oracle_matches(ibf, b2t, n_targets, hv, rel_cutoff)
        exp += [(r, t, c) for t, c in m]
    return exp


@pytest.mark.parametrize("copy_path", [False, True])
@pytest.mark.parametrize("rel_cutoff", [0.1, 0.6])
def test_gather_of_column_parts_equals_the_whole_filter_and_the_oracle(hip, monkeypatch, rel_cutoff, copy_path):
    ibf, b2t, n_targets, seqs = _case(seed=3)
    bases, off1, _ = gu.pack_reads(seqs, None)
    full = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, n_targets)
    st = hip.HipStream(full, len(seqs), bases.size)
    st.submit(bases, off1, None, K, W, rel_cutoff)
    nh, status, mo, m_full = st.fetch()
    exp = _oracle_matches(ibf, b2t, n_targets, seqs, rel_cutoff)
    assert [(int(x["read"]), int(x["target"]), int(x["count"])) for x in m_full] == exp and len(exp) > 100
    if copy_path:
        gu.SW.on("gather_copy")
    for world in (2, 3, 7):
        parts = _parts(hip, ibf, b2t, world)
        sts = [hip.HipStream(f, len(seqs), bases.size) for f, _ in parts]
        for s in sts:
            s.submit(bases, off1, None, K, W, rel_cutoff)
        g = hip.HipGather(0, [sl.targets_global for _, sl in parts])
        g.run(sts)
        mo2, m2 = g.fetch()
        assert np.array_equal(mo2, mo) and np.array_equal(m2, m_full), world
        moved = g.peer_bytes()
        assert (moved > 0) == copy_path
        if copy_path:
            assert moved == sum((len(seqs) + 1) * 8 + 12 * len(s.fetch()[3]) for s in sts)
        # a second batch through the same objects (buffers are reused, results replaced)
        half = len(seqs) // 2
        b2, o2, _ = gu.pack_reads(seqs[:half], None)
        for s in sts:
            s.submit(b2, o2, None, K, W, rel_cutoff)
        g.run(sts)
        mo3, m3 = g.fetch()
        assert np.array_equal(mo3, mo[:half + 1]) and np.array_equal(m3, m_full[:int(mo[half])])
        g.destroy()
        for s in sts:
            s.destroy()
        for f, _ in parts:
            f.free()
    st.destroy()
    full.free()


@pytest.mark.parametrize("apart", [False, True])
@pytest.mark.parametrize("rel_filter,fpr_query", [(0.1, 1e-5), (0.0, 1.0), (0.5, 1e-2), (1.0, 1.0)])
def test_prepass_over_parts_then_gather_equals_the_prepass_on_the_whole_filter(hip, monkeypatch, rel_filter, fpr_query, apart):
    # the whole filter applies filter_matches' rules with the read's own max/min; the parts must arrive at the same
    # survivors, flags, per-read maxima and dropped-match totals through the joint pass (per-read max/min exchanged between
    # the parts' devices) followed by the gather
    ibf, b2t, n_targets, seqs = _case(seed=5, density=0.45)
    rng = np.random.default_rng(1)
    tfpr = rng.choice([1e-4, 0.01, 0.05, 0.2], size=n_targets)
    bases, off1, _ = gu.pack_reads(seqs, None)
    full = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, n_targets)
    st = hip.HipStream(full, len(seqs), bases.size)
    st.set_postfilter(rel_filter, fpr_query, tfpr)
    st.submit(bases, off1, None, K, W, 0.15)
    nh, status, mo, m_full = st.fetch()
    mx, d_fil, d_fpr = st.fetch_postfilter()
    if apart:
        gu.SW.on("joint_apart")
        gu.SW.on("gather_copy")
    for world in (2, 5):
        parts = _parts(hip, ibf, b2t, world)
        sts = [hip.HipStream(f, len(seqs), bases.size) for f, _ in parts]
        for s, (_, sl) in zip(sts, parts):
            s.set_postfilter(rel_filter, fpr_query, tfpr[sl.targets_global] if len(sl.targets_global) else np.zeros(1), joint=True)
            s.submit(bases, off1, None, K, W, 0.15)
        hip.HipStream.postfilter_joint(sts)
        g = hip.HipGather(0, [sl.targets_global for _, sl in parts])
        g.run(sts)
        mo2, m2 = g.fetch()
        assert np.array_equal(mo2, mo) and np.array_equal(m2, m_full), world
        a = b = 0
        for s in sts:
            mx2, x, y = s.fetch_postfilter()
            assert np.array_equal(mx2, mx)
            a += x
            b += y
        assert (a, b) == (d_fil, d_fpr)
        g.destroy()
        for s in sts:
            s.destroy()
        for f, _ in parts:
            f.free()
    assert len(m_full) > 50 and (rel_filter == 1.0 or d_fil > 0 or rel_filter == 0.0)
    st.destroy()
    full.free()


def test_gather_refuses_what_it_cannot_do(hip):
    ibf, b2t, n_targets, seqs = _case(seed=8, bins=700, rows=997, n_reads=50)
    bases, off1, _ = gu.pack_reads(seqs, None)
    parts = _parts(hip, ibf, b2t, 2)
    sts = [hip.HipStream(f, len(seqs), bases.size) for f, _ in parts]
    g = hip.HipGather(0, [sl.targets_global for _, sl in parts])
    with pytest.raises(hip.GanonHipError):      # nothing classified yet
        g.run(sts)
    with pytest.raises(hip.GanonHipError):      # nothing gathered yet
        g.fetch()
    sts[0].submit(bases, off1, None, K, W, 0.5)
    b2, o2, _ = gu.pack_reads(seqs[:10], None)
    sts[1].submit(b2, o2, None, K, W, 0.5)
    with pytest.raises(hip.GanonHipError):      # different batches
        g.run(sts)
    with pytest.raises(hip.GanonHipError):      # wrong number of parts
        g.run(sts[:1])
    for s in sts:
        s.set_postfilter(0.1, 1.0, None, joint=True)
        s.submit(bases, off1, None, K, W, 0.5)
    with pytest.raises(hip.GanonHipError):      # the joint pass has not run
        g.run(sts)
    hip.HipStream.postfilter_joint(sts)
    g.run(sts)
    free, total = hip.device_memory(0)
    assert 0 < free <= total
    g.destroy()
    for s in sts:
        s.destroy()
    for f, _ in parts:
        f.free()


def test_gather_of_raw_device_buffers_with_offsets_of_any_origin(hip):
    # gn_gather_run_buffers: what the one-process-per-GPU exchange hands over -- per source rank a slice of a longer offset
    # array (its origin is not 0) and the records of the owner's reads only
    import torch
    ibf, b2t, n_targets, seqs = _case(seed=13)
    bases, off1, _ = gu.pack_reads(seqs, None)
    full = hip.HipFilter.ibf(ibf.data, ibf.bins, ibf.bin_size, ibf.hash_funs, b2t, n_targets)
    st = hip.HipStream(full, len(seqs), bases.size)
    st.submit(bases, off1, None, K, W, 0.2)
    nh, status, mo, m_full = st.fetch()
    parts = _parts(hip, ibf, b2t, 3)
    res = []
    for f, sl in parts:
        s = hip.HipStream(f, len(seqs), bases.size)
        s.submit(bases, off1, None, K, W, 0.2)
        res.append(s.fetch())
        # the device views the exchange sends from: offsets and records as tensors aliasing the library's buffers
        d_off, d_rec = s.device_offsets(0), s.device_records(0)
        assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), res[-1][2])
        assert np.array_equal(d_rec.cpu().numpy().view(np.uint32).reshape(-1, 3).view(hip.MATCH_DTYPE).reshape(-1), res[-1][3])
        s.destroy()
    g = hip.HipGather(0, [sl.targets_global for _, sl in parts])
    for lo, hi in ((0, len(seqs)), (len(seqs) // 3, 2 * len(seqs) // 3), (len(seqs) - 5, len(seqs)), (7, 7)):
        offs, recs = [], []
        for _, _, mo_p, m_p in res:
            offs.append(torch.from_numpy(mo_p[lo:hi + 1].astype(np.int64)).cuda())
            chunk = np.ascontiguousarray(m_p[int(mo_p[lo]):int(mo_p[hi])]).view(np.uint32).reshape(-1, 3).view(np.int32)
            recs.append(torch.from_numpy(chunk.copy()).cuda())
        torch.cuda.synchronize()
        g.run_buffers([t.data_ptr() for t in offs], [t.data_ptr() if t.numel() else 0 for t in recs], [t.shape[0] for t in recs], hi - lo)
        mo2, m2 = g.fetch()
        assert np.array_equal(mo2, mo[lo:hi + 1] - mo[lo]) and np.array_equal(m2, m_full[int(mo[lo]):int(mo[hi])]), (lo, hi)
    g.destroy()
    st.destroy()
    full.free()
    for f, _ in parts:
        f.free()


@pytest.mark.parametrize("paired", [False, True])
def test_streams_that_share_one_hashed_batch(hip, paired):
    # gn_stream_classify_shared: the batch is uploaded and hashed on ONE stream; the other filters of the device count the same
    # hashes (the reference hashes a read once and hands the hashes to every filter's agent, GanonClassify.cpp:693-735)
    ibf, b2t, n_targets, seqs = _case(seed=21)
    mates = [s[::-1] for s in seqs] if paired else None
    bases, off1, off2 = gu.pack_reads(seqs, mates)
    parts = _parts(hip, ibf, b2t, 3)
    hb = gf.random_hibf(60, 32, 2, seed=4, density=0.3, hash_funs=3)
    hflt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
    filters = [f for f, _ in parts] + [hflt]
    own = []
    for f in filters:                       # every filter with an upload of its own
        st = hip.HipStream(f, len(seqs), bases.size)
        st.submit(bases, off1, off2, K, W, 0.2)
        own.append(st.fetch())
        st.destroy()
    sts = [hip.HipStream(f, len(seqs), bases.size) for f in filters]
    for rep in range(2):                    # twice: the second batch is a shorter one
        n = len(seqs) if rep == 0 else len(seqs) // 3
        b2, o1, o2 = gu.pack_reads(seqs[:n], mates[:n] if paired else None)
        sts[0].submit(b2, o1, o2, K, W, 0.2)
        for st in sts[1:]:
            st.classify_shared(sts[0], 0.2)
        for st, (nh, status, mo, m) in zip(sts, own):
            nh2, status2, mo2, m2 = st.fetch()
            assert np.array_equal(nh2, nh[:n]) and np.array_equal(status2, status[:n])
            assert np.array_equal(mo2, mo[:n + 1]) and np.array_equal(m2, m[:int(mo[n])])
        assert sts[1].timings()["n_hashes"] == sts[0].timings()["n_hashes"] > 0
    # the source is what holds the hashes; a sharing stream is no source, and an upload makes a stream its own again
    with pytest.raises(hip.GanonHipError):
        sts[2].classify_shared(sts[1], 0.2)
    with pytest.raises(hip.GanonHipError):
        sts[1].fetch_hashes()
    sts[1].submit(bases, off1, off2, K, W, 0.2)
    assert np.array_equal(sts[1].fetch()[3], own[1][3])
    sts[2].classify_shared(sts[1], 0.2)
    assert np.array_equal(sts[2].fetch()[3], own[2][3])
    # with the joint pre-pass and the gather on top: the same result as with separate uploads
    tfpr = np.full(n_targets, 0.05)
    res = []
    for shared in (False, True):
        ps = [hip.HipStream(f, len(seqs), bases.size) for f, _ in parts]
        for i, (st, (_, sl)) in enumerate(zip(ps, parts)):
            st.set_postfilter(0.2, 1e-3, tfpr[sl.targets_global] if len(sl.targets_global) else np.zeros(1), joint=True)
            if shared and i:
                st.classify_shared(ps[0], 0.2)
            else:
                st.submit(bases, off1, off2, K, W, 0.2)
        hip.HipStream.postfilter_joint(ps)
        g = hip.HipGather(0, [sl.targets_global for _, sl in parts])
        g.run(ps)
        res.append(g.fetch())
        g.destroy()
        for st in ps:
            st.destroy()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and len(res[0][1]) > 20
    for st in sts:
        st.destroy()
    for f in filters:
        f.free()
