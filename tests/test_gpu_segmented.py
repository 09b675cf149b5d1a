"""A flat IBF with one unit per read (W <= 64 words: BASELINE configs[1]) leaves every read's matches as one segment of the match
buffer -- written once -- instead of copying 12 bytes per match behind each other after every batch.  The
contiguous CSR form is made when a consumer asks (gn_fetch_batch(matches), gn_stream_device_matches, the merges); the filter_matches
pre-pass reads the segments where they lie.  Switch `seg_result` restores the per-batch copy of round 5: every consumer must see the
same records either way -- for reads of the fast kernel (ascending targets by construction) and for reads it defers to the generic
kernel (> 127 minimisers: candidate-driven select, any order, put right by gn_seg_order_kernel).
select_matches semantics: /root/reference/src/ganon-classify/GanonClassify.cpp:504-541; filter_matches :579-613."""
import numpy as np
import pytest

import gpu_util as gu
import bench_workload as bw

pytestmark = pytest.mark.gpu


def _consume(st, wl, cutoff, postfilter):
    import torch
    if postfilter:
        st.set_postfilter(0.1, 1e-5, np.full(wl.bins, 0.0625, dtype=np.float64))
    st.classify(wl.k, wl.w, cutoff)
    st.sync()
    t_before = st.timings()
    dev = st.device_records(0).cpu().numpy().copy()           # gn_stream_device_matches
    nh, status, mo, m = st.fetch()                             # gn_fetch_batch
    t_after = st.timings()
    extra = st.fetch_postfilter() if postfilter else None
    if postfilter:
        st.set_postfilter(None)
    return dict(nh=nh.copy(), status=status.copy(), mo=mo.copy(), m=m.copy(), dev=dev, t0=t_before, t1=t_after, pf=extra)


@pytest.mark.parametrize("read_len,n_reads", [(150, 200_000), (1200, 20_000)])
@pytest.mark.parametrize("cutoff", [0.75, 0.2])
@pytest.mark.parametrize("postfilter", [False, True])
def test_same_records_from_segments_and_from_the_per_batch_copy(read_len, n_reads, cutoff, postfilter):
    import ganon_amd
    # (Bernoulli(0.5) bits, h = 4: a minimiser hits a bin by chance with 1/16 -- at cutoff 0.2 a 150 bp read has ~100 chance matches,
    #  a 1 200 bp read a few: hundreds of millions of records would only test numpy)
    wl = bw.make_device_flat_workload("seg", 4096, 1 << 15, 4, n_reads, False, seed=5, read_len=read_len, genome_len=max(3000, 4 * read_len))
    flt, _ = bw.device_filter(ganon_amd, wl, 0)
    st = ganon_amd.HipStream(flt, n_reads, wl.bases.size, n_reads * 2)
    st.upload(wl.bases, wl.off, wl.off2)
    a = _consume(st, wl, cutoff, postfilter)
    gu.SW.on("seg_result")
    b = _consume(st, wl, cutoff, postfilter)
    gu.SW.off("seg_result")
    for k in ("nh", "status", "mo", "m", "dev"):
        assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), k
    if postfilter:
        assert a["pf"][1:] == b["pf"][1:] and (a["pf"][0] == b["pf"][0]).all()
    m, mo = a["m"], a["mo"].astype(np.int64)
    assert len(m) > n_reads // 4 and int(mo[-1]) == len(m)
    # grouped by read, ascending target inside a read -- also for the reads the generic kernel took
    assert (np.diff(m["read"].astype(np.int64)) >= 0).all()
    same = m["read"][1:] == m["read"][:-1]
    assert (m["target"][1:][same] > m["target"][:-1][same]).all()
    assert (a["dev"].reshape(-1, 3).view(np.uint32) == m.view(np.uint32).reshape(-1, 3)).all()
    # the copy is made on demand and timed apart: none before a consumer asked, none at all under the switch or after a pre-pass
    assert a["t0"]["ms_compact"] == 0.0 and b["t1"]["ms_compact"] == 0.0
    assert (a["t1"]["ms_compact"] > 0.0) == (not postfilter)
    st.destroy()
    flt.free()


def test_a_second_fetch_makes_no_second_copy_and_a_new_batch_starts_clean():
    import ganon_amd
    wl = bw.make_device_flat_workload("seg2", 4096, 1 << 14, 4, 200_000, False, seed=9)
    flt, _ = bw.device_filter(ganon_amd, wl, 0)
    st = ganon_amd.HipStream(flt, wl.n_reads, wl.bases.size, wl.n_reads * 2)
    st.upload(wl.bases, wl.off, None)
    st.classify(wl.k, wl.w, 0.3)
    first = st.fetch()
    t1 = st.timings()["ms_compact"]
    again = st.fetch()
    assert st.timings()["ms_compact"] == t1 > 0.0
    for x, y in zip(first, again):
        assert (x == y).all()
    st.classify(wl.k, wl.w, 0.75)
    st.sync()
    assert st.timings()["ms_compact"] == 0.0
    third = st.fetch()
    assert len(third[3]) < len(first[3]) and len(third[3]) > 0
    st.destroy()
    flt.free()
