"""Units handed out on demand (fast count kernel, HIBF packed kernel) against the static hand-out the switch `on_demand` restores:
the same records, read for read, on batches large enough for the cursor to be used (64 units a wave / 8 batches a wave and more),
and on a batch below that (both runs take the static path: the switch must change nothing either)."""
import numpy as np
import pytest

import gpu_util as gu
import bench_workload as bw

pytestmark = pytest.mark.gpu


def _fetch(st, wl):
    st.classify(wl.k, wl.w, wl.rel_cutoff)
    nh, status, mo, m = st.fetch()
    return nh.copy(), status.copy(), mo.copy(), m.copy()


def _same(a, b):
    for x, y in zip(a, b):
        assert x.shape == y.shape and (x == y).all()


@pytest.mark.parametrize("n_reads,paired", [(1_200_000, False), (700_000, True), (50_000, False)])
def test_flat_same_records(n_reads, paired):
    import ganon_amd
    wl = bw.make_device_flat_workload("od", 1024, 1 << 16, 4, n_reads, paired, seed=7)
    flt, _ = bw.device_filter(ganon_amd, wl, 0)
    st = ganon_amd.HipStream(flt, n_reads, wl.bases.size, n_reads * 2)
    st.upload(wl.bases, wl.off, wl.off2)
    a = _fetch(st, wl)
    gu.SW.on("on_demand")
    b = _fetch(st, wl)
    gu.SW.off("on_demand")
    gu.SW.on("early_exit")       # (the every-row build of the kernel hands its units out the same way)
    c = _fetch(st, wl)
    gu.SW.off("early_exit")
    _same(a, b)
    _same(a, c)
    assert len(a[3]) > n_reads // 4  # planted reads are found
    st.destroy()
    flt.free()


@pytest.mark.parametrize("n_reads", [3_000_000, 100_000])
def test_hibf_same_records(n_reads):
    import ganon_amd
    wl, flt = bw.make_hibf_device_workload(ganon_amd, "od_hibf", 4096, 64, 1 << 12, 1 << 12, 3, n_reads, seed=11)
    st = ganon_amd.HipStream(flt, n_reads, wl.bases.size, n_reads * 2)
    st.upload(wl.bases, wl.off, None)
    a = _fetch(st, wl)
    gu.SW.on("on_demand")
    b = _fetch(st, wl)
    gu.SW.off("on_demand")
    _same(a, b)
    assert len(a[3]) > 0
    st.destroy()
    flt.free()


def test_hibf_skewed_tree_same_records():
    # several width classes a level: the largest one is run last and handed out on demand
    import ganon_amd
    n_reads = 2_000_000
    wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, "od_skew", 8192, 3, n_reads, seed=5, rows_scale=0.01)
    st = ganon_amd.HipStream(flt, n_reads, wl.bases.size, n_reads * 2)
    st.upload(wl.bases, wl.off, None)
    a = _fetch(st, wl)
    gu.SW.on("on_demand")
    b = _fetch(st, wl)
    gu.SW.off("on_demand")
    _same(a, b)
    assert len(a[3]) > 0
    st.destroy()
    flt.free()
