"""Regression tests for round 3's intermittent "the device counted zero" (DESIGN 7-5b).

Cause (named by profiles/r04_race_probe_by_runtime.jsonl): gn_filter_upload_ibf zero-filled a filter that is to be streamed
with a null-stream hipMemset, and gn_filter_write_rows copies on a stream created hipStreamNonBlocking -- no implicit order
against the null stream.  Whether that is a race depends on the HIP runtime in the process: /opt/rocm's 7.2.0 (what the
binaries and plain scripts get) completes the fill before hipMemset returns; the 7.0.2 runtime PyTorch bundles returns
before the fill has run -- and that is the runtime libganon_hip.so binds to whenever torch was imported first, as under
pytest (tests/conftest.py) and in bench.py.  There a fill still queued when the first row chunk landed wiped it: with a
4 GiB fill 6 of 9 repetitions lose their rows (round 3's sequence; 0 of 9 with the wait that is there now), with the
suite's 32 KiB filters only when the fill is slow to start (code objects loading on a fresh box): round 3's "2 of 7".

* the first test makes the window wide (a 4 GiB fill takes a millisecond or two, 128 KiB of rows land in microseconds); it
  runs under pytest, i.e. with PyTorch's runtime, where round 3's sequence fails it (profiles/r04_race_pytest_*.log);
* the third one is the failing sequence of the suite -- ganon-build -> load_ibf -> submit -> fetch -> dense tap -- as the
  first GPU work of a fresh process, >= 200 times, alternating between the two runtimes (every other child imports torch
  first), what the reference asks of a filter it just built (/root/reference/tests/ganon-build/GanonBuild.test.cpp:53-98:
  every inserted hash answers).
"""
import concurrent.futures as cf
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = os.path.join(HERE, "upload_order_child.py")


@pytest.fixture(scope="module")
def hip():
    import ganon_amd
    ganon_amd.load_library()
    assert ganon_amd.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return ganon_amd


def streamed_rows_survive(hip, S, B, n_rows, reps, settle=0.02):
    """-> number of repetitions in which rows written right behind the upload call were not what came back"""
    from ganon_amd import hip as H
    from ganon_amd import ibf_file
    L = hip.load_library()
    W = (B + 63) // 64
    pin = ibf_file._Pinned(n_rows * W * 8)
    src = pin.arr.view(np.uint64).reshape(n_rows, W)
    bad = 0
    try:
        for rep in range(reps):
            src[:] = np.uint64(0x0123456789ABCDEF) * np.uint64(rep + 1) | np.uint64(1)
            row0 = (0, S - n_rows, S // 2)[rep % 3]
            flt = hip.HipFilter.ibf(None, B, S, 4)
            H._check(L.gn_filter_write_rows(flt._h, 0, row0, n_rows, src.ctypes.data_as(C.c_void_p), W, 0))
            H._check(L.gn_filter_write_sync(flt._h))
            time.sleep(settle)  # whatever is still queued on the null stream lands now
            got = flt.download_rows(row0, n_rows, W)
            edge = flt.download_rows(0 if row0 else n_rows, 4, W)  # rows nobody wrote are zero
            flt.free()
            bad += int(not np.array_equal(got, src)) + int(edge.any())
    finally:
        pin.free()
    return bad


def test_zero_fill_of_a_streamed_filter_is_ordered_before_its_rows(hip):
    assert streamed_rows_survive(hip, S=1 << 23, B=4096, n_rows=256, reps=9) == 0   # 4 GiB fill
    assert streamed_rows_survive(hip, S=1 << 12, B=64, n_rows=16, reps=30, settle=0.0) == 0  # the suite's size of filter


def test_a_postfilter_without_fpr_table_starts_from_zeros(hip):
    # gn_stream_set_postfilter without target_fpr zero-fills its table: that fill is on the stream's own queue now
    rng = np.random.default_rng(5)
    B, S = 256, 4099
    rows = (rng.integers(0, 1 << 62, size=(S, 4), dtype=np.uint64) & rng.integers(0, 1 << 62, size=(S, 4), dtype=np.uint64))
    flt = hip.HipFilter.ibf(rows.reshape(-1), B, S, 3)
    reads = rng.integers(0, 4, size=(200, 150)).astype(np.uint8)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[reads].reshape(-1)
    off = np.arange(201, dtype=np.uint64) * np.uint64(150)
    outs = []
    for _ in range(5):
        st = hip.HipStream(flt, 200, bases.size)
        st.set_postfilter(rel_filter=0.1, fpr_query=1.0)
        st.submit(bases, off, None, 19, 31, 0.1)
        nh, status, mo, m = st.fetch()
        outs.append((mo.copy(), m.copy()))
        st.destroy()
    for mo, m in outs[1:]:
        assert np.array_equal(mo, outs[0][0]) and np.array_equal(m, outs[0][1])
    flt.free()


def test_built_filter_answers_for_every_inserted_hash_in_200_fresh_processes(hip, tmp_path):
    # (204 runs while the race was being hunted, rounds 4-5: green throughout; 96 keep the regression in the suite at half the
    # time -- the driver's whole -m gpu step has 1200 s.  $GANON_TEST_FRESH_RUNS=204 is the long form)
    runs = int(os.environ.get("GANON_TEST_FRESH_RUNS", "96"))
    combos = [(19, 32), (21, 23), (27, 27)]
    env = dict(os.environ)

    def one(i):
        k, w = combos[i % 3]
        d = tmp_path / f"r{i}"
        d.mkdir()
        p = subprocess.run([sys.executable, CHILD, str(d), str(k), str(w), "torch" if i % 2 == 0 else "plain"], capture_output=True, text=True,
                           env=env, timeout=600)
        return i, p.returncode, (p.stdout + p.stderr)[-2000:]

    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(one, range(runs)))
    failed = [(i, out) for i, rc, out in res if rc != 0]
    assert not failed, f"{len(failed)} of {runs} fresh processes failed; first: {failed[0]}"
