"""Bounded, fixed-seed samples of the randomised cross-checks (scripts/fuzz_early_exit.py, scripts/fuzz_split.py) and a
1 M-read split-bin case: a wrong early exit for split-bin maps once passed every small parity test and was only caught by a
full-size run, so the randomised and the large shapes belong to the GPU suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bench_workload as bw
import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,n_cfg", [("fuzz_early_exit.py", 11, 12), ("fuzz_early_exit.py", 12, 8), ("fuzz_split.py", 21, 12),
                                               ("fuzz_split.py", 22, 8)])
def test_randomised_cross_checks(script, seed, n_cfg):
    env = dict(os.environ, SEED=str(seed), N_CFG=str(n_cfg))
    for k in ("GANON_HIP_NO_EARLY_EXIT", "GANON_HIP_NO_SPLIT_KERNEL", "GANON_HIP_NO_CAND_SELECT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "BAD 0" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_split_bins_one_million_reads(monkeypatch):
    # 4096 technical bins / 2048 targets of two bins each, 1 GiB filter, 1 M reads: the three selects agree on every match
    # and a sample agrees with the oracle (rows fetched from the device)
    import ganon_amd
    bins, rows, n = 4096, 1 << 21, 1_000_000
    wl = bw.make_device_flat_workload("split1m", bins, rows, 4, n, seed=31)
    b2t = (np.arange(bins, dtype=np.uint32) // 2).astype(np.uint32)
    flt, _ = bw.device_filter(ganon_amd, wl, bin2target=b2t, n_targets=bins // 2)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2)
    st.upload(wl.bases, wl.off, None)
    outs = []
    for envs in ((), ("GANON_HIP_NO_SPLIT_KERNEL",), ("GANON_HIP_NO_SPLIT_KERNEL", "GANON_HIP_NO_CAND_SELECT")):
        for e in envs:
            monkeypatch.setenv(e, "1")
        st.classify(wl.k, wl.w, 0.5)
        nh, status, mo, m = st.fetch()
        outs.append((mo.copy(), m.copy()))
        for e in envs:
            monkeypatch.delenv(e)
    for mo2, m2 in outs[1:]:
        assert np.array_equal(outs[0][0], mo2) and np.array_equal(outs[0][1], m2)
    mo, m = outs[0]
    assert len(m) >= n // 2
    # planted reads report their genome's TARGET (bin // 2) with the full count
    pl = np.nonzero(wl.planted_genome >= 0)[0]
    want = wl.genome_bins[wl.planted_genome[pl]] // 2
    # (at cutoff 0.5 a read also collects a few chance matches: look the wanted (read, target) pair up in the sorted records)
    keys = m["read"].astype(np.uint64) << np.uint64(32) | m["target"].astype(np.uint64)
    wk = pl.astype(np.uint64) << np.uint64(32) | want.astype(np.uint64)
    pos = np.minimum(np.searchsorted(keys, wk), len(keys) - 1)
    found = (keys[pos] == wk) & (m["count"][pos] == nh[pl])
    assert found.all()
    ibf = bw.sampled_oracle_ibf(flt, wl)
    rng = np.random.default_rng(5)
    for r in np.unique(rng.integers(0, n, size=1500)).tolist():
        hh = oracle.minimiser_hash(oracle.to_ranks(wl.bases[int(wl.off[r]):int(wl.off[r + 1])]), wl.k, wl.w)
        c = ibf.bulk_count(hh).astype(np.int64)
        sums = np.minimum(c[0::2] + c[1::2], len(hh))          # per target: sum over its bins, capped (GanonClassify.cpp:516-527)
        thr = oracle.threshold_cutoff(len(hh), 0.5)
        exp = [(int(t), int(sums[t])) for t in np.nonzero(sums >= thr)[0]]
        got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[r]):int(mo[r + 1])]]
        assert got == exp, (r, got, exp)
    st.destroy()
    flt.free()
