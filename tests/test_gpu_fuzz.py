"""Bounded, fixed-seed samples of the randomised cross-checks (scripts/fuzz_early_exit.py, scripts/fuzz_split.py) and a
1 M-read split-bin case: a wrong early exit for split-bin maps once passed every small parity test and was only caught by a
full-size run, so the randomised and the large shapes belong to the GPU suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bench_workload as bw
import gpu_util as gu
import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,n_cfg", [("fuzz_early_exit.py", 11, 12), ("fuzz_early_exit.py", 12, 8), ("fuzz_split.py", 21, 12),
                                               ("fuzz_split.py", 22, 8), ("fuzz_predrop.py", 31, 16)])
def test_randomised_cross_checks(script, seed, n_cfg):
    env = dict(os.environ, SEED=str(seed), N_CFG=str(n_cfg))
    env.pop("GANON_HIP_ABLATE", None)
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
    for envs in ((), ("split_kernel",), ("split_kernel", "cand_select")):
        for e in envs:
            gu.SW.on(e)
        st.classify(wl.k, wl.w, 0.5)
        nh, status, mo, m = st.fetch()
        outs.append((mo.copy(), m.copy()))
        for e in envs:
            gu.SW.off(e)
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


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_hibf_randomised_layouts(seed, monkeypatch):
    # random raptor-style layouts (1-3 levels, 64..640 technical bins per IBF, 1-5 hash functions, split user bins, mixed
    # lane widths inside one level), random cutoffs, reads of 40..2500 bp, pairs: the three HIBF kernel paths agree with
    # each other and with the oracle's counting_agent_type::bulk_count
    import ganon_amd as hip
    import ganon_fixtures as gf
    import gpu_util as gu
    rng = np.random.default_rng(seed)
    for cfg in range(8):
        n_ub = int(rng.integers(30, 2500))
        tmax = int(rng.choice([64, 64, 128, 192, 256, 640]))
        depth = int(rng.integers(1, 4))
        h = int(rng.integers(1, 6))
        cutoff = float(rng.choice([0.0, 0.1, 0.25, 0.5, 0.75, 1.0]))
        k = int(rng.choice([19, 21, 15]))
        w = k + int(rng.integers(0, 14))
        genomes = {ub: gu.random_seq(rng, 2600) for ub in range(0, n_ub, max(1, n_ub // 60))}
        uh = {ub: np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for ub, g in genomes.items()}
        hb = gf.random_hibf(n_ub, tmax, depth, seed=seed * 100 + cfg, density=float(rng.uniform(0.1, 0.45)), hash_funs=h, user_hashes=uh)
        flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
        keys = sorted(genomes)
        s1, s2 = [], []
        for i in range(300):
            L = int(rng.choice([40, 100, 150, 150, 250, 900, 2500]))
            g = genomes[keys[i % len(keys)]]
            p = int(rng.integers(0, max(1, len(g) - L)))
            a = bytearray(g[p:p + L]) if i % 3 else bytearray(gu.random_seq(rng, L))
            for _ in range(int(rng.integers(0, 6))):
                a[int(rng.integers(0, len(a)))] = b"ACGTN"[int(rng.integers(0, 5))]
            s1.append(bytes(a))
            s2.append(g[max(0, p - 50):max(0, p - 50) + 150])
        paired = bool(cfg % 2)
        bases, off1, off2 = gu.pack_reads(s1, s2 if paired else None)
        outs = []
        for sw in (None, "hibf_pack", "hibf_reg", "hibf_one_pack"):
            if sw:
                gu.SW.on(sw)
            st = hip.HipStream(flt, len(s1), max(bases.size, 1))
            st.submit(bases, off1, off2, k, w, cutoff)
            nh, status, mo, m = st.fetch()
            ho, hs = st.fetch_hashes()
            outs.append((mo.copy(), m.copy(), st.timings()["algo_bytes"]))
            st.destroy()
            if sw:
                gu.SW.off(sw)
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) and outs[0][2] == o[2], (seed, cfg)
        mo, m, _ = outs[0]
        for i in range(0, len(s1), 3):
            hh = hs[int(ho[i]):int(ho[i + 1])]
            if status[i] != 0:
                continue
            thr = oracle.threshold_cutoff(len(hh), cutoff)
            ec = hb.bulk_count(hh, thr)
            exp = [(int(u), int(min(c, len(hh)))) for u, c in enumerate(ec) if c > 0]
            got = [(int(x["target"]), int(x["count"])) for x in m[int(mo[i]):int(mo[i + 1])]]
            assert got == exp, (seed, cfg, i, got[:4], exp[:4])
        flt.free()
