"""Randomised cross-check of the fast kernel's early exit / narrowing: many (bins, h, fill, cutoff, read length, paired)
combinations, results with the exit == results without it == oracle on a sample.  Not part of the test-suite (minutes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganon_amd as hip, ganon_fixtures as gf, gpu_util as gu, oracle

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
n_cfg = int(os.environ.get("N_CFG", "40"))
bad = 0
for c in range(n_cfg):
    bins = int(rng.choice([4096, 8192, 12288, 20480, 4032]))
    h = int(rng.integers(1, 6)); fill = float(rng.uniform(0.15, 0.6)); rows = int(rng.integers(1500, 6000))
    k = int(rng.choice([19, 15, 21, 31])); w = k + int(rng.integers(0, 16))
    cutoff = float(rng.choice([0.1, 0.25, 0.5, 0.6, 0.75, 0.8, 0.9, 1.0]))
    paired = bool(rng.integers(0, 2)); L = int(rng.integers(max(w, 60), 320))
    ibf = gf.random_ibf(bins, rows, h, fill, seed=c)
    genomes = [gu.random_seq(rng, 1500) for _ in range(40)]
    for gi, g in enumerate(genomes):
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), int(rng.integers(0, bins)))
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h)
    s1, s2 = [], []
    for i in range(1200):
        g = genomes[i % 40]; p = int(rng.integers(0, 1500 - 2 * L)) if 1500 > 2 * L + 1 else 0
        a = bytearray(g[p:p + L]); b = bytearray(g[p + L // 2:p + L // 2 + L])
        for _ in range(int(rng.integers(0, 10))):
            q = int(rng.integers(0, len(a))); a[q] = b"ACGT"[int(rng.integers(0, 4))]
        if i % 4 == 0:
            a = bytearray(gu.random_seq(rng, L))
        s1.append(bytes(a)); s2.append(bytes(b))
    bases, off1, off2 = gu.pack_reads(s1, s2 if paired else None)
    res = []
    for env in (None, "1"):
        hip.set_ablation("early_exit" if env else "")
        st = hip.HipStream(flt, len(s1), max(bases.size, 1)); st.submit(bases, off1, off2, k, w, cutoff)
        nh, status, mo, m = st.fetch(); tm = st.timings(); ho, hs = st.fetch_hashes(); res.append((mo.copy(), m.copy(), tm)); st.destroy()
    hip.set_ablation("")
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    b2t = np.arange(bins, dtype=np.uint32); okc = True
    for i in range(0, len(s1), 7):
        exp_m, _ = gu.oracle_matches(ibf, b2t, bins, hs[int(ho[i]):int(ho[i + 1])], cutoff)
        got = [(int(x["target"]), int(x["count"])) for x in res[0][1][int(res[0][0][i]):int(res[0][0][i + 1])]]
        okc &= got == exp_m
    frac = res[0][2]["fetched_bytes"] / max(1, res[0][2]["algo_bytes"])
    print(f"cfg {c}: bins {bins} h {h} fill {fill:.2f} k {k} w {w} cutoff {cutoff} paired {paired} L {L}: matches {len(res[0][1])} "
          f"fetched {frac:.3f} same {same} oracle {okc}", flush=True)
    bad += (not same) or (not okc)
    flt.free()
print("BAD", bad)
sys.exit(1 if bad else 0)
