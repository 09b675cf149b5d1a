"""Randomised cross-check of the three selects for split-bin maps (split kernel, generic kernel with the candidate select,
generic kernel with the per-target scan) against each other and the oracle.  Not part of the test-suite (minutes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganon_amd as hip, ganon_fixtures as gf, gpu_util as gu, oracle

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
n_cfg = int(os.environ.get("N_CFG", "30"))
bad = 0
for c in range(n_cfg):
    bins = int(rng.choice([1024, 4096, 4032, 8192, 12288, 20480, 36864]))
    h = int(rng.integers(1, 6)); fill = float(rng.uniform(0.15, 0.55)); rows = int(rng.integers(900, 4000))
    k, w = 19, 19 + int(rng.integers(0, 14))
    cutoff = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5, 0.75, 0.9, 1.0]))
    contiguous = bool(rng.integers(0, 2)); paired = bool(rng.integers(0, 2))
    choices = [[1, 1, 2], [1, 1, 1, 2, 2, 3, 4], [1, 2, 3, 4, 5, 9, 40, 300], [2, 2, 2]][int(rng.integers(0, 4))]
    full = bool(rng.integers(0, 2))  # every bin has a target (with `contiguous`: the CSR is the identity, the uniform / run selects apply)
    sizes, left = [], bins - (0 if full else int(rng.integers(0, bins // 8)))
    while left > 0:
        s = min(int(rng.choice(choices)), left); sizes.append(s); left -= s
    order = np.arange(bins) if contiguous else rng.permutation(bins)
    b2t = np.full(bins, 0xFFFFFFFF, dtype=np.uint32); tb, pos = [], 0
    for t, s in enumerate(sizes):
        b2t[order[pos:pos + s]] = t; tb.append(order[pos:pos + s]); pos += s
    nt = len(sizes)
    ibf = gf.random_ibf(bins, rows, h, fill, seed=1000 + c)
    genomes = []
    for gi in range(40):
        t = int(rng.integers(0, nt)); g = gu.random_seq(rng, 1200)
        hs = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w))
        mode = gi % 3
        if mode == 0:      # spread over the target's bins
            for pi, part in enumerate(np.array_split(hs, min(len(tb[t]), 8))):
                if len(part): ibf.emplace_many(part, int(tb[t][pi]))
        elif mode == 1:    # every hash in every bin (sums pass n)
            for b in tb[t][:6]: ibf.emplace_many(hs, int(b))
        else:              # one bin only
            ibf.emplace_many(hs, int(tb[t][0]))
        genomes.append(g)
    flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, nt)
    s1, s2 = [], []
    for i in range(1000):
        g = genomes[i % 40]; p = int(rng.integers(0, 800))
        a = bytearray(g[p:p + 150])
        for _ in range(int(rng.integers(0, 8))):
            q = int(rng.integers(0, 150)); a[q] = b"ACGT"[int(rng.integers(0, 4))]
        s1.append(bytes(a) if i % 4 else gu.random_seq(rng, 150)); s2.append(g[p + 100:p + 250])
    bases, off1, off2 = gu.pack_reads(s1, s2 if paired else None)
    out = []
    for envs in ((), ("split_kernel",), ("split_kernel", "cand_select")):
        hip.set_ablation(envs)
        st = hip.HipStream(flt, len(s1), max(bases.size, 1)); st.submit(bases, off1, off2, k, w, cutoff)
        nh, status, mo, m = st.fetch(); ho, hs = st.fetch_hashes(); out.append((mo.copy(), m.copy())); st.destroy()
    hip.set_ablation("")
    same = all(np.array_equal(out[0][0], o[0]) and np.array_equal(out[0][1], o[1]) for o in out[1:])
    okc = True
    for i in range(0, len(s1), 9):
        exp_m, _ = gu.oracle_matches(ibf, b2t, nt, hs[int(ho[i]):int(ho[i + 1])], cutoff)
        got = [(int(x["target"]), int(x["count"])) for x in out[0][1][int(out[0][0][i]):int(out[0][0][i + 1])]]
        okc &= got == exp_m
    print(f"cfg {c}: bins {bins} h {h} fill {fill:.2f} w {w} cutoff {cutoff} contiguous {contiguous} full {full} paired {paired} sizes {choices}: "
          f"targets {nt} matches {len(out[0][1])} same {same} oracle {okc}", flush=True)
    bad += (not same) or (not okc)
    flt.free()
print("BAD", bad)
sys.exit(1 if bad else 0)
