"""Randomised check of the count kernels' pre-drop (matches that a filter_matches pre-pass is bound to drop are not written,
gn_kernels.hip / gn_split.hip / gn_hibf.hip): with the switch `predrop` (gn_ablate) the same batch must give the same survivors in the same
order with the same marks, the same maxima and the same two totals.  Flat filters with identity, consecutive and permuted
split-bin maps, one to nine column slices per read, joint and single mode; HIBFs of random layout.  Not part of the test-suite
as a whole (tests/test_gpu_fuzz.py runs bounded samples).   env: SEED, N_CFG"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganon_amd as hip, ganon_fixtures as gf, gpu_util as gu, oracle

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
n_cfg = int(os.environ.get("N_CFG", "20"))
bad = 0
for c in range(n_cfg):
    k, w = 19, 19 + int(rng.integers(0, 14))
    cutoff = float(rng.choice([0.05, 0.1, 0.2, 0.3, 0.5]))
    rel_filter = float(rng.choice([0.0, 0.1, 0.3, 0.7, 0.99]))
    fpr_query = float(rng.choice([1.0, 1e-2, 1e-5]))
    joint = bool(rng.integers(0, 2))
    is_hibf = c % 4 == 3
    genomes = [gu.random_seq(rng, 2000) for _ in range(12)]
    if is_hibf:
        n_ub = int(rng.choice([300, 900, 2500]))
        uh = {int(u): np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)) for u, g in zip(rng.permutation(n_ub)[:12], genomes)}
        hb = gf.random_hibf(n_ub, int(rng.choice([64, 128])), 2, seed=int(rng.integers(1, 1000)), density=float(rng.uniform(0.3, 0.5)),
                            hash_funs=int(rng.integers(2, 5)), user_hashes=uh)
        flt = hip.HipFilter.hibf(*gf.hibf_upload_args(hb))
        n_targets, desc = n_ub, f"hibf {n_ub} user bins"
    else:
        bins = int(rng.choice([1024, 4096, 4032, 9000, 20480, 36864]))
        h = int(rng.integers(1, 6)); fill = float(rng.uniform(0.3, 0.55)); rows = int(rng.integers(900, 3000))
        kind = str(rng.choice(["identity", "consecutive", "permuted"]))
        ibf = gf.random_ibf(bins, rows, h, fill, seed=int(rng.integers(1, 1000)))
        for gi, g in enumerate(genomes):
            ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), k, w)), int(rng.integers(0, bins)))
        b2t, n_targets = None, bins
        if kind != "identity":
            runs = np.cumsum(rng.random(bins) < 0.5).astype(np.uint32)
            runs -= runs[0]
            n_targets = int(runs[-1]) + 1
            b2t = runs if kind == "consecutive" else runs[rng.permutation(bins)]
        flt = hip.HipFilter.ibf(ibf.data, bins, rows, h, b2t, n_targets)
        desc = f"bins {bins} h {h} fill {fill:.2f} {kind}"
    tfpr = rng.choice([1e-4, 0.02, 0.11, 0.3], size=n_targets)
    seqs = []
    for i in range(300):
        L = int(rng.choice([60, 100, 150, 250, 700]))
        g = genomes[i % 12]
        p = int(rng.integers(0, 2000 - min(L, 1900)))
        seqs.append(g[p:p + L] if i % 3 else gu.random_seq(rng, L))
    bases, off1, off2 = gu.pack_reads(seqs, None)
    st = hip.HipStream(flt, len(seqs), bases.size)
    res = {}
    for tag in ("predrop", "plain"):
        hip.set_ablation("predrop" if tag == "plain" else "")
        st.set_postfilter(rel_filter, fpr_query, tfpr, joint=joint)
        st.submit(bases, off1, off2, k, w, cutoff)
        if joint:
            hip.HipStream.postfilter_joint([st])
        _, _, mo, m = st.fetch()
        mx, a, b = st.fetch_postfilter()
        res[tag] = (mo.copy(), m.copy(), mx.copy(), int(a), int(b))
    hip.set_ablation("")
    same = (np.array_equal(res["predrop"][0], res["plain"][0]) and np.array_equal(res["predrop"][1], res["plain"][1])
            and np.array_equal(res["predrop"][2], res["plain"][2]) and res["predrop"][3:] == res["plain"][3:])
    print(f"cfg {c}: {desc} w {w} cutoff {cutoff} rel_filter {rel_filter} fpr_query {fpr_query} joint {joint}: survivors {len(res['plain'][1])} "
          f"dropped {res['plain'][3]} + {res['plain'][4]} same {same}", flush=True)
    bad += not same
    st.destroy()
    flt.free()
print("BAD", bad)
sys.exit(1 if bad else 0)
