"""Randomised differential test of the product binary: the same command line through `ganon-classify` (HIP hot path, with
and without the device-side pre-pass of filter_matches) and through the oracle-backend twin (same host code, CPU oracle as
the hot path) must write identical files -- over random hierarchies (several filters per level, several levels, targets
shared between filters), per-filter cutoffs, per-level --rel-filter / --fpr-query, single and paired reads, IBF and HIBF."""
import os

import numpy as np
import pytest

import cli_util as cu
import ganon_fixtures as gf
import oracle

pytestmark = pytest.mark.gpu
K, W = 19, 31


@pytest.fixture(scope="module")
def oracle_bin():
    return cu.build_oracle_binary()


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("fuzz"))
    rng = np.random.default_rng(2025)

    def rnd(n):
        return "".join("ACGT"[x] for x in rng.integers(0, 4, size=n))

    genomes = {f"S{t}": rnd(int(rng.integers(1500, 4000))) for t in range(30)}
    names = list(genomes)
    ibfs, hibfs = [], []
    for i in range(3):  # flat filters over overlapping subsets of the genomes, different sizing each
        sub = [n for n in names if rng.random() < 0.55] or names[:5]
        built = gf.build_ibf({n: genomes[n] for n in sub}, K, W, max_fp=[0.05, 0.01, 0.2][i], hash_functions=[0, 3, 2][i],
                             mode=["avg", "smallest", "fastest"][i])
        p = os.path.join(d, f"f{i}.ibf")
        gf.write_ibf(p, built)
        ibfs.append(p)
    for i in range(2):
        sub = [n for n in names if rng.random() < 0.6] or names[:6]
        uh = {j: np.unique(oracle.minimiser_hash(oracle.to_ranks(genomes[n].encode()), K, W)) for j, n in enumerate(sub)}
        hb = gf.random_hibf(len(sub), 64, 2, seed=40 + i, density=0.03, hash_funs=3, rows=(9000, 16000), user_hashes=uh)
        p = os.path.join(d, f"h{i}.hibf")
        gf.write_hibf(p, hb, [[f"/x/{n}.minimiser"] for n in sub], K, W, [0.05, 0.2][i])
        hibfs.append(p)
    # ... and filters with DISJOINT targets (the joint device pre-pass of filter_matches runs on levels made of these)
    d_ibfs, d_hibfs = [], []
    thirds = [names[0::3], names[1::3], names[2::3]]
    for i, sub in enumerate(thirds):
        built = gf.build_ibf({n: genomes[n] for n in sub}, K, W, max_fp=[0.05, 0.1, 0.02][i], hash_functions=[3, 0, 4][i])
        p = os.path.join(d, f"d{i}.ibf")
        gf.write_ibf(p, built)
        d_ibfs.append(p)
    for i, sub in enumerate([names[0::2], names[1::2]]):
        uh = {j: np.unique(oracle.minimiser_hash(oracle.to_ranks(genomes[n].encode()), K, W)) for j, n in enumerate(sub)}
        hb = gf.random_hibf(len(sub), 64, 2, seed=60 + i, density=0.03, hash_funs=3, rows=(9000, 16000), user_hashes=uh)
        p = os.path.join(d, f"dh{i}.hibf")
        gf.write_hibf(p, hb, [[f"/x/{n}.minimiser"] for n in sub], K, W, [0.05, 0.1][i])
        d_hibfs.append(p)
    tax = {n: f"G{i % 6}" for i, n in enumerate(names)}
    tax.update({f"G{i}": f"F{i % 2}" for i in range(6)})
    tax.update({"F0": "1", "F1": "1"})
    tax_path = os.path.join(d, "t.tax")
    gf.write_tax(tax_path, tax)

    def mutate(s):
        s = list(s)
        for _ in range(int(rng.integers(0, 7))):
            s[int(rng.integers(0, len(s)))] = "ACGTN"[int(rng.integers(0, 5))]
        return "".join(s)

    r1, r2 = [], []
    for i in range(700):
        L = int(rng.choice([60, 100, 150, 150, 250]))
        if i % 4:
            g = genomes[names[int(rng.integers(0, 30))]]
            p = int(rng.integers(0, len(g) - L))
            a, b = mutate(g[p:p + L]), mutate(g[max(0, p - 40):max(0, p - 40) + L])
        else:
            a, b = rnd(L), rnd(int(rng.choice([20, L])))
        r1.append((f"read{i} some description", a))
        r2.append((f"read{i}/2", b))
    fq1, fq2, fa = os.path.join(d, "r.1.fq"), os.path.join(d, "r.2.fq"), os.path.join(d, "r.fa")
    gf.write_fastq(fq1, r1)
    gf.write_fastq(fq2, r2)
    gf.write_fasta(fa, r1[:300])
    return dict(dir=d, ibfs=ibfs, hibfs=hibfs, d_ibfs=d_ibfs, d_hibfs=d_hibfs, tax=tax_path, fq1=fq1, fq2=fq2, fa=fa)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GANON_FUZZ_SEEDS", "24"))))  # (soak runs: GANON_FUZZ_SEEDS=100)
def test_random_hierarchies_hip_equals_oracle_backend(oracle_bin, world, tmp_path, monkeypatch, seed):
    rng = np.random.default_rng(1000 + seed)
    hibf = bool(seed % 3 == 2)
    disjoint = 12 <= seed < 20 or seed % 4 == 1  # filters with disjoint targets: several per level still get the device pre-pass
    pool = world[("d_" if disjoint else "") + ("hibfs" if hibf else "ibfs")]
    n_f = len(pool) if seed >= 12 else int(rng.integers(1, len(pool) + 1))
    files = [pool[int(x)] for x in rng.permutation(len(pool))[:n_f]]
    labels = sorted(str(int(x)) for x in rng.integers(1, 3, size=n_f))  # one or two levels, possibly several filters each
    if seed >= 16:
        labels = ["1"] * n_f  # every filter on one level
    levels = sorted(set(labels))
    args = ["--ibf", ",".join(files), "--hierarchy-labels", ",".join(labels)]  # (vectors are comma-separated, as with cxxopts)
    args += ["--rel-cutoff", ",".join(repr(float(rng.choice([0.0, 0.05, 0.2, 0.5, 0.8]))) for _ in files)]
    args += ["--rel-filter", ",".join(repr(float(rng.choice([0.0, 0.1, 0.5, 1.0]))) for _ in levels)]
    args += ["--fpr-query", ",".join(repr(float(rng.choice([1.0, 0.5, 1e-2, 1e-5]))) for _ in levels)]
    if rng.random() < 0.5:
        args += ["--paired-reads", world["fq1"] + "," + world["fq2"]]
    elif rng.random() < 0.5:
        args += ["--single-reads", world["fq1"]]
    else:
        args += ["--single-reads", world["fa"] + "," + world["fq2"]]
    if rng.random() < 0.7:
        args += ["--tax", ",".join([world["tax"]] * n_f)]  # one per filter
    else:
        args += ["--skip-lca"]
    if rng.random() < 0.3:
        args += ["--output-single"]
    if hibf:
        args += ["--hibf"]
    args += ["--output-all", "--output-lca", "--output-unclassified", "--output-stats", "--quiet"]
    outs = {}
    # "pieces": the same command line with the input cut into many small pieces -- uncompressed single-end FASTQ then reaches the
    # device as text (records found there, pieces accepted in file order), everything else goes through the slab parsers, and
    # every worker thread keeps two batches in flight
    pieces = {"GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_SLAB_BYTES": "65536", "GANON_HOST_PARSE_THREADS": "3", "GANON_HOST_BATCH_READS": "211"}
    for tag, binary, env in (("hip", cu.BIN_HIP, None), ("host_only", cu.BIN_HIP, "1"), ("pieces", cu.BIN_HIP, None), ("oracle", oracle_bin, None)):
        d = tmp_path / tag
        d.mkdir()
        for k_, v_ in pieces.items():
            if tag == "pieces":
                monkeypatch.setenv(k_, v_)
            else:
                monkeypatch.delenv(k_, raising=False)
        if env:
            monkeypatch.setenv("GANON_HOST_NO_PREFILTER", env)
        else:
            monkeypatch.delenv("GANON_HOST_NO_PREFILTER", raising=False)
        if tag == "hip":
            monkeypatch.setenv("GANON_HOST_TIMING", "1")
        p = cu.run(binary, args + ["-o", str(d / "o")])
        monkeypatch.delenv("GANON_HOST_TIMING", raising=False)
        if tag == "hip" and seed >= 16:  # every filter on one level: the joint device pre-pass must be what ran (seeds 20..: the
            # filters share targets, so the device also replays the level's merge)
            kind = "disjoint" if disjoint else "shared between filters"
            assert f"pre-pass on the device on ({n_f} filter(s), targets {kind})" in p.stderr, p.stderr[-500:]
        outs[tag] = {f: open(d / f, "rb").read() for f in sorted(os.listdir(d))}
    assert list(outs["hip"]) == list(outs["oracle"]) == list(outs["host_only"]) == list(outs["pieces"]) and len(outs["hip"]) >= 2
    for f in outs["hip"]:
        assert outs["hip"][f] == outs["oracle"][f], (seed, f, args)
        assert outs["hip"][f] == outs["host_only"][f], (seed, f, args)
        assert outs["pieces"][f] == outs["oracle"][f], (seed, f, args)
    assert any(len(v) > 0 for f, v in outs["hip"].items() if f.endswith(".all") or ".all" in f or f.endswith(".rep"))
