"""End-to-end known-answer tests of the ganon-classify host pipeline (CLI -> file formats -> readers ->
post-processing -> writers) on the reference's scenarios.  On CPU the test-only oracle backend supplies the hot
path; the gpu-marked twins run the product binary (libganon_hip.so through the C ABI)."""
import os
import shutil

import pytest

import cli_util as cu
import ganon_fixtures as gf


@pytest.fixture(scope="module")
def files(kat, tmp_path_factory):
    return cu.KatFiles(kat, str(tmp_path_factory.mktemp("kat_files")))


@pytest.fixture(scope="module")
def oracle_bin():
    return cu.build_oracle_binary()


def _all_cases(binary, kat, files, outdir):
    for case in kat["cases"]:
        cu.check_case(binary, files, case, outdir)


def test_kat_cases_oracle_backend(oracle_bin, kat, files, tmp_path):
    _all_cases(oracle_bin, kat, files, str(tmp_path))


@pytest.mark.gpu
def test_kat_cases_hip(kat, files, tmp_path):
    assert os.path.exists(cu.BIN_HIP), "ganon-classify was not built"
    _all_cases(cu.BIN_HIP, kat, files, str(tmp_path))


def _flags_and_files(binary, kat, files, tmp):
    base = ["--ibf", files.ibf("build1"), "--single-reads", files.read("readA"), "--rel-cutoff", "0", "--rel-filter", "1",
            "--quiet"]
    # output files only when asked for (GanonClassify.test.cpp:271-315)
    p = os.path.join(tmp, "only_rep")
    cu.run(binary, base + ["-o", p])
    assert os.path.exists(p + ".rep")
    for ext in (".all", ".one", ".unc", ".sta"):
        assert not os.path.exists(p + ext), ext
    p = os.path.join(tmp, "wo_lca")
    cu.run(binary, base + ["-o", p, "--output-all", "--output-unclassified", "--output-stats"])
    assert os.path.exists(p + ".all") and os.path.exists(p + ".unc") and os.path.exists(p + ".sta")
    assert not os.path.exists(p + ".one")
    # short flags, = syntax, output dir creation (:1390-1397)
    p = os.path.join(tmp, "sub", "dir", "short")
    cu.run(binary, ["-i", files.ibf("build1"), "-r", files.read("readA"), "-c", "0", "-d", "1", "-a", "-u", "--output-prefix=" + p,
                    "--quiet"])
    res = cu.Res(p, lca_file=False)
    assert res.all["readA"] == {"A": 5, "T": 5}
    # verbose writes to stderr only (:224-251)
    p = os.path.join(tmp, "verbose")
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "--single-reads", files.read("readA"), "-o", p, "--verbose"])
    assert r.stdout == "" and "ganon-classify processed 1 sequences" in r.stderr and "--output-prefix" in r.stderr
    # exit codes (main.cpp:7-17, Config.hpp:70-172)
    assert cu.run(binary, [], check=False).returncode == 1
    assert cu.run(binary, ["-h"], check=False).returncode == 0
    assert cu.run(binary, ["--version"], check=False).returncode == 0
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "--single-reads", files.read("readA")], check=False)
    assert r.returncode == 1 and "--output-prefix is mandatory" in r.stderr
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "-o", p], check=False)
    assert r.returncode == 1 and "mandatory" in r.stderr
    r = cu.run(binary, ["--ibf", "/nonexistent.ibf", "--single-reads", files.read("readA"), "-o", p], check=False)
    assert r.returncode == 1 and "file not found" in r.stderr
    r = cu.run(binary, base + ["-o", p, "--rel-cutoff", "1.5"], check=False)
    assert r.returncode == 1 and "--rel-cutoff values should be set between 0 and 1" in r.stderr
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "--paired-reads", files.read("readA"), "-o", p], check=False)
    assert r.returncode == 1 and "even number" in r.stderr


def test_flags_and_files_oracle_backend(oracle_bin, kat, files, tmp_path):
    _flags_and_files(oracle_bin, kat, files, str(tmp_path))


@pytest.mark.gpu
def test_flags_and_files_hip(kat, files, tmp_path):
    _flags_and_files(cu.BIN_HIP, kat, files, str(tmp_path))


def _batch_reads(binary, kat, files, tmp):
    # --batch-reads == separate runs (GanonClassify.test.cpp:364-424), compared like aux::filesAreEqualSorted
    common = ["--ibf", files.ibf("build1"), "--tax", files.tax("full"), "--rel-cutoff", "0", "--rel-filter", "1", "--output-all",
              "--output-lca", "--output-unclassified", "--quiet"]
    pp = os.path.join(tmp, "paired")
    cu.run(binary, common + ["-o", pp, "--paired-reads", files.read("readA") + "," + files.read("readT")])
    ps = os.path.join(tmp, "single")
    cu.run(binary, common + ["-o", ps, "--single-reads", files.read("readC")])
    tsv = os.path.join(tmp, "batch.tsv")
    with open(tsv, "w") as f:
        f.write(f"batch_paired\t{files.read('readA')}\t{files.read('readT')}\n")
        f.write(f"batch_single\t{files.read('readC')}\n")
    pb = os.path.join(tmp, "batch")
    cu.run(binary, common + ["-o", pb, "--batch-reads", tsv])
    for ext in (".all", ".one", ".unc", ".rep"):
        assert sorted(open(pp + ext).read()) == sorted(open(pb + "batch_paired" + ext).read()), ext
        assert sorted(open(ps + ext).read()) == sorted(open(pb + "batch_single" + ext).read()), ext
    # hierarchy + batch prefixes -> one file per (prefix, label) (:459-508)
    tsv2 = os.path.join(tmp, "batch2.tsv")
    with open(tsv2, "w") as f:
        f.write(f"batchA\t{files.read('readC')}\nbatchB\t{files.read('readG')}\nbatchC\t{files.read('readA')}\t{files.read('readT')}\n")
    ph = os.path.join(tmp, "hier")
    cu.run(binary, ["--ibf", files.ibf("build1") + "," + files.ibf("build1"), "--tax", files.tax("full") + "," + files.tax("full"),
                    "--hierarchy-labels", "DB1,DB2", "--rel-cutoff", "0", "--rel-filter", "1", "--output-all", "--output-lca",
                    "--output-unclassified", "--quiet", "-o", ph, "--batch-reads", tsv2])
    for b in ("batchA", "batchB", "batchC"):
        for lab in ("DB1", "DB2"):
            assert os.path.exists(f"{ph}{b}.{lab}.all") and os.path.exists(f"{ph}{b}.{lab}.one")
        assert os.path.exists(f"{ph}{b}.unc") and os.path.exists(f"{ph}{b}.rep")
    # two levels without --output-single: both .all files non-empty (:778-793)
    p2 = os.path.join(tmp, "two_levels")
    cu.run(binary, ["--ibf", files.ibf("build1") + "," + files.ibf("build2"), "--hierarchy-labels", "one,two", "--rel-cutoff", "0",
                    "--rel-filter", "1", "--output-all", "--quiet", "-o", p2, "--single-reads",
                    files.read("readA") + "," + files.read("readCG")])
    assert os.path.getsize(p2 + ".one.all") > 0 and os.path.getsize(p2 + ".two.all") > 0


def test_batch_reads_oracle_backend(oracle_bin, kat, files, tmp_path):
    _batch_reads(oracle_bin, kat, files, str(tmp_path))


@pytest.mark.gpu
def test_batch_reads_hip(kat, files, tmp_path):
    _batch_reads(cu.BIN_HIP, kat, files, str(tmp_path))


# ---------------------------------------------------------------------------------------------------------------
# realistic inputs: the reference's 98-pair 150 bp FASTQ fixture (tests/ganon/data/classify/sim.{1,2}.fq.gz) against a
# synthetic multi-target database with split bins, flat (.ibf) and hierarchical (.hibf)
# ---------------------------------------------------------------------------------------------------------------
def _sim_reads():
    import gzip
    here = os.path.join(os.path.dirname(__file__), "golden")
    out = []
    for fn in ("sim.1.fq.gz", "sim.2.fq.gz"):
        recs = []
        with gzip.open(os.path.join(here, fn), "rt") as f:
            lines = f.read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            recs.append((lines[i][1:], lines[i + 1]))
        out.append(recs)
    return out


@pytest.fixture(scope="module")
def sim_db(tmp_path_factory):
    """Targets = 40 synthetic genomes, each containing a few of the fixture reads verbatim -> true matches."""
    import numpy as np
    d = str(tmp_path_factory.mktemp("sim_db"))
    r1, r2 = _sim_reads()
    rng = np.random.default_rng(99)
    targets = {}
    for t in range(40):
        parts = []
        for j in range(6):
            parts.append("".join("ACGT"[x] for x in rng.integers(0, 4, size=300)))
            idx = (t * 6 + j) % len(r1)
            parts.append(r1[idx][1] if j % 2 == 0 else r2[idx][1])
        targets[f"T{t}.1"] = "".join(parts)
    built = gf.build_ibf(targets, 19, 31, max_fp=0.05, filter_size=0.0)
    # force split bins: rebuild with a small max_hashes_bin through the size optimiser's mode "fastest" is not needed;
    # instead halve the capacity so that every target spans several technical bins
    ibf_path = os.path.join(d, "sim.ibf")
    gf.write_ibf(ibf_path, built)
    tax = {t: "G" + str(i % 5) for i, t in enumerate(targets)}
    tax.update({"G" + str(i): "ROOTG" for i in range(5)})
    tax["ROOTG"] = "1"
    tax_path = os.path.join(d, "sim.tax")
    gf.write_tax(tax_path, tax)
    # HIBF over the same targets
    import oracle
    uh = {i: np.unique(oracle.minimiser_hash(oracle.to_ranks(s), 19, 31)) for i, s in enumerate(targets.values())}
    hb = gf.random_hibf(len(targets), 64, 2, seed=3, density=0.02, hash_funs=3, rows=(20000, 30000), user_hashes=uh)
    hibf_path = os.path.join(d, "sim.hibf")
    names = list(targets)
    gf.write_hibf(hibf_path, hb, [[f"/some/dir/{n.replace('.', '|||')}.minimiser"] for n in names], 19, 31, 0.05)
    here = os.path.join(os.path.dirname(__file__), "golden")
    return dict(dir=d, ibf=ibf_path, hibf=hibf_path, tax=tax_path, targets=targets, built=built, hibf_obj=hb, names=names,
                fq1=os.path.join(here, "sim.1.fq.gz"), fq2=os.path.join(here, "sim.2.fq.gz"))


def _run_sim(binary, sim_db, tmp, hibf=False, extra=()):
    p = os.path.join(tmp, "hibf" if hibf else "ibf")
    args = ["--ibf", sim_db["hibf"] if hibf else sim_db["ibf"], "--tax", sim_db["tax"], "--paired-reads",
            sim_db["fq1"] + "," + sim_db["fq2"], "-o", p, "--output-all", "--output-lca", "--output-unclassified",
            "--output-stats", "--quiet", "--rel-cutoff", "0.25", "--rel-filter", "0.1"] + (["--hibf"] if hibf else []) + list(extra)
    cu.run(binary, args)
    return p


def _check_sim_against_oracle_level(sim_db, prefix, hibf):
    """.all of the run == oracle.Level (GanonClassify.cpp:676-768) on the same reads"""
    import oracle
    if hibf:
        flt = oracle.Filter(hibf=sim_db["hibf_obj"], targets=sim_db["names"], target_bins=[[i] for i in range(len(sim_db["names"]))],
                            target_fpr=[0.05] * len(sim_db["names"]), rel_cutoff=0.25)
    else:
        flt = sim_db["built"].as_filter(0.25)
    lvl = oracle.Level([flt], 19, 31, rel_filter=0.1, fpr_query=1.0)
    r1, r2 = _sim_reads()
    res = cu.Res(prefix)
    res.sanity_check(has_tax=True)
    n_class = 0
    for (rid, s1), (_, s2) in zip(r1, r2):
        rr = lvl.classify(oracle.to_ranks(s1), oracle.to_ranks(s2))
        if rr.kept:
            n_class += 1
            assert res.all.get(rid) == rr.kept, (rid, res.all.get(rid), rr.kept)
        else:
            assert rid not in res.all and rid in res.unc
    assert res.total_classified == n_class and n_class > 20
    assert res.total_classified + res.total_unclassified == len(r1)


def test_sim_fastq_gz_ibf_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_sim_against_oracle_level(sim_db, _run_sim(oracle_bin, sim_db, str(tmp_path)), False)


def test_sim_fastq_gz_hibf_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_sim_against_oracle_level(sim_db, _run_sim(oracle_bin, sim_db, str(tmp_path), hibf=True), True)


@pytest.mark.gpu
@pytest.mark.parametrize("hibf", [False, True])
def test_sim_hip_equals_oracle_backend_bytes(oracle_bin, sim_db, tmp_path, hibf):
    # same host code, different hot path: every output file must be byte-identical
    a = _run_sim(cu.BIN_HIP, sim_db, str(tmp_path / "hip"), hibf=hibf)
    b = _run_sim(oracle_bin, sim_db, str(tmp_path / "ora"), hibf=hibf)
    _check_sim_against_oracle_level(sim_db, a, hibf)
    for ext in (".all", ".one", ".unc", ".rep", ".sta"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[0]: 64-bin IBF (k=19, w=31, h=3), 10 k synthetic 150 bp reads -- the plumbing case
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config1(tmp_path_factory):
    import struct

    import numpy as np
    import oracle
    d = str(tmp_path_factory.mktemp("config1"))
    rng = np.random.default_rng(2024)
    genomes = ["".join("ACGT"[x] for x in rng.integers(0, 4, size=2000)) for _ in range(64)]
    ibf = oracle.Ibf(64, 40009, 3)
    hc = []
    for b, g in enumerate(genomes):
        hv = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), 19, 31))
        ibf.emplace_many(hv, b)
        hc.append((f"G{b}", len(hv)))
    built = gf.BuiltIbf()
    built.ibf = ibf
    built.config = dict(n_bins=64, max_hashes_bin=max(c for _, c in hc), hash_functions=3, kmer_size=19, window_size=31,
                        bin_size_bits=40009, max_fp=0.05, true_max_fp=0.05, true_avg_fp=0.05)
    built.hashes_count = hc
    built.bin_map = [(b, f"G{b}") for b in range(64)]
    gf.write_ibf(os.path.join(d, "c1.ibf"), built)
    recs = []
    for i in range(10000):
        if i % 2:
            g = genomes[i % 64]
            p = int(rng.integers(0, 1850))
            recs.append((f"r{i}", g[p:p + 150]))
        else:
            recs.append((f"r{i}", "".join("ACGT"[x] for x in rng.integers(0, 4, size=150))))
    gf.write_fastq(os.path.join(d, "reads.fq"), recs)
    return dict(dir=d, ibf=os.path.join(d, "c1.ibf"), fq=os.path.join(d, "reads.fq"), built=built, recs=recs)


def _run_config1(binary, c1, out):
    cu.run(binary, ["--ibf", c1["ibf"], "--single-reads", c1["fq"], "-o", out, "--output-all", "--output-unclassified",
                    "--output-stats", "--quiet"])  # reference defaults: rel-cutoff 0.2, rel-filter 0, fpr-query 1
    return out


def test_config1_oracle_backend(oracle_bin, config1, tmp_path):
    import oracle
    p = _run_config1(oracle_bin, config1, str(tmp_path / "c1"))
    res = cu.Res(p, lca_file=False)
    res.sanity_check(output_lca=False)
    assert res.total_classified + res.total_unclassified == 10000 and res.total_classified >= 5000
    lvl = oracle.Level([config1["built"].as_filter(0.2)], 19, 31, rel_filter=0.0, fpr_query=1.0)
    for rid, seq in config1["recs"][:400]:
        rr = lvl.classify(oracle.to_ranks(seq))
        assert res.all.get(rid, {}) == rr.kept, rid


@pytest.mark.gpu
def test_config1_hip_equals_oracle_backend(oracle_bin, config1, tmp_path):
    a = _run_config1(cu.BIN_HIP, config1, str(tmp_path / "hip"))
    b = _run_config1(oracle_bin, config1, str(tmp_path / "ora"))
    for ext in (".all", ".unc", ".rep", ".sta"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext
