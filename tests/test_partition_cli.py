"""BASELINE config 5 as a feature of the product binary: a flat filter that does not fit into one device's budget is cut by
technical-bin range at target boundaries, its column parts are placed on different --device entries, every device
classifies every batch against its parts, and the sparse matches are put back together on the batch's owner device
(gn_gather) -- the reference simply loads a filter of any size (GanonClassify.cpp:949-986,1007-1039), so every output byte
must be the one the unpartitioned run (and the oracle-backend twin) writes.

One GPU on the test box: $GANON_DEVICE_BUDGET turns every --device ENTRY into a placement device with that budget, so
`--device 0,0,0` spreads the filter over three of them."""
import os
import subprocess

import pytest

import cli_util as cu
from test_cli_kat import config1, oracle_bin, wide_db  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

EXTS = (".all", ".unc", ".rep")


def _run(binary, db, out, extra=(), env=None, check=True):
    e = dict(os.environ)
    e.update(env or {})
    cutoff = [] if "--rel-cutoff" in extra else ["--rel-cutoff", "0.5"]
    p = subprocess.run([binary, "--ibf", db["ibf"], "--single-reads", db["fq"], "-o", out, "--output-all", "--output-unclassified",
                        "--quiet"] + cutoff + list(extra), capture_output=True, text=True, env=e, timeout=900)
    if check:
        assert p.returncode == 0, p.stderr
    return p


def _same(a, b, exts=EXTS):
    for ext in exts:
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext


def test_filter_over_budget_is_partitioned_and_changes_no_output_byte(oracle_bin, wide_db, tmp_path):
    # 150 000 bins x 311 rows = 5.8 MB; a budget of 2.5 MB per placement device forces a spread over three
    whole, ora = str(tmp_path / "whole"), str(tmp_path / "ora")
    _run(cu.BIN_HIP, wide_db, whole)
    _run(oracle_bin, wide_db, ora)
    _same(whole, ora)
    assert os.path.getsize(whole + ".all") > 1000
    spread = {"GANON_DEVICE_BUDGET": "2500000", "GANON_HOST_TIMING": "1"}
    for name, devs, env in (("three", "0,0,0", {}), ("four", "0,0,0,0", {}), ("copies", "0,0,0", {"GANON_HIP_ABLATE": "gather_copy"}),
                            ("batches", "0,0,0", {"GANON_HOST_BATCH_READS": "37"}),
                            ("one_worker", "0,0,0", {"GANON_PARTITION_WORKERS": "1", "GANON_HOST_BATCH_READS": "100"}),
                            # the reads as pieces of FASTQ text, records found on the device (the part that takes a batch takes the text)
                            ("text", "0,0,0", {"GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_SLAB_BYTES": "65536", "GANON_HOST_PARSE_THREADS": "3"}),
                            ("parsed", "0,0,0", {"GANON_HOST_PARALLEL_MIN": "0", "GANON_HOST_SLAB_BYTES": "65536", "GANON_HOST_PARSE_THREADS": "3",
                                                 "GANON_HOST_DEVICE_FASTQ": "0"})):
        out = str(tmp_path / name)
        p = _run(cu.BIN_HIP, wide_db, out, ["--device", devs], dict(spread, **env))
        assert "partitioned by bin range" in p.stderr and p.stderr.count("-> device 0") == len(devs.split(",")), p.stderr
        _same(out, whole)
        if "GANON_HIP_ABLATE" in env:
            assert "moved between devices" in p.stderr
            # per device pair: how the matches travelled and how many bytes (on a node with several GPUs: "peer access enabled")
            assert "[gather] level" in p.stderr and "MiB of matches gathered" in p.stderr, p.stderr[-800:]
        assert ("tokenised on the device" in p.stderr) == (name == "text"), p.stderr[-500:]
    # without the budget the same command line replicates (one copy: the entries name one GPU)
    p = _run(cu.BIN_HIP, wide_db, str(tmp_path / "repl"), ["--device", "0,0,0"], {"GANON_HOST_TIMING": "1"})
    assert "replicated on 1 device(s)" in p.stderr and "partitioned" not in p.stderr
    _same(str(tmp_path / "repl"), whole)


@pytest.mark.parametrize("thr", [["--rel-filter", "0.3", "--fpr-query", "1e-3"], ["--rel-cutoff", "0.2", "--rel-filter", "0.1", "--fpr-query", "1e-5"],
                                 ["--rel-filter", "1"]])
def test_partitioned_filter_with_the_filter_matches_prepass(oracle_bin, wide_db, tmp_path, thr):
    # the parts take part in ONE joint device pre-pass across the placement devices (per-read max/min exchanged between
    # them), survivors only are gathered; .sta carries the discarded-match totals
    ora, got = str(tmp_path / "ora"), str(tmp_path / "got")
    _run(oracle_bin, wide_db, ora, thr + ["--output-stats"])
    env = {"GANON_DEVICE_BUDGET": "2500000", "GANON_HOST_TIMING": "1", "GANON_HIP_ABLATE": "gather_copy,joint_apart"}
    p = _run(cu.BIN_HIP, wide_db, got, thr + ["--output-stats", "--device", "0,0,0"], env)
    assert "partitioned by bin range" in p.stderr and "pre-pass on the device on (1 filter(s)" in p.stderr
    _same(got, ora, EXTS + (".sta",))
    # and with the pre-pass off (everything judged on the host): the same bytes
    p = _run(cu.BIN_HIP, wide_db, got + "_host", thr + ["--output-stats", "--device", "0,0,0"], dict(env, GANON_HOST_NO_PREFILTER="1"))
    _same(got + "_host", ora, EXTS + (".sta",))


def test_level_with_a_replicated_and_a_partitioned_filter(oracle_bin, wide_db, config1, tmp_path):
    # two filters on one hierarchy level (disjoint target names): the small one fits everywhere and is replicated, the wide
    # one is spread over what the devices have left
    args = lambda out: ["--ibf", config1["ibf"] + "," + wide_db["ibf"], "--single-reads", wide_db["fq"] + "," + config1["fq"], "-o", out,  # noqa: E731
                        "--output-all", "--output-unclassified", "--output-stats", "--rel-cutoff", "0.4", "--rel-filter", "0.2", "--quiet"]
    ora, got = str(tmp_path / "ora"), str(tmp_path / "got")
    cu.run(oracle_bin, args(ora))
    env = dict(os.environ, GANON_DEVICE_BUDGET="2800000", GANON_HOST_TIMING="1", GANON_HOST_BATCH_READS="3000")
    p = subprocess.run([cu.BIN_HIP] + args(got) + ["--device", "0,0,0"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr
    assert "replicated on 1 device(s)" in p.stderr and "partitioned by bin range" in p.stderr, p.stderr
    _same(got, ora, EXTS + (".sta",))


def test_what_cannot_be_placed_fails_loudly(wide_db, tmp_path):
    p = _run(cu.BIN_HIP, wide_db, str(tmp_path / "x"), ["--device", "0,0"], {"GANON_DEVICE_BUDGET": "2000000"}, check=False)
    assert p.returncode != 0 and "does not fit" in p.stderr


def test_hibf_over_budget_is_refused(tmp_path):
    import ganon_fixtures as gf
    import numpy as np
    import oracle
    rng = np.random.default_rng(5)
    uh = {ub: np.unique(oracle.minimiser_hash(oracle.to_ranks(bytes(rng.choice(list(b"ACGT"), size=500).astype(np.uint8))), 19, 31))
          for ub in range(20)}
    hb = gf.random_hibf(20, 64, 2, seed=3, density=0.2, hash_funs=2, user_hashes=uh)
    path = str(tmp_path / "x.hibf")
    gf.write_hibf(path, hb, [[f"/x/ub{i}.minimiser"] for i in range(20)], 19, 31, 0.05)
    fq = str(tmp_path / "r.fq")
    gf.write_fastq(fq, [("r0", "ACGT" * 40)])
    p = subprocess.run([cu.BIN_HIP, "--ibf", path, "--hibf", "--single-reads", fq, "-o", str(tmp_path / "o"), "--quiet", "--device", "0,0"],
                       capture_output=True, text=True, env=dict(os.environ, GANON_DEVICE_BUDGET="1000"), timeout=900)
    assert p.returncode != 0 and "cannot be partitioned" in p.stderr, p.stderr
