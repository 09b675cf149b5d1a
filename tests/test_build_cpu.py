"""ganon-build without a GPU: the sizing arithmetic (product C++ vs the oracle restatement, and the reference's own test
properties on the reference's own 25-genome data set) and the command line's validation (which runs before any device
is touched).  /root/reference/tests/ganon-build/GanonBuild.test.cpp is the model."""
import gzip
import math
import os
import subprocess

import numpy as np
import pytest

import oracle
from oracle import build_params as bp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN_BUILD = os.path.join(ROOT, "ganon_amd", "host", "ganon-build")
DATA = os.path.join(HERE, "golden", "build_mode")  # the reference's tests/ganon-build/data (25 genomes + mode_input.tsv)


@pytest.fixture(scope="module")
def check_bin():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host_oracle"), "-s", "build_params_check"])
    return os.path.join(HERE, "host_oracle", "build_params_check")


@pytest.fixture(scope="module")
def build_bin():
    import ganon_amd.build as b
    b.build_host()
    return BIN_BUILD


def read_fasta_gz(path):
    seqs, cur = [], []
    for line in gzip.open(path, "rt"):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip())
    if cur:
        seqs.append("".join(cur))
    return seqs


@pytest.fixture(scope="module")
def mode_counts():
    """distinct (19, 32)-minimisers per target of the reference's mode_input.tsv (defaultConfig: k 19, w 32)"""
    counts = []
    for line in open(os.path.join(DATA, "mode_input.tsv")):
        f, _ = line.rstrip("\n").split("\t")
        hs = [oracle.minimiser_hash(oracle.to_ranks(s.encode()), 19, 32) for s in read_fasta_gz(os.path.join(DATA, f))]
        counts.append(len(np.unique(np.concatenate(hs))))
    return counts


def run_check(check_bin, cases):
    text = "".join(f"{mf!r} {fs!r} {h} {mode} {len(c)} {' '.join(map(str, c))}\n" for mf, fs, h, mode, c in cases)
    out = subprocess.run([check_bin], input=text, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(cases)
    return [line.split() for line in out]


def oracle_line(mf, fs, h, mode, counts):
    cfg = bp.optimal_hashes(mf, fs, counts, h, mode)
    spans, digest = 0, 1469598103934665603
    if cfg.n_bins:
        cfg.true_max_fp, cfg.true_avg_fp = bp.true_false_positive(counts, cfg.max_hashes_bin, cfg.bin_size_bits, cfg.hash_functions)
        for t, a, b in bp.create_bin_map(cfg.max_hashes_bin, counts):
            for v in (t, a, b):
                digest = ((digest ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            spans += 1
    return [str(cfg.n_bins), str(cfg.max_hashes_bin), str(cfg.hash_functions), str(cfg.bin_size_bits), float(cfg.max_fp).hex(),
            float(cfg.true_max_fp).hex(), float(cfg.true_avg_fp).hex(), str(spans), str(digest)]


def same(a, b):
    return a[:4] == b[:4] and [float.fromhex(x) for x in a[4:7]] == [float.fromhex(x) for x in b[4:7]] and a[7:] == b[7:]


def test_product_sizing_equals_oracle_restatement(check_bin, mode_counts):
    rng = np.random.default_rng(7)
    cases = []
    shapes = [mode_counts, [50] * 10, [1], [99], [100], [101], [7, 0, 300, 12], [200_000] + [1500] * 20]
    for _ in range(14):
        n = int(rng.integers(1, 80))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            c = rng.integers(1, 5000, size=n)
        elif kind == 1:
            c = (10 ** rng.uniform(1, 4.8, size=n)).astype(np.int64)
        else:
            c = np.full(n, int(rng.integers(1, 30000)))
        shapes.append([int(x) for x in c])
    settings = ((0.05, 0.0, 0), (0.05, 0.0, 4), (0.001, 0.0, 0), (0.5, 0.0, 2), (0.01, 0.0, 5), (0.0, 1.0, 4), (0.0, 0.1, 0), (0.0, 64.0, 3))
    modes = ("avg", "smaller", "smallest", "faster", "fastest")
    for i, c in enumerate(shapes):
        for j, mode in enumerate(modes):
            for s, (mf, fs, h) in enumerate(settings):
                if i < 8 or (i + j + s) % 4 == 0:  # everything on the fixed shapes, a quarter of the grid on the random ones
                    cases.append((mf, fs, h, mode, c))
    got = run_check(check_bin, cases)
    bad = [(case[:4], g, oracle_line(*case)) for case, g in zip(cases, got) if not same(g, oracle_line(*case))]
    assert not bad, bad[:3]
    assert sum(1 for g in got if g[0] != "0") > 0.9 * len(got)


def _validate_filter_properties(cfg_max_fp, filter_size, res):
    # validate_filter (GanonBuild.test.cpp:35-46): floor(true fp * 100) <= floor(requested * 100) unless --filter-size
    if not filter_size:
        assert math.floor(res.true_max_fp * 100.0) / 100.0 <= math.floor(cfg_max_fp * 100.0) / 100.0
        assert math.floor(res.true_avg_fp * 100.0) / 100.0 <= math.floor(cfg_max_fp * 100.0) / 100.0


def _sized(counts, max_fp=0.05, filter_size=0.0, h=4, mode="avg"):
    cfg = bp.optimal_hashes(0.0 if filter_size else max_fp, filter_size, counts, h, mode)
    cfg.true_max_fp, cfg.true_avg_fp = bp.true_false_positive(counts, cfg.max_hashes_bin, cfg.bin_size_bits, cfg.hash_functions)
    return cfg


def test_reference_properties_hold_for_the_sizing(mode_counts):
    # the SECTIONs of "building indices" that are about numbers, on the oracle restatement
    ten = [50] * 10  # ten 80-bp sequences have ~50 (19,32)-minimisers each; the exact counts run in the GPU twin
    for mf in (0.05, 0.01, 0.5):
        _validate_filter_properties(mf, 0, _sized(ten, max_fp=mf))
    small, big = _sized(ten, max_fp=0.5), _sized(ten, max_fp=0.01)
    assert bp.optimal_bins(big.n_bins) * big.bin_size_bits > bp.optimal_bins(small.n_bins) * small.bin_size_bits  # :259
    a, b = _sized(ten, filter_size=0.1), _sized(ten, filter_size=1.0)
    assert bp.optimal_bins(a.n_bins) * a.bin_size_bits < bp.optimal_bins(b.n_bins) * b.bin_size_bits  # :287
    for h in (0, 2):
        cfg = _sized(ten, h=h)
        assert 1 <= cfg.hash_functions <= 5 and (h == 0 or cfg.hash_functions == h)
    # --mode on the reference's 25 genomes (:290-352)
    avg, smallest = _sized(mode_counts, max_fp=0.001, mode="avg"), _sized(mode_counts, max_fp=0.05, mode="smallest")
    _validate_filter_properties(0.001, 0, avg)
    _validate_filter_properties(0.05, 0, smallest)
    assert bp.optimal_bins(smallest.n_bins) * smallest.bin_size_bits < bp.optimal_bins(avg.n_bins) * avg.bin_size_bits
    f_avg, f_smallest, f_fastest = (_sized(mode_counts, filter_size=1.0, mode=m) for m in ("avg", "smallest", "fastest"))
    assert f_smallest.max_fp < f_avg.max_fp
    assert f_fastest.n_bins < f_avg.n_bins
    for cfg in (avg, smallest, f_avg, f_smallest, f_fastest):
        assert len(bp.create_bin_map(cfg.max_hashes_bin, mode_counts)) == cfg.n_bins


@pytest.mark.parametrize("args,msg", [
    ([], "Try 'ganon-build -h/--help' for more information."),
    (["-o", "x.ibf"], "--input-file is mandatory"),
    (["-i", "/nonexistent/input.tsv", "-o", "x.ibf"], "--input-file not found: /nonexistent/input.tsv"),
    (["-i", "EMPTY", "-o", "x.ibf"], "--input-file is empty: "),
    (["-i", "INPUT"], "--output-file is mandatory"),
    (["-i", "INPUT", "-o", "x.ibf", "-m", "/nonexistent/tmp/"], "--tmp-output-folder not found"),
    (["-i", "INPUT", "-o", "x.ibf", "--hash-functions", "6"], "--hash-functions must be <=5"),
    (["-i", "INPUT", "-o", "x.ibf", "--max-fp", "0"], "--max-fp or --filter-size is mandatory"),
    (["-i", "INPUT", "-o", "x.ibf", "-k", "32", "-w", "12"], "--window-size has to be >= --kmer-size"),
    (["-i", "INPUT", "-o", "x.ibf", "--mode", "tiny"], "Invalid --mode"),
    (["-i", "INPUT", "-o", "x.ibf", "-k", "35", "-w", "42"], "--kmer-size has to be <= 32"),
    (["-i", "INPUT", "-o", "x.ibf", "--frobnicate", "1"], "Option '--frobnicate' does not exist"),
    (["-i", "INPUT", "-o", "x.ibf", "-k"], "is missing an argument"),
    (["-iINPUT", "-ox.ibf", "-k35", "-w42"], "--kmer-size has to be <= 32"),          # attached short forms, as cxxopts takes them
    (["-i=INPUT", "-ox.ibf", "-s6"], "--hash-functions must be <=5"),
])
def test_command_line_validation(build_bin, tmp_path, args, msg):
    # Config.hpp:29-107 / GanonBuild.test.cpp "invalid" SECTIONs: rejected before anything is read or any device is used
    inp = tmp_path / "input.tsv"
    inp.write_text("a.fasta\tT1\n")
    (tmp_path / "empty.tsv").write_text("")
    args = [a.replace("INPUT", str(inp)) if "INPUT" in a else str(tmp_path / "empty.tsv") if a == "EMPTY" else a for a in args]
    p = subprocess.run([build_bin] + args, capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 1
    assert msg in p.stderr
    assert not (tmp_path / "x.ibf").exists()


def test_quiet_suppresses_validation_messages_and_help_exits_zero(build_bin, tmp_path):
    p = subprocess.run([build_bin, "-o", "x.ibf", "--quiet"], capture_output=True, text=True)
    assert p.returncode == 1 and p.stderr == ""
    for flag in ("-h", "--help", "-v", "--version"):
        p = subprocess.run([build_bin, flag], capture_output=True, text=True)
        assert p.returncode == 0 and ("Usage" in p.stderr or "version: " in p.stderr)
