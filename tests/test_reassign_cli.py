"""f-4: `ganon-reassign` (host/reassign.cpp + csrc/gn_reassign.hip) against the vectors the reference's own reassign.py
produced (tests/golden/reassign/, scripts/make_reassign_golden.py) -- .one, .rep and the log, byte for byte.

* CPU: the text side through tests/host_oracle/ganon-reassign-oracle (the same host sources linked against a plain-loop
  checker of the gn_reassign_* calls);
* GPU: the product binary (EM on the device) on the same vectors, and the gn_reassign_* calls through the C ABI against
  oracle/reassign.py on seeded tables: diffs bit for bit, counts, choices; long entry lists (one wave per read), more
  targets than the LDS histogram holds, a million reads.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import reassign as orr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "reassign")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)))
BIN_HIP = os.path.join(ROOT, "ganon_amd", "host", "ganon-reassign")
BIN_CHECK = os.path.join(ROOT, "tests", "host_oracle", "ganon-reassign-oracle")


@pytest.fixture(scope="module")
def checker():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host_oracle"), "-s", "ganon-reassign-oracle"])
    return BIN_CHECK


def run_case(binary, case, tmp_path, extra=()):
    """the golden maker's call (input prefix `in`, output prefix `out`, relative, so that the log holds the same paths)"""
    d = tmp_path / case
    d.mkdir()
    for fn in os.listdir(os.path.join(GOLD, case)):
        if fn.startswith("in"):
            shutil.copy(os.path.join(GOLD, case, fn), d / fn)
    cfg = json.load(open(os.path.join(GOLD, case, "cfg.json")))
    p = subprocess.run([binary, "-i", "in", "-o", "out", "-e", str(cfg["max_iter"]), "-s", repr(cfg["threshold"])] + list(extra),
                       cwd=d, capture_output=True, text=True, timeout=600)
    return d, cfg, p


def assert_equals_golden(d, case, p):
    assert p.returncode == 0, p.stderr
    want = {fn: open(os.path.join(GOLD, case, fn)).read() for fn in os.listdir(os.path.join(GOLD, case)) if fn.startswith("out")}
    got = {fn: open(d / fn).read() for fn in os.listdir(d) if fn.startswith("out")}
    assert got == want
    assert p.stderr == open(os.path.join(GOLD, case, "log.txt")).read()
    assert p.stdout == ""


@pytest.mark.parametrize("case", CASES)
def test_text_side_equals_the_reference(checker, case, tmp_path):
    d, _, p = run_case(checker, case, tmp_path)
    assert_equals_golden(d, case, p)


def test_cli_behaviour(checker, tmp_path):
    # missing .all -> False -> exit 1 (reassign.py:52-59); no .rep at all -> nothing to do, exit 0; --skip-one / --skip-rep /
    # --remove-all / --quiet; no -o: the .rep is overwritten in place (config.py:766)
    d, _, p = run_case(checker, "sim_default", tmp_path)
    os.remove(d / "in.all")
    q = subprocess.run([checker, "-i", "in", "-o", "x"], cwd=d, capture_output=True, text=True)
    assert q.returncode == 1 and "No matching files for given .rep [in*.all]" in q.stderr
    q = subprocess.run([checker, "-i", "nothing_here"], cwd=d, capture_output=True, text=True)
    assert q.returncode == 0 and q.stderr == "Reassigning reads\n\n"
    assert subprocess.run([checker], cwd=d, capture_output=True).returncode == 2
    assert subprocess.run([checker, "-h"], cwd=d, capture_output=True).returncode == 0

    d2 = tmp_path / "flags"
    d2.mkdir()
    for fn in ("in.rep", "in.all"):
        shutil.copy(os.path.join(GOLD, "sim_default", fn), d2 / fn)
    q = subprocess.run([checker, "-i", "in", "-o", "o1", "--skip-one", "--quiet"], cwd=d2, capture_output=True, text=True)
    assert q.returncode == 0 and q.stderr == "" and not (d2 / "o1.one").exists()
    assert open(d2 / "o1.rep").read() == open(os.path.join(GOLD, "sim_default", "out.rep")).read()
    q = subprocess.run([checker, "-i", "in", "-o", "o2", "--skip-rep"], cwd=d2, capture_output=True, text=True)
    assert q.returncode == 0 and not (d2 / "o2.rep").exists() and "New .rep file" not in q.stderr
    assert open(d2 / "o2.one").read() == open(os.path.join(GOLD, "sim_default", "out.one")).read()
    q = subprocess.run([checker, "--input-prefix=in", "--remove-all"], cwd=d2, capture_output=True, text=True)
    assert q.returncode == 0 and not (d2 / "in.all").exists()
    assert open(d2 / "in.rep").read() == open(os.path.join(GOLD, "sim_default", "out.rep")).read()
    assert open(d2 / "in.one").read() == open(os.path.join(GOLD, "sim_default", "out.one")).read()
    # a damaged .all: Python raises (ValueError in the unpacking), the binary says which line and exits 1
    (d2 / "bad.rep").write_text(open(os.path.join(GOLD, "sim_default", "in.rep")).read())
    (d2 / "bad.all").write_text("r1\tT1\t5\nr2\tT1\n")
    q = subprocess.run([checker, "-i", "bad"], cwd=d2, capture_output=True, text=True)
    assert q.returncode == 1 and "line 2" in q.stderr


def test_several_rep_files_under_one_prefix(checker, tmp_path):
    # reassign.py:19-25: with several .rep files the output prefix is a PREFIX of the file's own stem
    for stem, case in (("runA", "sim_default"), ("runB", "syn_ties")):
        shutil.copy(os.path.join(GOLD, case, "in.rep"), tmp_path / f"{stem}.rep")
        shutil.copy(os.path.join(GOLD, case, "in.all"), tmp_path / f"{stem}.all")
    q = subprocess.run([checker, "-i", "run", "-o", "new_"], cwd=tmp_path, capture_output=True, text=True)
    assert q.returncode == 0, q.stderr
    for stem, case in (("runA", "sim_default"), ("runB", "syn_ties")):
        assert open(tmp_path / f"new_{stem}.one").read() == open(os.path.join(GOLD, case, "out.one")).read()
        assert open(tmp_path / f"new_{stem}.rep").read() == open(os.path.join(GOLD, case, "out.rep")).read()
    # a directory as the prefix: every .rep in it (util.py:174-177)
    sub = tmp_path / "only_inputs"
    sub.mkdir()
    for stem in ("runA", "runB"):
        for ext in (".rep", ".all"):
            shutil.copy(tmp_path / (stem + ext), sub / (stem + ext))
    q = subprocess.run([checker, "-i", str(sub), "-o", str(tmp_path / "dir_")], capture_output=True, text=True)
    assert q.returncode == 0 and (tmp_path / "dir_runA.one").exists() and (tmp_path / "dir_runB.rep").exists()
    # (the outputs of the first call are .rep files without tables: visiting them is a failure in the reference too, :52-59)
    q = subprocess.run([checker, "-i", str(tmp_path)], capture_output=True, text=True)
    assert q.returncode == 1 and "No matching files for given .rep" in q.stderr


def test_round_repr_of_the_log_is_pythons(checker):
    # the iteration lines print str(round(diff, 6)): the binary's own rendering on values of every shape, through a tiny .all
    # whose first iteration's diff is known: 2 targets, 1 unique read of 3 -> covered by the vectors; here the renderer alone
    src = os.path.join(ROOT, "tests", "host_oracle", "_round_repr_main.cpp")
    exe = os.path.join(ROOT, "tests", "host_oracle", "round_repr_check")
    open(src, "w").write('#include "../../ganon_amd/host/reassign.hpp"\n#include <cstdio>\n#include <cstdlib>\n'
                         'int main(int c, char** v) { for (int i = 1; i < c; ++i) std::puts(gnhost::py_round6_repr(std::strtod(v[i], nullptr)).c_str()); }\n')
    try:
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", exe, src,
                               os.path.join(ROOT, "ganon_amd", "host", "reassign.cpp"), os.path.join(ROOT, "tests", "host_oracle", "reassign_checker.cpp")])
        rng = np.random.default_rng(3)
        vals = [0.0, 1.0, 2.0, 0.5, 1e-7, 4e-7, 5e-7, 5.000001e-7, 1e-6, 1.5e-6, 9.9999995e-5, 1e-4, 0.000123456, 0.816327, 1.0000005, 1.9999995,
                0.1 + 0.2, 123456.789, 1e16, 2.5e-6, 3.5e-6] + list(rng.random(200)) + list(rng.random(100) * 1e-4) + list(rng.random(50) * 2)
        out = subprocess.run([exe] + [repr(float(v)) for v in vals], capture_output=True, text=True, check=True).stdout.split("\n")[:-1]
        assert out == [str(round(float(v), 6)) for v in vals]
    finally:
        for f in (src, exe):
            if os.path.exists(f):
                os.remove(f)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def hip():
    import ganon_amd
    ganon_amd.load_library()
    assert ganon_amd.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    assert os.path.exists(BIN_HIP), "ganon-reassign is built by __graft_entry__.build()"
    return ganon_amd


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_reassign_equals_the_reference(hip, case, tmp_path):
    d, _, p = run_case(BIN_HIP, case, tmp_path)
    assert_equals_golden(d, case, p)


def random_table(seed, n_reads, n_targets, p_unique, max_deg, skew=3.0, heavy=0):
    rng = np.random.default_rng(seed)
    deg = np.where(rng.random(n_reads) < p_unique, 1, rng.integers(2, max_deg + 1, size=n_reads))
    if heavy:
        deg[rng.choice(n_reads, size=heavy, replace=False)] = rng.integers(33, 700, size=heavy)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(deg, out=off[1:])
    w = rng.random(n_targets) ** skew + 1e-3
    tgt = rng.choice(n_targets, size=int(off[-1]), p=w / w.sum()).astype(np.int64)
    return off, tgt


def oracle_table(off, tgt, n_targets):
    return orr.Table([""] * (len(off) - 1), [""] * n_targets, off, tgt, np.zeros(len(tgt), dtype=np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_reads,n_targets,p_unique,max_deg,heavy,max_iter,threshold", [
    (1, 1, 1, 1.0, 2, 0, 10, 0.0),            # one read, one target
    (2, 5000, 40, 0.5, 6, 0, 10, 0.0),
    (3, 5000, 40, 0.0, 6, 0, 10, 0.0),        # nothing unique: every probability starts at zero (:238 falls to the first entry)
    (4, 20000, 3, 0.2, 3, 0, 0, 0.0),         # unbounded iterations, few hot targets (LDS histogram contention)
    (5, 30000, 700, 0.3, 30, 40, 25, 0.0),    # reads long enough for a wave of their own
    (6, 30000, 20000, 0.3, 8, 10, 10, 0.0),   # more targets than the LDS histogram holds: global atomics
    (7, 30000, 4096, 0.3, 8, 0, 10, 1e-3),    # exactly the LDS histogram's capacity; a threshold above zero
    (9, 30000, 4097, 0.3, 8, 0, 10, 0.0),     # one more
    (10, 200000, 2, 0.1, 5, 0, 10, 0.0),      # two targets: whole waves agree on one (the ballot path of the adds)
    (8, 1 << 20, 5000, 0.4, 12, 200, 10, 0.0),
])
def test_em_through_the_abi_equals_the_oracle(hip, seed, n_reads, n_targets, p_unique, max_deg, heavy, max_iter, threshold):
    off, tgt = random_table(seed, n_reads, n_targets, p_unique, max_deg, heavy=heavy)
    want = orr.em(oracle_table(off, tgt, n_targets), max_iter, threshold)
    g = hip.HipReassign(off, tgt, n_targets)
    diffs, counts, unique, prob, choice = g.run(max_iter, threshold)
    info = g.info()
    g.free()
    assert len(diffs) == want.iterations
    assert diffs.tobytes() == np.asarray(want.diffs, dtype=np.float64).tobytes()          # bit for bit, the stop rule depends on it
    assert np.array_equal(counts.astype(np.int64), want.counts)
    assert prob.tobytes() == want.prob.tobytes()
    deg = np.diff(off)
    assert np.array_equal(unique.astype(np.int64), np.bincount(tgt[off[:-1][deg == 1]], minlength=n_targets))
    assert np.array_equal(choice.astype(np.int64), np.where(deg == 1, off[:-1], want.choice))
    assert info["unique_reads"] == int((deg == 1).sum()) and info["multi_reads"] == int((deg > 1).sum())
    assert info["wave_reads"] == int((deg > 32).sum())


@pytest.mark.gpu
def test_abi_refuses_a_broken_table(hip):
    off = np.array([0, 2, 1], dtype=np.uint64)
    with pytest.raises(hip.GanonHipError):
        hip.HipReassign(off, np.zeros(1, dtype=np.uint32), 1)
    with pytest.raises(hip.GanonHipError):
        hip.HipReassign(np.array([0, 1], dtype=np.uint64), np.array([3], dtype=np.uint32), 2)
    with pytest.raises(hip.GanonHipError):
        hip.HipReassign(np.array([0, 1], dtype=np.uint64), np.array([0], dtype=np.uint32), 1, device=99)
    # a read without entries is not part of the reference's dict (reassign.py builds it from .all lines): refused, not run past the table
    with pytest.raises(hip.GanonHipError, match="no entries"):
        hip.HipReassign(np.array([0, 1, 1, 2], dtype=np.uint64), np.array([0, 1], dtype=np.uint32), 2)


@pytest.mark.gpu
def test_negative_threshold_runs_to_max_iter_as_the_reference_does(hip):
    # argparse takes any float for -s; `diff <= threshold` (reassign.py:141) never holds for a negative one
    off, tgt = random_table(3, 4000, 64, 0.5, 6)
    want = orr.em(oracle_table(off, tgt, 64), 7, -1.0)
    with hip.HipReassign(off, tgt, 64) as g:
        diffs, counts, _, prob, _ = g.run(7, -1.0)
    assert want.iterations == 7 and len(diffs) == 7
    assert diffs.tobytes() == np.asarray(want.diffs, dtype=np.float64).tobytes() and prob.tobytes() == want.prob.tobytes()


@pytest.mark.gpu
def test_reassign_after_classify_end_to_end(hip, tmp_path):
    # what `ganon classify --multiple-matches em` does (classify.py:76-88): the binary's .rep/.all, then reassign; product
    # binaries on both sides, compared with the oracle's restatement run on the same files
    import cli_util as cu
    import test_cli_kat as tk
    os.makedirs(tmp_path / "db")
    db = tk.make_sim_db(str(tmp_path / "db"))
    pre = str(tmp_path / "cls")
    cu.run(cu.BIN_HIP, ["--paired-reads", db["fq1"] + "," + db["fq2"], "--ibf", db["ibf"], "--tax", db["tax"], "--output-prefix", pre, "--output-all",
                        "--skip-lca", "--quiet", "--rel-cutoff", "0.25", "--rel-filter", "0.1"])
    want = orr.reassign_files(pre + ".rep")
    assert want is not None
    p = subprocess.run([BIN_HIP, "-i", pre, "-o", pre + "_em", "--verbose"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert open(pre + "_em.rep").read() == want[0] and open(pre + "_em.one").read() == want[1][""]
    assert "[reassign]" in p.stderr
    # the same vectors the golden maker derived from this classification (sim_em_mode): same input -> same output
    assert open(pre + ".all").read() == open(os.path.join(GOLD, "sim_em_mode", "in.all")).read()
    assert open(pre + "_em.one").read() == open(os.path.join(GOLD, "sim_em_mode", "out.one")).read()


@pytest.mark.parametrize("split_reads", [False, True])
def test_table_read_by_several_threads_equals_the_oracle(checker, tmp_path, split_reads):
    # an .all of ~20 MB is parsed in several chunks (reassign.cpp read_table): a read's lines straddle chunk borders, targets appear first
    # in different chunks, and -- split_reads -- some reads are listed again far from their first lines (then one read, entries in file order)
    rng = np.random.default_rng(19)
    n_reads, n_targets = 250_000, 3000
    names = [f"GCF_{rng.integers(10**8, 10**9)}.{t % 7}" for t in range(n_targets)]
    deg = np.where(rng.random(n_reads) < 0.4, 1, rng.integers(2, 9, size=n_reads))
    w = rng.random(n_targets) ** 3 + 1e-3
    tg = rng.choice(n_targets, size=int(deg.sum()), p=w / w.sum())
    lines, later, at = [], [], 0
    for r in range(n_reads):
        rid = f"A00000:12:HXXXXXXX:1:{1101 + r % 50}:{r}:{(r * 7919) % 100000} 1:N:0:ACGTACGT"
        for j in range(int(deg[r])):
            line = f"{rid}\t{names[tg[at]]}\t{10 + (at & 63)}\n"
            (later if split_reads and j >= 2 and r % 1000 == 3 else lines).append(line)
            at += 1
    text = "".join(lines) + "".join(later)
    assert len(text) > 16 << 20
    (tmp_path / "big.all").write_text(text)
    used = sorted(set(tg.tolist()))
    (tmp_path / "big.rep").write_text("".join(f"H1\t{names[t]}\t{3 + t % 5}\t{t % 3}\t0\tspecies\tname of {t}\n" for t in used)
                                      + f"#total_classified\t{n_reads}\n#total_unclassified\t0\n")
    want = orr.reassign_files(str(tmp_path / "big.rep"), 10, 0)
    p = subprocess.run([checker, "-i", "big", "-o", "out", "--quiet"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert open(tmp_path / "out.rep").read() == want[0]
    assert open(tmp_path / "out.one").read() == want[1][""]
    # a bad line deep in the file is reported with its number in the file
    bad_at = text.count("\n") // 2
    parts = text.split("\n")
    parts[bad_at] = parts[bad_at].replace("\t", " ", 1)
    (tmp_path / "bad.all").write_text("\n".join(parts))
    (tmp_path / "bad.rep").write_text((tmp_path / "big.rep").read_text())
    p = subprocess.run([checker, "-i", "bad", "-o", "x", "--quiet"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and f"line {bad_at + 1} " in p.stderr, p.stderr[-300:]
