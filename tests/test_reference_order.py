"""--reference-order: .all lines and .rep rows in the iteration order of the reference's robin_hood maps
(GanonClassify.cpp:53,180,583,836).  The library is not in the reference tree, so nothing here is a run of the reference;
what IS pinned: the two string hashes against independent implementations (MurmurHash64A as published; libstdc++'s own
std::hash<std::string> through a C++ helper), the slot bookkeeping against an independently written Python table
(tests/robin_twin.py) on random, clustered and hand-worked insert sequences, and the host pipeline's use of it (which keys
are inserted in which order) against a replay of the reference's loops in Python."""
import os
import subprocess

import numpy as np
import pytest

import cli_util as cu
import robin_twin as rt
from test_cli_kat import oracle_bin, sim_db  # noqa: F401  (fixtures)

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tap():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "host_oracle"), "-s", "robin_check"])
    exe = os.path.join(HERE, "host_oracle", "robin_check")

    def ask(text: str):
        p = subprocess.run([exe], input=text, capture_output=True, text=True, check=True, timeout=900)
        return p.stdout.splitlines()
    return ask


def test_string_hashes(tap):
    rng = np.random.default_rng(1)
    words = ["", "a", "1", "T0.1", "Bacteria", "GCF_000005845.2", "e1F_e2R", "0123456", "01234567", "012345678"]
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_.|-", dtype=np.uint8)
    for n in range(0, 41):
        words.append(bytes(alphabet[rng.integers(0, len(alphabet), size=n)]).decode())
    out = tap("".join(f"H {w}\n" for w in words))
    for w, line in zip(words, out):
        rh, std_restated, std_real = (int(x, 16) for x in line.split())
        assert std_restated == std_real == rt.std_hash(w.encode()), w      # restatement == libstdc++ == published MurmurHash64A
        assert rh == rt.rh_hash_bytes(w.encode()), w
    a, b = tap("P reads.1 T17.1\nP x y\n")
    for line, (x, y) in zip((a, b), (("reads.1", "T17.1"), ("x", "y"))):
        mine, real = (int(v, 16) for v in line.split())
        assert mine == real == rt.pair_hash(x.encode(), y.encode())


def _table_order(tap, seq):
    return [int(x) for x in tap(f"T {len(seq)}\n" + "".join(f"{k} {h:x}\n" for k, h in seq))[0].split()] if seq else []


def _find_hash(rng, mult, mask, idx, low):
    """a 64-bit value whose mixed form has home slot `idx` (under `mask`) and low five bits `low`"""
    while True:
        h = int(rng.integers(0, 1 << 63))
        m = (h * mult) & rt.M64
        m ^= m >> 33
        if ((m >> 5) & mask) == idx and (m & 31) == low:
            return h


def test_slot_order_hand_worked(tap):
    # 8 buckets (the initial size), everything with low bits 0 so that equal homes tie on the info byte:
    #   a, b, c at home 3 -> slots 3, 4, 5 in insertion order;  d at home 4 is poorer than b and c: after them, slot 6;
    #   e at home 2 -> its own slot;  f at home 3, inserted last, displaces d: richer cells move up (robin hood)
    rng = np.random.default_rng(5)
    mult = 0xc4ceb9fe1a85ec53
    a, b, c = (_find_hash(rng, mult, 7, 3, 0) for _ in range(3))
    d, e, f = _find_hash(rng, mult, 7, 4, 0), _find_hash(rng, mult, 7, 2, 0), _find_hash(rng, mult, 7, 3, 0)
    seq = [(1, a), (2, b), (3, c), (4, d), (5, e), (6, f)]
    assert _table_order(tap, seq) == [5, 1, 2, 3, 6, 4]
    # within one home the cell with the LARGER low hash bits sits first, whatever the insertion order
    g, h = _find_hash(rng, mult, 7, 1, 9), _find_hash(rng, mult, 7, 1, 20)
    assert _table_order(tap, [(1, g), (2, h)]) == [2, 1] and _table_order(tap, [(2, h), (1, g)]) == [2, 1]
    # the 7th key does not fit 80 % of 8: the table doubles and is refilled in OLD SLOT ORDER
    keys = [(i + 1, _find_hash(rng, mult, 15, [9, 1, 9, 1, 9, 1, 4][i], 0)) for i in range(7)]
    # homes under 16 buckets: 9 1 9 1 9 1 4; after the growth slot order = 1s (in the order the 8-bucket table held them), 4, 9s
    small = rt.RobinTable()
    for k, hv in keys[:6]:
        small.insert(k, hv)
    held = small.order()
    exp = [k for k in held if k in (2, 4, 6)] + [7] + [k for k in held if k in (1, 3, 5)]
    assert _table_order(tap, keys) == exp
    # re-inserting a key changes nothing
    assert _table_order(tap, keys + keys[:3]) == exp


@pytest.mark.parametrize("n,clustered", [(5, False), (6, False), (7, False), (40, False), (700, False), (5000, False), (300, True), (2000, True)])
def test_slot_order_against_the_python_table(tap, n, clustered):
    rng = np.random.default_rng(n + clustered)
    if clustered:   # many keys per home slot: long displacement chains -> the info byte runs out -> fewer hash bits per info
        # (try_increase_info), and where that is not enough a rehash with another multiplier at the same size
        base = [int(x) for x in rng.integers(0, 1 << 62, size=max(2, n // 60))]
        seq = [(i, (base[i % len(base)] + (i // len(base)) * (1 << 58)) & rt.M64) for i in range(n)]
    else:
        seq = [(i, int(x)) for i, x in enumerate(rng.integers(0, 1 << 63, size=n))]
    twin = rt.RobinTable()
    for k, h in seq:
        twin.insert(k, h)
    assert _table_order(tap, seq) == twin.order()
    assert sorted(twin.order()) == list(range(n))


def _predict(sim_db, all_lines_by_read, read_order):
    """replay of the reference's loops: per read the TMatches slot order, then the report rows' first-touch order through the
    thread's TRep and sum_reports' copy"""
    targets = [t for t, _ in sim_db["built"].hashes_count] if hasattr(sim_db["built"], "hashes_count") else list(sim_db["targets"])
    # TMap: targets in bin-map order (first appearance)
    seen, first = set(), []
    for _, t in sim_db["built"].bin_map:
        if t not in seen:
            seen.add(t)
            first.append(t)
    tmap = rt.RobinTable()
    for t in first:
        tmap.insert(t, rt.rh_hash_bytes(t.encode()))
    rank = {t: i for i, t in enumerate(tmap.order())}
    lines, touched = {}, []
    for rid in read_order:
        got = all_lines_by_read[rid]
        tm = rt.RobinTable()
        for t in sorted(got, key=lambda t: rank[t]):
            tm.insert(t, rt.rh_hash_bytes(t.encode()))
        lines[rid] = tm.order()
        touched += lines[rid]
        if len(got) > 1:
            touched.append("1")      # no tax: multi-match reads count on the root node (:794-799)
    return lines, touched, targets


def _rep_order(prefix, touched):
    a = rt.RobinTable()
    for t in touched:
        a.insert(t, rt.pair_hash(prefix.encode(), t.encode()))
    b = rt.RobinTable()
    for t in a.order():
        b.insert(t, rt.pair_hash(prefix.encode(), t.encode()))
    return b.order()


def test_pipeline_follows_the_replayed_maps(oracle_bin, sim_db, tmp_path):
    out = str(tmp_path / "ref")
    base = ["--ibf", sim_db["ibf"], "--paired-reads", sim_db["fq1"] + "," + sim_db["fq2"], "--output-all", "--output-unclassified",
            "--rel-cutoff", "0.1", "--rel-filter", "1", "--quiet"]          # --rel-filter 1: nothing is dropped
    cu.run(oracle_bin, base + ["-o", out, "--reference-order", "--threads", "1"])
    cu.run(oracle_bin, base + ["-o", out + "_default"])
    ref_lines = [l.rstrip("\n").split("\t") for l in open(out + ".all")]
    def_lines = [l.rstrip("\n").split("\t") for l in open(out + "_default.all")]
    assert sorted(map(tuple, ref_lines)) == sorted(map(tuple, def_lines)) and len(ref_lines) > 100
    assert ref_lines != def_lines                                                # (the order does differ)
    by_read, order = {}, []
    for rid, t, _ in ref_lines:
        if rid not in by_read:
            order.append(rid)
        by_read.setdefault(rid, []).append(t)
    assert any(len(v) > 3 for v in by_read.values())
    lines, touched, _ = _predict(sim_db, {r: set(v) for r, v in by_read.items()}, order)
    for rid in order:
        assert by_read[rid] == lines[rid], rid
    rows = [l.split("\t") for l in open(out + ".rep") if not l.startswith("#")]
    assert [r[1] for r in rows] == _rep_order("", touched)
    assert sorted(tuple(r) for r in rows) == sorted(tuple(l.split("\t")) for l in open(out + "_default.rep") if not l.startswith("#"))
    # the order is a property of the input, not of how the pipeline cut it into batches or how many device workers ran
    for tag, env, dev in (("b7", {"GANON_HOST_BATCH_READS": "7"}, "0"), ("w3", {"GANON_HOST_BATCH_READS": "11"}, "0,0,0")):
        p = subprocess.run([oracle_bin] + base + ["-o", out + tag, "--reference-order", "--threads", "1", "--device", dev],
                           capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
        assert p.returncode == 0, p.stderr
        for ext in (".all", ".rep", ".unc"):
            assert open(out + tag + ext, "rb").read() == open(out + ext, "rb").read(), (tag, ext)


def test_reference_order_with_a_taxonomy_and_rules(oracle_bin, sim_db, tmp_path):
    # with LCA, --rel-filter and --fpr-query: the same lines and rows as the default order, reordered only
    out = str(tmp_path / "t")
    base = ["--ibf", sim_db["ibf"], "--tax", sim_db["tax"], "--paired-reads", sim_db["fq1"] + "," + sim_db["fq2"], "--output-all",
            "--output-lca", "--output-unclassified", "--output-stats", "--rel-cutoff", "0.2", "--rel-filter", "0.3", "--fpr-query", "1e-3", "--quiet"]
    cu.run(oracle_bin, base + ["-o", out, "--reference-order", "--threads", "1"])
    cu.run(oracle_bin, base + ["-o", out + "_d"])
    for ext in (".all", ".rep"):
        assert sorted(open(out + ext).read().splitlines()) == sorted(open(out + "_d" + ext).read().splitlines()), ext
    for ext in (".one", ".unc", ".sta"):
        assert open(out + ext, "rb").read() == open(out + "_d" + ext, "rb").read(), ext
    res = cu.Res(out)
    res.sanity_check(has_tax=True)


def test_reference_order_needs_one_thread(oracle_bin, sim_db, tmp_path):
    p = cu.run(oracle_bin, ["--ibf", sim_db["ibf"], "--single-reads", sim_db["fq1"], "-o", str(tmp_path / "x"), "--reference-order",
                            "--threads", "2"], check=False)
    assert p.returncode != 0 and "--reference-order needs --threads 1" in p.stderr
    h = cu.run(oracle_bin, ["-h"], check=False)
    assert "reference-order" in (h.stdout + h.stderr) and "UNVERIFIED" in (h.stdout + h.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("hibf", [False, True])
def test_reference_order_hip_equals_oracle_backend(oracle_bin, sim_db, tmp_path, hibf):
    # the product binary in this mode: the device hands over every match that passed the cutoff (no pre-pass), the host replays
    # the maps -- byte for byte what the oracle-backend twin writes
    base = ["--ibf", sim_db["hibf"] if hibf else sim_db["ibf"], "--tax", sim_db["tax"], "--paired-reads", sim_db["fq1"] + "," + sim_db["fq2"],
            "--output-all", "--output-lca", "--output-unclassified", "--output-stats", "--rel-cutoff", "0.2", "--rel-filter", "0.3",
            "--fpr-query", "1e-3", "--quiet", "--reference-order", "--threads", "1"] + (["--hibf"] if hibf else [])
    a, b = str(tmp_path / "hip"), str(tmp_path / "ora")
    cu.run(cu.BIN_HIP, base + ["-o", a])
    cu.run(oracle_bin, base + ["-o", b])
    for ext in (".all", ".one", ".unc", ".rep", ".sta"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext
