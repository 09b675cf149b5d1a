"""Pins the CPU oracle against the reference's own known-answer tests
(/root/reference/tests/ganon-classify/GanonClassify.test.cpp, ported as data in
tests/golden/kat_classify.json) and SURVEY.md Appendix B golden intermediates."""
import numpy as np
import pytest

import ganon_fixtures as gf
import oracle


@pytest.fixture(scope="module")
def builds(kat):
    out = {}
    for name, b in kat["builds"].items():
        out[name] = gf.build_ibf(b["targets"], b["k"], b["w"], max_fp=b["max_fp"])
    return out


def test_adjust_seed():
    # src/utils/include/utils/adjust_seed.hpp:33-37
    assert oracle.adjust_seed(32) == 0x8F3F73B5CF1C9ADE
    assert oracle.adjust_seed(4) == 0x8F
    assert oracle.adjust_seed(19) == 0x8F3F73B5CF1C9ADE >> 26


def _ibf_constants_from_their_closed_forms():
    """The six constants of seqan3::interleaved_bloom_filter::hash_and_fit are restated from memory -- SeqAn3 is not in
    /root/reference -- but they are not arbitrary 20-digit numbers: each follows from a closed form over 2^64 (80-digit decimal
    arithmetic; every quotient is far enough from an integer that 80 digits decide the floor)."""
    from decimal import Decimal, getcontext, ROUND_FLOOR
    getcontext().prec = 80
    two64 = Decimal(2) ** 64

    def exp1():   # e = sum 1/k!
        s, t = Decimal(0), Decimal(1)
        for k in range(1, 80):
            s += t
            t /= k
        return s

    def pi():     # Machin: pi = 16 atan(1/5) - 4 atan(1/239)
        def atan_inv(n):
            x = Decimal(1) / n
            s, t, k = Decimal(0), x, 0
            while abs(t) > Decimal(10) ** -78:
                s += t / (2 * k + 1) * (-1 if k & 1 else 1)
                t *= x * x
                k += 1
            return s
        return 16 * atan_inv(5) - 4 * atan_inv(239)

    def fl(x):
        return int(x.to_integral_value(rounding=ROUND_FLOOR))

    seeds = [fl(two64 / (exp1() / 2)),
             fl(two64 / Decimal(2).sqrt()) | 1,                 # made odd: the floor is even (...212), the constant is ...213
             fl(two64 / Decimal(3).sqrt()),
             fl(two64 / (Decimal(5).sqrt() / 2)),
             fl(two64 / 2 / (3 * pi() / 5))]                    # half of the documented form: 2^63
    golden = (1 + Decimal(5).sqrt()) / 2
    return seeds, fl(two64 / golden)


def test_ibf_hash_constants_follow_from_their_closed_forms_everywhere():
    """... and the oracle (oracle/ganon_oracle.c GNO_IBF_SEEDS / GNO_IBF_MULTIPLIER), the product's ONE definition
    (include/ganon_ibf_hash.h, read as text) and what libganon_hip.so was built with (gn_ibf_hash_constants) all hold exactly those.  A
    mis-remembered digit would not survive this; what it cannot show is that SeqAn3 uses these forms (first contact)."""
    import ctypes as C
    import os
    import re
    import ganon_amd
    seeds, mul = _ibf_constants_from_their_closed_forms()
    assert seeds == [13572355802537770549, 13043817825332782213, 10650232656628343401, 16499269484942379435, 4893150838803335377]
    assert mul == 0x9E3779B97F4A7C15 == 11400714819323198485
    assert all(s & 1 for s in seeds) and mul & 1          # odd: multiplication by them is a bijection on 64-bit words
    L = oracle.lib()
    assert list((C.c_uint64 * 5).in_dll(L, "GNO_IBF_SEEDS")) == seeds
    assert C.c_uint64.in_dll(L, "GNO_IBF_MULTIPLIER").value == mul
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ganon_ibf_hash.h")).read()
    assert [int(x) for x in re.findall(r"(\d+)ULL", re.search(r"#define GN_IBF_SEED_LIST \{([^}]*)\}", hdr).group(1))] == seeds
    assert int(re.search(r"#define GN_IBF_MULTIPLIER (\d+)ULL", hdr).group(1)) == mul
    assert ganon_amd.hip.ibf_hash_constants() == (seeds, mul)
    # one definition: no other literal copy of a seed or of the multiplier under ganon_amd/ (round 5 had five)
    for d, _, files in os.walk(os.path.join(root, "ganon_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".hpp", ".cpp", ".py")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert not any(str(c) in text for c in seeds + [mul]), os.path.join(d, f)


def test_golden_ibf_config(kat, builds):
    for name, want in kat["golden_ibf_config"].items():
        if name.startswith("_"):
            continue
        cfg = builds[name].config
        for key, v in want.items():
            assert cfg[key] == v, (name, key, cfg[key], v)
    # hash_shift = countl_zero(bin_size)
    assert builds["build3"].ibf.hash_shift == 56
    assert builds["build1"].ibf.hash_shift == 60


def test_golden_target_fpr(builds):
    # per-target fpr (GanonClassify.cpp:968-982) for scenario 2
    f = builds["build3"].as_filter()
    fpr = dict(zip(f.targets, f.target_fpr))
    assert fpr["e0"] == pytest.approx(0.011027948716788773, rel=1e-12)
    assert fpr["e1F_e2R"] == pytest.approx(0.009430929226122473, rel=1e-12)
    assert fpr["e2F_e2R"] == pytest.approx(0.00799684369643487, rel=1e-12)
    cnt = dict(builds["build3"].hashes_count)
    assert [cnt[t] for t in ("e0", "e1F", "e1F_e1R", "e2F_e1R", "e1F_e2R", "e2F_e2R")] == [25, 25, 25, 25, 24, 23]


def _run_case(kat, builds, case):
    """One hierarchy run the way GanonClassify.cpp:1461-1639 does it; returns {read: {target: count}}."""
    labels = case.get("hierarchy_labels") or ["H1"] * len(case["ibf"])
    if len(labels) == 1:
        labels = labels * len(case["ibf"])
    rel_cutoff = case["rel_cutoff"] * (len(case["ibf"]) if len(case["rel_cutoff"]) == 1 else 1)
    levels = {}
    for i, (ibf_name, lab) in enumerate(zip(case["ibf"], labels)):
        levels.setdefault(lab, []).append(builds[ibf_name].as_filter(rel_cutoff[i]))
    uniq = sorted(levels)
    rel_filter = case["rel_filter"] * (len(uniq) if len(case["rel_filter"]) == 1 else 1)
    fpr_query = case["fpr_query"] * (len(uniq) if len(case["fpr_query"]) == 1 else 1)
    reads = [(r, kat["reads"][r], None) for r in case["single"]]
    reads += [(a, kat["reads"][a], kat["reads"][b]) for a, b in case["paired"]]
    out = {}
    pending = reads
    for li, lab in enumerate(uniq):
        b0 = kat["builds"][case["ibf"][labels.index(lab)]]
        lvl = oracle.Level(levels[lab], b0["k"], b0["w"], rel_filter[li], fpr_query[li])
        nxt = []
        for rid, s1, s2 in pending:
            res = lvl.classify(gf.literal_to_ranks(s1), gf.literal_to_ranks(s2) if s2 else None)
            if res.status == 0 and res.kept:
                out[rid] = (res.kept, res.max_count)
            else:
                nxt.append((rid, s1, s2))
        pending = nxt
    return out


def test_kat_all(kat, builds):
    for case in kat["cases"]:
        got = _run_case(kat, builds, case)
        for rid, size in case["expected_sizes"].items():
            assert rid in got, (case["name"], rid, "unclassified")
            assert len(got[rid][0]) == size, (case["name"], rid, got[rid][0])
        for rid, tc in case["expected_counts"].items():
            for t, c in tc.items():
                assert got[rid][0].get(t) == c, (case["name"], rid, t, got[rid][0])


def test_kat_lca(kat, builds):
    for case in kat["cases"]:
        if "expected_lca" not in case:
            continue
        got = _run_case(kat, builds, case)
        # merge_tax first-wins (GanonClassify.cpp:1324-1341) + missing targets -> root (:1343-1362)
        tax = {"1": "0"}
        for tname in case["tax"]:
            for node, parent in kat["tax"][tname].items():
                tax.setdefault(node, parent)
        for ibf_name in case["ibf"]:
            for t in kat["builds"][ibf_name]["targets"]:
                tax.setdefault(t, "1")
        lca = oracle.Lca([(p, c) for c, p in tax.items()], "1")
        for rid, want in case["expected_lca"].items():
            kept, max_count = got[rid]
            node = list(kept)[0] if len(kept) == 1 else lca.lca(list(kept))
            cnt = list(kept.values())[0] if len(kept) == 1 else max_count
            assert {node: cnt} == want, (case["name"], rid, node, cnt)


def test_no_false_positive_targets(kat, builds):
    # SURVEY App. B "IBF-level check": with the recalled IBF hash constants the tiny KAT filters
    # show no spurious targets beyond those the reference's tests expect.
    for case in kat["cases"]:
        if case["rel_cutoff"] != [0] or case["rel_filter"] != [1] or case["fpr_query"] != [1.0]:
            continue
        got = _run_case(kat, builds, case)
        for rid, size in case["expected_sizes"].items():
            assert len(got[rid][0]) == size


def test_lca_reference_vectors():
    # /root/reference/tests/utils/LCA.test.cpp:17-107 on tests/golden/lca_*.tax
    import os
    here = os.path.join(os.path.dirname(__file__), "golden")

    def load(fn):
        edges = []
        for line in open(os.path.join(here, fn)):
            f = line.rstrip("\n").split("\t")
            if len(f) >= 2:
                edges.append((f[1], f[0]))
        return oracle.Lca(edges, "1")

    tree = load("lca_tree.tax")
    for want, nodes in [("D0", ["E0", "E1"]), ("C3", ["C3", "F4"]), ("A0", ["G0", "C3", "D5"]), ("1", ["G0", "G5"]),
                        ("B1", ["B1", "C2"]), ("B1", ["C2", "B1"]), ("B0", ["C0", "E1", "F2"]),
                        ("B0", ["F2", "E1", "C0"]), ("B0", ["E1", "C0", "F2"])]:
        assert tree.lca(nodes) == want
    ncbi = load("lca_ncbi.tax")
    for want, nodes in [("1224", ["366602", "470"]), ("2", ["366602", "470", "1406"]),
                        ("2290931", ["2223", "51589"]), ("10239", ["2025595", "491893"]),
                        # (the reference's map literal re-assigns key "1" three times; only the last,
                        #  seven-node query survives and is the one actually REQUIRE'd)
                        ("1", ["366602", "470", "1406", "2223", "51589", "2025595", "491893"])]:
        assert ncbi.lca(nodes) == want


def test_hibf_agent_of_the_longreads_build():
    """oracle.Hibf.bulk_count_longreads = hierarchical_interleaved_bloom_filter.hpp:432-460 with value_t = uint32_t (the reference's
    -DLONGREADS build, GanonClassify.cpp:45-49).  Where no count or sum reaches 2^16 it must agree with the default (uint16) agent;
    where a user bin's sum passes 65535 the default agent wraps (:438,442) and this one does not."""
    import numpy as np
    import ganon_fixtures as gf
    import oracle
    rng = np.random.default_rng(4)
    hb = gf.random_hibf(120, 32, 3, seed=5, density=0.2, hash_funs=2, rows=(700, 900))
    for n, thr in ((1, 1), (40, 3), (500, 20), (3000, 1)):
        hh = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64)
        assert np.array_equal(hb.bulk_count(hh, thr).astype(np.uint32), hb.bulk_count_longreads(hh, thr))
    # one hash repeated 70 000 times: every bin that holds it counts 70 000 -> 4464 after a 16-bit wrap
    for _ in range(200):
        one = np.full(70_000, rng.integers(0, 2 ** 63, dtype=np.uint64), dtype=np.uint64)
        wide, narrow = hb.bulk_count_longreads(one, 1), hb.bulk_count(one, 1)
        hit = np.nonzero(wide)[0]
        if len(hit):
            break
    assert len(hit) > 0 and (wide[hit] % 70_000 == 0).all()
    assert ((wide[hit] & 0xFFFF) == narrow[hit]).all() or (narrow[hit] == 0).any()   # (a wrapped sum of 0 is not reported at all)
