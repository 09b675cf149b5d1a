"""f-4: the oracle's restatement of `ganon reassign` against vectors the reference's own reassign.py produced
(tests/golden/reassign/, made by scripts/make_reassign_golden.py in the build container)."""
import json
import os

import pytest

from oracle import reassign as orr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reassign")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)))


def golden(case):
    d = os.path.join(GOLD, case)
    cfg = json.load(open(os.path.join(d, "cfg.json")))
    ones = {fn[len("out"):-len(".one")].lstrip("."): open(os.path.join(d, fn)).read() for fn in os.listdir(d) if fn.startswith("out") and fn.endswith(".one")}
    return d, cfg, open(os.path.join(d, "out.rep")).read(), ones, open(os.path.join(d, "log.txt")).read()


def test_the_vectors_cover_what_they_claim():
    assert len(CASES) >= 15
    logs = {c: golden(c)[4] for c in CASES}
    assert any(" - Iteration 15 " in l for l in logs.values())              # long runs
    assert " - 0 reassigned reads" in logs["syn_all_unique"]
    assert " - Iteration 1 (1.0)" in logs["syn_no_unique"]                   # no unique read: every prob starts at 0
    assert logs["syn_max_iter_1"].count("Iteration") == 1 and logs["syn_threshold"].count("Iteration") >= 2
    two = golden("sim_two_levels")
    assert set(two[3]) == {"1_first", "2_second"} and set(golden("sim_two_levels_single")[3]) == {""}
    # the reference's tie-break depends on the line order of .all: the two orders of the same classification differ
    assert golden("sim_default")[3][""] != golden("sim_reference_order")[3][""]


@pytest.mark.parametrize("case", CASES)
def test_oracle_reassign_equals_the_reference(case):
    d, cfg, rep, ones, log = golden(case)
    assert cfg["returned"] is True
    got = orr.reassign_files(os.path.join(d, "in.rep"), cfg["max_iter"], cfg["threshold"])
    assert got is not None
    new_rep, new_ones, results = got
    assert new_rep == rep
    assert new_ones == ones
    # the log's iteration lines: " - Iteration <i> (<round(diff, 6)>)" per table, in table order (:131-138)
    want = [l for l in log.split("\n") if l.startswith(" - Iteration")]
    have = [f" - Iteration {i + 1} ({round(x, 6)})" for r in results for i, x in enumerate(r.diffs)]
    assert have == want


def test_properties_the_reference_tests_check():
    # tests/ganon/integration/test_reassign.py:121-134: one line per read in .one, and unique + lca of the new .rep is the
    # number of classified reads
    for case in CASES:
        d, cfg, rep, ones, _ = golden(case)
        total = 0
        for h, text in ones.items():
            tb = orr.read_table(os.path.join(d, "in.all" if h == "" else f"in.{h}.all"))
            lines = text.split("\n")[:-1]
            assert len(lines) == len(tb.read_ids) == len({l.split("\t")[0] for l in lines})
            total += len(lines)
        summed = sum(int(l.split("\t")[3]) + int(l.split("\t")[4]) for l in rep.split("\n") if l and l[0] != "#")
        assert summed == total
