"""SURVEY 8 f-4, second half: ganon-classify's .rep must be what `ganon report` reads.  The reference's reader, parse_rep
(report.py:163-209), was run in the build container on .rep files of this repository's binary (scripts/make_report_golden.py: the function
is executed from the reference's file, nothing of it is copied); its outputs are committed as data.  Here: the restatement in
oracle/report_rep.py returns exactly those, and fresh .rep files of the binaries keep the properties build_report relies on."""
import glob
import json
import os

import pytest

import cli_util as cu
from oracle import report_rep

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "report", "*.rep")))


@pytest.mark.parametrize("rep", CASES, ids=lambda p: os.path.basename(p)[:-4])
def test_restated_parse_rep_equals_the_references_output(rep):
    want = json.load(open(rep[:-4] + ".parse_rep.json"))
    for normalize in (False, True):
        reports, counts = report_rep.parse_rep(rep, normalize)
        got = json.loads(json.dumps({"reports": reports, "counts": counts}))
        assert got == want["normalize_" + str(normalize).lower()]
    assert len(CASES) >= 6


def _check_fresh(binary, tmp, hibf=False):
    import test_cli_kat as tk
    db = tk.make_sim_db(str(tmp))
    prefix = tk._run_sim(binary, db, str(tmp), hibf=hibf)
    reports, counts = report_rep.parse_rep(prefix + ".rep")
    res = cu.Res(prefix)
    total = counts["total"]
    assert total["reads"] == res.total_classified and total["unclassified"] == res.total_unclassified
    assert total["reads"] + total["unclassified"] == 98                      # every pair of the fixture is accounted for
    levels = [k for k in counts if k != "total"]
    assert sum(counts[k]["reads"] for k in levels) == total["reads"]          # unique + lca reads of the levels = classified reads
    assert total["matches"] == sum(1 for _ in open(prefix + ".all"))          # one .all line per direct match
    for lv in levels:
        assert all(r["unique_reads"] + r["lca_reads"] + r["direct_matches"] > 0 for r in reports[lv].values())   # :841: no empty rows
    return open(prefix + ".rep", "rb").read()


def test_fresh_rep_of_the_checker_binary_satisfies_the_report_contract(tmp_path):
    _check_fresh(cu.build_oracle_binary(), tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("hibf", [False, True])
def test_fresh_rep_of_the_product_binary_satisfies_the_report_contract(tmp_path, hibf):
    _check_fresh(cu.BIN_HIP, tmp_path, hibf=hibf)
