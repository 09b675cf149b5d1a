"""scripts/first_contact.sh -- the one-command check against a real ganon install -- exercised with this repo's own binaries
standing in for "theirs" (there is no SeqAn3 build in this image): every step runs, every row of the verdict table is produced,
and a stand-in that writes in the reference-order model makes the byte-wise rows pass too."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(tmp, *extra):
    return subprocess.run(["bash", os.path.join(ROOT, "scripts", "first_contact.sh"), os.path.join(ROOT, "ganon_amd", "host"), "--work", str(tmp), *extra],
                          capture_output=True, text=True, timeout=900)


def test_first_contact_with_our_binaries_standing_in(tmp_path):
    p = _run(tmp_path / "a", "--their-classify-args=--reference-order")
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    rows = [ln for ln in p.stdout.splitlines() if ln.startswith(("  PASS", "  FAIL", "  diff"))]
    assert len(rows) == 26 and all(r.startswith("  PASS") for r in rows), p.stdout
    assert "ALL REQUIRED ROWS PASS; 0 informational row(s) differ" in p.stdout
    log = open(tmp_path / "a" / "first_contact.log").read()
    assert "--verify-filter" in log and "--inspect-filter" in log and log.count("rc 0") >= 12


def test_first_contact_reports_a_different_line_order_as_information(tmp_path):
    # a stand-in that writes ascending target order: same lines (required rows pass), other bytes (informational rows differ)
    p = _run(tmp_path / "b")
    assert p.returncode == 0, p.stdout[-4000:]
    assert "ALL REQUIRED ROWS PASS" in p.stdout
    assert any(ln.startswith("  diff") and "byte-identical" in ln for ln in p.stdout.splitlines())


def test_first_contact_needs_both_binaries(tmp_path):
    p = subprocess.run(["bash", os.path.join(ROOT, "scripts", "first_contact.sh"), str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "missing or not executable" in p.stderr
