"""Runs a ganon-classify binary on the reference's known-answer scenarios end to end (files in, files out) and
parses the outputs the way the reference's own test harness does (tests/ganon-classify/GanonClassify.test.cpp:40-168)."""
from __future__ import annotations

import os
import subprocess
from typing import Dict, List

import ganon_fixtures as gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN_HIP = os.path.join(ROOT, "ganon_amd", "host", "ganon-classify")
BIN_ORACLE = os.path.join(ROOT, "tests", "host_oracle", "ganon-classify-oracle")


def build_oracle_binary() -> str:
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host_oracle"), "-s"])
    return BIN_ORACLE


def run(binary: str, args: List[str], check: bool = True):
    p = subprocess.run([binary] + args, capture_output=True, text=True, timeout=600)  # (a hang fails the test)
    if check and p.returncode != 0:
        raise AssertionError(f"{binary} {' '.join(args)} -> rc {p.returncode}\n{p.stderr}")
    return p


class Res:
    """GanonClassify.test.cpp:40-145"""

    def __init__(self, prefix: str, all_file=True, lca_file=True, unc_file=True):
        self.total_classified = self.total_unclassified = 0
        self.matches = self.unique_reads = self.lca_reads = 0
        self.rep_rows = []
        for line in open(prefix + ".rep"):
            f = line.rstrip("\n").split("\t")
            if f[0] == "#total_classified":
                self.total_classified = int(f[1])
            elif f[0] == "#total_unclassified":
                self.total_unclassified = int(f[1])
            else:
                self.rep_rows.append(f)
                self.matches += int(f[2])
                self.unique_reads += int(f[3])
                self.lca_reads += int(f[4])
        self.all, self.lines_all = self._parse(prefix + ".all") if all_file else ({}, 0)
        self.lca, self.lines_lca = self._parse(prefix + ".one") if lca_file else ({}, 0)
        self.unc = [l.rstrip("\n") for l in open(prefix + ".unc")] if unc_file else []

    @staticmethod
    def _parse(path):
        out: Dict[str, Dict[str, int]] = {}
        n = 0
        for line in open(path):
            f = line.rstrip("\n").split("\t")
            out.setdefault(f[0], {})[f[1]] = int(f[2])
            n += 1
        return out, n

    def sanity_check(self, output_all=True, output_lca=True, has_tax=False, output_unc=True):
        """GanonClassify.test.cpp:147-168"""
        if output_all:
            assert len(self.all) == self.total_classified
            assert self.lines_all == self.matches
        if output_lca and has_tax:
            assert len(self.lca) == self.total_classified
            assert self.lines_lca == self.total_classified
        if output_unc:
            assert len(self.unc) == self.total_unclassified


class KatFiles:
    """Materialises the KAT builds (.ibf), reads (.fasta) and taxonomies (.tax) in a directory."""

    def __init__(self, kat: dict, workdir: str):
        self.kat = kat
        self.dir = workdir
        os.makedirs(workdir, exist_ok=True)
        self.built = {}
        for name, b in kat["builds"].items():
            built = gf.build_ibf(b["targets"], b["k"], b["w"], max_fp=b["max_fp"])
            gf.write_ibf(self.ibf(name), built)
            self.built[name] = built
        for rid, seq in kat["reads"].items():
            # '-' is the reference literal for "unknown char, becomes A" -- N takes the same route through the parser
            gf.write_fasta(self.read(rid), [(rid, seq.replace("-", "N"))])
        for tname, tax in kat["tax"].items():
            gf.write_tax(self.tax(tname), tax)

    def ibf(self, name):
        return os.path.join(self.dir, name + ".ibf")

    def read(self, rid):
        return os.path.join(self.dir, rid + ".fasta")

    def tax(self, name):
        return os.path.join(self.dir, name + ".tax")

    def case_args(self, case: dict, prefix: str) -> List[str]:
        """flags as GanonClassify.test.cpp builds its Config (defaultConfig :21-33: all outputs on, threads 4, quiet)"""
        a = ["--output-prefix", prefix, "--output-all", "--output-lca", "--output-stats", "--output-unclassified",
             "--threads", "4", "--quiet", "--ibf", ",".join(self.ibf(n) for n in case["ibf"])]
        if case["single"]:
            a += ["--single-reads", ",".join(self.read(r) for r in case["single"])]
        if case["paired"]:
            a += ["--paired-reads", ",".join(self.read(r) for pair in case["paired"] for r in pair)]
        a += ["--rel-cutoff", ",".join(str(x) for x in case["rel_cutoff"])]
        a += ["--rel-filter", ",".join(str(x) for x in case["rel_filter"])]
        a += ["--fpr-query", ",".join(repr(float(x)) for x in case["fpr_query"])]
        if case.get("tax"):
            a += ["--tax", ",".join(self.tax(t) for t in case["tax"])]
        if case.get("hierarchy_labels"):
            a += ["--hierarchy-labels", ",".join(case["hierarchy_labels"])]
        if case.get("output_single"):
            a += ["--output-single"]
        return a


def check_case(binary: str, files: KatFiles, case: dict, outdir: str) -> Res:
    prefix = os.path.join(outdir, case["name"])
    run(binary, files.case_args(case, prefix))
    has_tax = bool(case.get("tax"))
    res = Res(prefix, lca_file=has_tax)
    assert os.path.exists(prefix + ".sta")
    res.sanity_check(has_tax=has_tax)
    for rid, size in case["expected_sizes"].items():
        assert rid in res.all, (case["name"], rid, "not classified")
        assert len(res.all[rid]) == size, (case["name"], rid, res.all[rid])
    for rid, tc in case["expected_counts"].items():
        for t, c in tc.items():
            assert res.all[rid].get(t) == c, (case["name"], rid, t, res.all[rid])
    for rid, want in case.get("expected_lca", {}).items():
        assert res.lca.get(rid) == want, (case["name"], rid, res.lca.get(rid))
    return res
