"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the `.rep` reader of `ganon report`, restated.

`ganon report` starts by reading the classify run's .rep with parse_rep (/root/reference/src/ganon/report.py:163-209): per hierarchy
label and target the direct matches, unique reads and lca reads (rows of one label and target are ADDED), per label the sums
"matches" and "reads" (= unique + lca), and a "total" entry from the two `#total_*` lines (unclassified forced to 0 when the report
normalises).  This is the consumer contract of ganon-classify's .rep writer (report.cpp): pinned by what the reference's
own function returned for .rep files of this repository's binary (tests/golden/report/, scripts/make_report_golden.py)."""
from __future__ import annotations


def parse_rep(path: str, normalize: bool = False):
    reports, counts = {}, {}
    total_matches = 0
    classified = unclassified = None
    with open(path) as f:
        for line in f:
            fields = line.rstrip().split("\t")              # :169 (rstrip: a trailing tab-separated empty column would vanish)
            if fields[0] == "#total_classified":             # :170-171
                classified = int(fields[1])
            elif fields[0] == "#total_unclassified":         # :172-173
                unclassified = int(fields[1]) if not normalize else 0
            else:                                            # :174-200
                label, target = fields[0], fields[1]
                direct, unique, lca = int(fields[2]), int(fields[3]), int(fields[4])
                level = reports.setdefault(label, {})
                tally = counts.setdefault(label, {"matches": 0, "reads": 0})
                row = level.setdefault(target, {"direct_matches": 0, "unique_reads": 0, "lca_reads": 0})
                row["direct_matches"] += direct
                row["unique_reads"] += unique
                row["lca_reads"] += lca
                tally["matches"] += direct
                tally["reads"] += unique + lca
                total_matches += direct
    counts["total"] = {"matches": total_matches, "reads": classified, "unclassified": unclassified}   # :202-206
    return reports, counts
