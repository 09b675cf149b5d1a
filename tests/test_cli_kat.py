"""End-to-end known-answer tests of the ganon-classify host pipeline (CLI -> file formats -> readers ->
post-processing -> writers) on the reference's scenarios.  On CPU the test-only oracle backend supplies the hot
path; the gpu-marked twins run the product binary (libganon_hip.so through the C ABI)."""
import os
import shutil
import subprocess

import pytest

import cli_util as cu
import ganon_fixtures as gf


@pytest.fixture(scope="module")
def files(kat, tmp_path_factory):
    return cu.KatFiles(kat, str(tmp_path_factory.mktemp("kat_files")))


@pytest.fixture(scope="module")
def oracle_bin():
    return cu.build_oracle_binary()


def _all_cases(binary, kat, files, outdir):
    for case in kat["cases"]:
        cu.check_case(binary, files, case, outdir)


def test_kat_cases_oracle_backend(oracle_bin, kat, files, tmp_path):
    _all_cases(oracle_bin, kat, files, str(tmp_path))


def _expected_sta_of_case(kat, files, case):
    """The .sta file of a KAT scenario as write_stats prints it (GanonClassify.cpp:1167-1218; rows by write_stats_db :1130-1165): one
    row per hierarchy label in the order of the std::map (sorted), every row with the RUN's seq_processed / seq_unclassified /
    kmers_processed and the label's own counts (add_totals / add_reports :197-246), a `-total-` row when there are several labels.
    The per-read results come from the oracle's Level (the hierarchy loop of :1461-1639 as in tests/test_oracle_kat.py)."""
    import oracle
    labels = case.get("hierarchy_labels") or ["H1"] * len(case["ibf"])
    if len(labels) == 1:
        labels = labels * len(case["ibf"])
    rel_cutoff = case["rel_cutoff"] * (len(case["ibf"]) if len(case["rel_cutoff"]) == 1 else 1)
    levels = {}
    for i, (ibf_name, lab) in enumerate(zip(case["ibf"], labels)):
        levels.setdefault(lab, []).append(files.built[ibf_name].as_filter(rel_cutoff[i]))
    uniq = sorted(levels)
    rel_filter = case["rel_filter"] * (len(uniq) if len(case["rel_filter"]) == 1 else 1)
    fpr_query = case["fpr_query"] * (len(uniq) if len(case["fpr_query"]) == 1 else 1)
    reads = [(kat["reads"][r], None) for r in case["single"]] + [(kat["reads"][a], kat["reads"][b]) for a, b in case["paired"]]
    zero = lambda: dict(seqs_classified=0, seqs_unique=0, matches=0, dis_filter=0, dis_fpr=0, kmers_matches=0, kmers_from_classified_seqs=0)  # noqa: E731
    run = dict(zero(), seqs_processed=0, kmers_processed=0)
    per = {}
    pending = reads
    for li, lab in enumerate(uniq):
        b0 = kat["builds"][case["ibf"][labels.index(lab)]]
        lvl = oracle.Level(levels[lab], b0["k"], b0["w"], rel_filter[li], fpr_query[li])
        t = per[lab] = zero()
        nxt = []
        for s1, s2 in pending:
            rr = lvl.classify(gf.literal_to_ranks(s1), gf.literal_to_ranks(s2) if s2 else None)
            if li == 0 and rr.status == 0:          # :706-714 (hierarchy_first)
                run["seqs_processed"] += 1
                run["kmers_processed"] += rr.n_hashes
            if rr.status != 0:                       # :737-747: skipped reads are not passed on
                continue
            t["dis_filter"] += len(rr.discarded_filter)
            t["dis_fpr"] += len(rr.discarded_fpr)
            if rr.kept:
                t["seqs_classified"] += 1
                t["seqs_unique"] += 1 if len(rr.kept) == 1 else 0
                t["matches"] += len(rr.kept)
                t["kmers_from_classified_seqs"] += rr.n_hashes
                t["kmers_matches"] += rr.max_count
            else:
                nxt.append((s1, s2))
        pending = nxt
        for k2 in t:
            run[k2] += t[k2]
    f6 = lambda x: "%.6f" % x                        # noqa: E731  (std::fixed << std::setprecision(6))
    seq_processed = float(run["seqs_processed"]) if run["seqs_processed"] > 0 else 1.0
    unclassified = run["seqs_processed"] - run["seqs_classified"]

    def row(label, t):
        multiple = t["seqs_classified"] - t["seqs_unique"]
        avg = t["matches"] / float(t["seqs_classified"]) if t["seqs_classified"] else 0.0
        kperc = t["kmers_matches"] / float(t["kmers_from_classified_seqs"]) * 100 if t["kmers_matches"] else 0.0
        return "\t".join(["", label, str(int(seq_processed)), str(unclassified), str(t["seqs_classified"]), f6(t["seqs_classified"] / seq_processed * 100),
                          str(t["seqs_unique"]), f6(t["seqs_unique"] / seq_processed * 100), str(multiple), f6(multiple / seq_processed * 100),
                          str(t["matches"]), f6(avg), str(t["dis_filter"]), str(t["dis_fpr"]), str(run["kmers_processed"]), str(t["kmers_matches"]),
                          str(t["kmers_from_classified_seqs"]), f6(kperc)]) + "\n"

    head = "\t".join(["prefix", "hierarchy_label", "seq_processed", "seq_unclassified", "seq_classified", "seq_classified_perc", "seq_unique_matches",
                      "seq_unique_matches_perc", "seq_multiple_matches", "seq_multiple_matches_perc", "matches", "avg_matches_ref_seq",
                      "dis_matches_rel_filter", "dis_matches_fpr_query", "kmers_proccessed", "kmers_matched", "kmers_from_classified_seqs",
                      "kmers_matched_perc"]) + "\n"
    text = head + "".join(row(lab, per[lab]) for lab in uniq)
    if len(uniq) > 1:
        text += row("-total-", run)
    return text


def test_sta_files_of_every_kat_scenario_are_what_write_stats_prints(oracle_bin, kat, files, tmp_path):
    """31 scenarios of GanonClassify.test.cpp, several with two or three hierarchy labels (one row per label + `-total-`): the bytes of
    the .sta file against a restatement of the reference's writer -- columns, order of the rows, fixed six-digit doubles, which counts
    are the run's and which the label's.  (the writers were pinned by "both backends agree" only.)"""
    multi = 0
    for case in kat["cases"]:
        prefix = os.path.join(str(tmp_path), case["name"])
        cu.run(oracle_bin, files.case_args(case, prefix))
        want = _expected_sta_of_case(kat, files, case)
        assert open(prefix + ".sta").read() == want, case["name"]
        multi += want.count("-total-")
    assert multi >= 2   # (the scenarios with several hierarchy labels)


def _expected_rep_lines_of_case(kat, files, case):
    """The .rep file of a KAT scenario as write_report / write_report_totals print it (GanonClassify.cpp:834-863), as a sorted list of
    lines (the reference iterates a robin_hood map: its row order is not defined by the source).  Per hierarchy label and target:
    matches (filter_matches :579-613 counts every kept match), unique reads (:773-778), lca reads (lca_matches :615-627 -- or the root
    node without a taxonomy, :794-799); rank and name columns only with a taxonomy; targets the taxonomy lacks hang under the root as
    "no rank" (:1343-1362).  Totals: classified, and input minus classified (:855-863)."""
    import oracle
    labels = case.get("hierarchy_labels") or ["H1"] * len(case["ibf"])
    if len(labels) == 1:
        labels = labels * len(case["ibf"])
    rel_cutoff = case["rel_cutoff"] * (len(case["ibf"]) if len(case["rel_cutoff"]) == 1 else 1)
    levels, level_ibfs = {}, {}
    for i, (ibf_name, lab) in enumerate(zip(case["ibf"], labels)):
        levels.setdefault(lab, []).append(files.built[ibf_name].as_filter(rel_cutoff[i]))
        level_ibfs.setdefault(lab, []).append(i)
    uniq = sorted(levels)
    rel_filter = case["rel_filter"] * (len(uniq) if len(case["rel_filter"]) == 1 else 1)
    fpr_query = case["fpr_query"] * (len(uniq) if len(case["fpr_query"]) == 1 else 1)
    reads = [(kat["reads"][r], None) for r in case["single"]] + [(kat["reads"][a], kat["reads"][b]) for a, b in case["paired"]]
    has_tax = bool(case.get("tax"))
    lines, classified = [], 0
    pending = reads
    for li, lab in enumerate(uniq):
        b0 = kat["builds"][case["ibf"][labels.index(lab)]]
        lvl = oracle.Level(levels[lab], b0["k"], b0["w"], rel_filter[li], fpr_query[li])
        tax = {}
        if has_tax:   # merge_tax of the level's filters, the first one wins (:1324-1341); root "1" as tests/ganon_fixtures.write_tax writes it
            tax["1"] = ("0", "root", "root")
            tnames = case["tax"] if len(case["tax"]) == len(case["ibf"]) else case["tax"] * len(case["ibf"])
            for i in level_ibfs[lab]:
                for node, parent in kat["tax"][tnames[i]].items():
                    tax.setdefault(node, (parent, f"rank-{node}", f"name-{node}"))
            for i in level_ibfs[lab]:
                for t in kat["builds"][case["ibf"][i]]["targets"]:
                    tax.setdefault(t, ("1", "no rank", t))
        lca = oracle.Lca([(v[0], k2) for k2, v in tax.items()], "1") if has_tax else None
        rep = {}
        nxt = []
        for s1, s2 in pending:
            rr = lvl.classify(gf.literal_to_ranks(s1), gf.literal_to_ranks(s2) if s2 else None)
            if rr.status != 0:
                continue
            if not rr.kept:
                nxt.append((s1, s2))
                continue
            classified += 1
            for t in rr.kept:
                rep.setdefault(t, [0, 0, 0])[0] += 1
            if len(rr.kept) == 1:
                rep[list(rr.kept)[0]][1] += 1
            else:
                rep.setdefault(lca.lca(list(rr.kept)) if has_tax else "1", [0, 0, 0])[2] += 1
        pending = nxt
        for t, (m, u, l) in rep.items():
            if m or u or l:
                lines.append("\t".join([lab, t, str(m), str(u), str(l)] + ([tax[t][1], tax[t][2]] if has_tax else [])))
    lines.append(f"#total_classified\t{classified}")
    lines.append(f"#total_unclassified\t{len(reads) - classified}")
    return sorted(lines)


def test_rep_files_of_every_kat_scenario_are_what_write_report_prints(oracle_bin, kat, files, tmp_path):
    n_rows = 0
    for case in kat["cases"]:
        prefix = os.path.join(str(tmp_path), case["name"])
        cu.run(oracle_bin, files.case_args(case, prefix))
        got = sorted(open(prefix + ".rep").read().splitlines())
        want = _expected_rep_lines_of_case(kat, files, case)
        assert got == want, (case["name"], got[:4], want[:4])
        n_rows += len(want) - 2
    assert n_rows > 60


@pytest.mark.gpu
def test_kat_cases_hip(kat, files, tmp_path):
    assert os.path.exists(cu.BIN_HIP), "ganon-classify was not built"
    _all_cases(cu.BIN_HIP, kat, files, str(tmp_path))


def _flags_and_files(binary, kat, files, tmp):
    base = ["--ibf", files.ibf("build1"), "--single-reads", files.read("readA"), "--rel-cutoff", "0", "--rel-filter", "1",
            "--quiet"]
    # output files only when asked for (GanonClassify.test.cpp:271-315)
    p = os.path.join(tmp, "only_rep")
    cu.run(binary, base + ["-o", p])
    assert os.path.exists(p + ".rep")
    for ext in (".all", ".one", ".unc", ".sta"):
        assert not os.path.exists(p + ext), ext
    p = os.path.join(tmp, "wo_lca")
    cu.run(binary, base + ["-o", p, "--output-all", "--output-unclassified", "--output-stats"])
    assert os.path.exists(p + ".all") and os.path.exists(p + ".unc") and os.path.exists(p + ".sta")
    assert not os.path.exists(p + ".one")
    # short flags, = syntax, output dir creation (:1390-1397)
    p = os.path.join(tmp, "sub", "dir", "short")
    cu.run(binary, ["-i", files.ibf("build1"), "-r", files.read("readA"), "-c", "0", "-d", "1", "-a", "-u", "--output-prefix=" + p,
                    "--quiet"])
    res = cu.Res(p, lca_file=False)
    assert res.all["readA"] == {"A": 5, "T": 5}
    # verbose writes to stderr only (:224-251)
    p = os.path.join(tmp, "verbose")
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "--single-reads", files.read("readA"), "-o", p, "--verbose"])
    assert r.stdout == "" and "ganon-classify processed 1 sequences" in r.stderr and "--output-prefix" in r.stderr
    # exit codes (main.cpp:7-17, Config.hpp:70-172)
    assert cu.run(binary, [], check=False).returncode == 1
    assert cu.run(binary, ["-h"], check=False).returncode == 0
    assert cu.run(binary, ["--version"], check=False).returncode == 0
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "--single-reads", files.read("readA")], check=False)
    assert r.returncode == 1 and "--output-prefix is mandatory" in r.stderr
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "-o", p], check=False)
    assert r.returncode == 1 and "mandatory" in r.stderr
    r = cu.run(binary, ["--ibf", "/nonexistent.ibf", "--single-reads", files.read("readA"), "-o", p], check=False)
    assert r.returncode == 1 and "file not found" in r.stderr
    r = cu.run(binary, base + ["-o", p, "--rel-cutoff", "1.5"], check=False)
    assert r.returncode == 1 and "--rel-cutoff values should be set between 0 and 1" in r.stderr
    r = cu.run(binary, ["--ibf", files.ibf("build1"), "--paired-reads", files.read("readA"), "-o", p], check=False)
    assert r.returncode == 1 and "even number" in r.stderr


def test_flags_and_files_oracle_backend(oracle_bin, kat, files, tmp_path):
    _flags_and_files(oracle_bin, kat, files, str(tmp_path))


@pytest.mark.gpu
def test_flags_and_files_hip(kat, files, tmp_path):
    _flags_and_files(cu.BIN_HIP, kat, files, str(tmp_path))


def _batch_reads(binary, kat, files, tmp):
    # --batch-reads == separate runs (GanonClassify.test.cpp:364-424), compared like aux::filesAreEqualSorted
    common = ["--ibf", files.ibf("build1"), "--tax", files.tax("full"), "--rel-cutoff", "0", "--rel-filter", "1", "--output-all",
              "--output-lca", "--output-unclassified", "--quiet"]
    pp = os.path.join(tmp, "paired")
    cu.run(binary, common + ["-o", pp, "--paired-reads", files.read("readA") + "," + files.read("readT")])
    ps = os.path.join(tmp, "single")
    cu.run(binary, common + ["-o", ps, "--single-reads", files.read("readC")])
    tsv = os.path.join(tmp, "batch.tsv")
    with open(tsv, "w") as f:
        f.write(f"batch_paired\t{files.read('readA')}\t{files.read('readT')}\n")
        f.write(f"batch_single\t{files.read('readC')}\n")
    pb = os.path.join(tmp, "batch")
    cu.run(binary, common + ["-o", pb, "--batch-reads", tsv])
    for ext in (".all", ".one", ".unc", ".rep"):
        assert sorted(open(pp + ext).read()) == sorted(open(pb + "batch_paired" + ext).read()), ext
        assert sorted(open(ps + ext).read()) == sorted(open(pb + "batch_single" + ext).read()), ext
    # hierarchy + batch prefixes -> one file per (prefix, label) (:459-508)
    tsv2 = os.path.join(tmp, "batch2.tsv")
    with open(tsv2, "w") as f:
        f.write(f"batchA\t{files.read('readC')}\nbatchB\t{files.read('readG')}\nbatchC\t{files.read('readA')}\t{files.read('readT')}\n")
    ph = os.path.join(tmp, "hier")
    cu.run(binary, ["--ibf", files.ibf("build1") + "," + files.ibf("build1"), "--tax", files.tax("full") + "," + files.tax("full"),
                    "--hierarchy-labels", "DB1,DB2", "--rel-cutoff", "0", "--rel-filter", "1", "--output-all", "--output-lca",
                    "--output-unclassified", "--quiet", "-o", ph, "--batch-reads", tsv2])
    for b in ("batchA", "batchB", "batchC"):
        for lab in ("DB1", "DB2"):
            assert os.path.exists(f"{ph}{b}.{lab}.all") and os.path.exists(f"{ph}{b}.{lab}.one")
        assert os.path.exists(f"{ph}{b}.unc") and os.path.exists(f"{ph}{b}.rep")
    # two levels without --output-single: both .all files non-empty (:778-793)
    p2 = os.path.join(tmp, "two_levels")
    cu.run(binary, ["--ibf", files.ibf("build1") + "," + files.ibf("build2"), "--hierarchy-labels", "one,two", "--rel-cutoff", "0",
                    "--rel-filter", "1", "--output-all", "--quiet", "-o", p2, "--single-reads",
                    files.read("readA") + "," + files.read("readCG")])
    assert os.path.getsize(p2 + ".one.all") > 0 and os.path.getsize(p2 + ".two.all") > 0


def test_batch_reads_oracle_backend(oracle_bin, kat, files, tmp_path):
    _batch_reads(oracle_bin, kat, files, str(tmp_path))


@pytest.mark.gpu
def test_batch_reads_hip(kat, files, tmp_path):
    _batch_reads(cu.BIN_HIP, kat, files, str(tmp_path))


# ---------------------------------------------------------------------------------------------------------------
# realistic inputs: the reference's 98-pair 150 bp FASTQ fixture (tests/ganon/data/classify/sim.{1,2}.fq.gz) against a
# synthetic multi-target database with split bins, flat (.ibf) and hierarchical (.hibf)
# ---------------------------------------------------------------------------------------------------------------
def _sim_reads():
    import gzip
    here = os.path.join(os.path.dirname(__file__), "golden")
    out = []
    for fn in ("sim.1.fq.gz", "sim.2.fq.gz"):
        recs = []
        with gzip.open(os.path.join(here, fn), "rt") as f:
            lines = f.read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            recs.append((lines[i][1:], lines[i + 1]))
        out.append(recs)
    return out


@pytest.fixture(scope="module")
def sim_db(tmp_path_factory):
    return make_sim_db(str(tmp_path_factory.mktemp("sim_db")))


def make_sim_db(d):
    """Targets = 40 synthetic genomes, each containing a few of the fixture reads verbatim -> true matches."""
    import numpy as np
    r1, r2 = _sim_reads()
    rng = np.random.default_rng(99)
    targets = {}
    for t in range(40):
        parts = []
        for j in range(6):
            parts.append("".join("ACGT"[x] for x in rng.integers(0, 4, size=300)))
            idx = (t * 6 + j) % len(r1)
            parts.append(r1[idx][1] if j % 2 == 0 else r2[idx][1])
        targets[f"T{t}.1"] = "".join(parts)
    built = gf.build_ibf(targets, 19, 31, max_fp=0.05, filter_size=0.0)
    # force split bins: rebuild with a small max_hashes_bin through the size optimiser's mode "fastest" is not needed;
    # instead halve the capacity so that every target spans several technical bins
    ibf_path = os.path.join(d, "sim.ibf")
    gf.write_ibf(ibf_path, built)
    tax = {t: "G" + str(i % 5) for i, t in enumerate(targets)}
    tax.update({"G" + str(i): "ROOTG" for i in range(5)})
    tax["ROOTG"] = "1"
    tax_path = os.path.join(d, "sim.tax")
    gf.write_tax(tax_path, tax)
    # HIBF over the same targets
    import oracle
    uh = {i: np.unique(oracle.minimiser_hash(oracle.to_ranks(s), 19, 31)) for i, s in enumerate(targets.values())}
    hb = gf.random_hibf(len(targets), 64, 2, seed=3, density=0.02, hash_funs=3, rows=(20000, 30000), user_hashes=uh)
    hibf_path = os.path.join(d, "sim.hibf")
    names = list(targets)
    gf.write_hibf(hibf_path, hb, [[f"/some/dir/{n.replace('.', '|||')}.minimiser"] for n in names], 19, 31, 0.05)
    here = os.path.join(os.path.dirname(__file__), "golden")
    return dict(dir=d, ibf=ibf_path, hibf=hibf_path, tax=tax_path, targets=targets, built=built, hibf_obj=hb, names=names,
                fq1=os.path.join(here, "sim.1.fq.gz"), fq2=os.path.join(here, "sim.2.fq.gz"))


def _run_sim(binary, sim_db, tmp, hibf=False, extra=(), thresholds=("--rel-cutoff", "0.25", "--rel-filter", "0.1")):
    p = os.path.join(tmp, "hibf" if hibf else "ibf")
    args = ["--ibf", sim_db["hibf"] if hibf else sim_db["ibf"], "--tax", sim_db["tax"], "--paired-reads",
            sim_db["fq1"] + "," + sim_db["fq2"], "-o", p, "--output-all", "--output-lca", "--output-unclassified",
            "--output-stats", "--quiet"] + list(thresholds) + (["--hibf"] if hibf else []) + list(extra)
    cu.run(binary, args)
    return p


def _check_sim_against_oracle_level(sim_db, prefix, hibf):
    """.all of the run == oracle.Level (GanonClassify.cpp:676-768) on the same reads"""
    import oracle
    if hibf:
        flt = oracle.Filter(hibf=sim_db["hibf_obj"], targets=sim_db["names"], target_bins=[[i] for i in range(len(sim_db["names"]))],
                            target_fpr=[0.05] * len(sim_db["names"]), rel_cutoff=0.25)
    else:
        flt = sim_db["built"].as_filter(0.25)
    lvl = oracle.Level([flt], 19, 31, rel_filter=0.1, fpr_query=1.0)
    r1, r2 = _sim_reads()
    res = cu.Res(prefix)
    res.sanity_check(has_tax=True)
    n_class = n_matches = dis_filter = dis_fpr = 0
    stats = dict(seqs_processed=0, seqs_classified=0, seqs_unique=0, matches=0, dis_filter=0, dis_fpr=0, kmers_processed=0, kmers_matches=0,
                 kmers_from_classified_seqs=0)
    for (rid, s1), (_, s2) in zip(r1, r2):
        rr = lvl.classify(oracle.to_ranks(s1), oracle.to_ranks(s2))
        dis_filter += len(rr.discarded_filter)
        dis_fpr += len(rr.discarded_fpr)
        if rr.status == 0:                                   # :706-714 (reads too short / with too many minimisers are not "processed")
            stats["seqs_processed"] += 1
            stats["kmers_processed"] += rr.n_hashes
        stats["dis_filter"] += len(rr.discarded_filter)
        stats["dis_fpr"] += len(rr.discarded_fpr)
        if rr.kept:                                          # :764-768
            stats["seqs_classified"] += 1
            stats["seqs_unique"] += 1 if len(rr.kept) == 1 else 0
            stats["matches"] += len(rr.kept)
            stats["kmers_from_classified_seqs"] += rr.n_hashes
            stats["kmers_matches"] += rr.max_count
        if rr.kept:
            n_class += 1
            n_matches += len(rr.kept)
            assert res.all.get(rid) == rr.kept, (rid, res.all.get(rid), rr.kept)
        else:
            assert rid not in res.all and rid in res.unc
    assert res.total_classified == n_class and n_class > 20
    assert res.total_classified + res.total_unclassified == len(r1)
    # the .sta row (write_stats :1167-1218): classified reads, matches and the two discarded-match totals are the oracle's
    head, row = [line.rstrip("\n").split("\t") for line in open(prefix + ".sta")][:2]
    sta = dict(zip(head, row))
    assert (int(sta["seq_classified"]), int(sta["matches"])) == (n_class, n_matches)
    assert (int(sta["dis_matches_rel_filter"]), int(sta["dis_matches_fpr_query"])) == (dis_filter, dis_fpr)
    assert hibf or dis_filter > 0
    # ... and the whole file is what write_stats / write_stats_db print (GanonClassify.cpp:1130-1218: 18 tab-separated columns, doubles
    # `std::fixed << std::setprecision(6)`, one row per hierarchy label, no -total- row for a single label), restated here from the
    # reference's text and fed the oracle's per-read results -- the writers are pinned by more than "both backends agree" 
    assert open(prefix + ".sta").read() == _expected_sta("", "H1", stats)


def _expected_sta(prefix, label, t):
    """write_stats (:1167-1218) + write_stats_db (:1130-1165) for ONE hierarchy label; t = the Total of :197-246 as a dict"""
    head = ["prefix", "hierarchy_label", "seq_processed", "seq_unclassified", "seq_classified", "seq_classified_perc", "seq_unique_matches",
            "seq_unique_matches_perc", "seq_multiple_matches", "seq_multiple_matches_perc", "matches", "avg_matches_ref_seq", "dis_matches_rel_filter",
            "dis_matches_fpr_query", "kmers_proccessed", "kmers_matched", "kmers_from_classified_seqs", "kmers_matched_perc"]
    seq_processed = float(t["seqs_processed"]) if t["seqs_processed"] > 0 else 1.0            # :1195-1196
    multiple = t["seqs_classified"] - t["seqs_unique"]                                          # :1139
    avg = t["matches"] / float(t["seqs_classified"]) if t["seqs_classified"] else 0.0            # :1140-1141
    kperc = t["kmers_matches"] / float(t["kmers_from_classified_seqs"]) * 100 if t["kmers_matches"] else 0.0   # :1142-1144
    f6 = lambda x: "%.6f" % x                                                                    # noqa: E731  (:1146)
    row = [prefix, label, str(int(seq_processed)), str(t["seqs_processed"] - t["seqs_classified"]), str(t["seqs_classified"]),
           f6(t["seqs_classified"] / seq_processed * 100), str(t["seqs_unique"]), f6(t["seqs_unique"] / seq_processed * 100), str(multiple),
           f6(multiple / seq_processed * 100), str(t["matches"]), f6(avg), str(t["dis_filter"]), str(t["dis_fpr"]), str(t["kmers_processed"]),
           str(t["kmers_matches"]), str(t["kmers_from_classified_seqs"]), f6(kperc)]
    return "\t".join(head) + "\n" + "\t".join(row) + "\n"


def test_sim_fastq_gz_ibf_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_sim_against_oracle_level(sim_db, _run_sim(oracle_bin, sim_db, str(tmp_path)), False)


def test_sim_fastq_gz_hibf_oracle_backend(oracle_bin, sim_db, tmp_path):
    _check_sim_against_oracle_level(sim_db, _run_sim(oracle_bin, sim_db, str(tmp_path), hibf=True), True)


@pytest.mark.gpu
@pytest.mark.parametrize("hibf", [False, True])
def test_sim_hip_equals_oracle_backend_bytes(oracle_bin, sim_db, tmp_path, hibf):
    # same host code, different hot path: every output file must be byte-identical
    a = _run_sim(cu.BIN_HIP, sim_db, str(tmp_path / "hip"), hibf=hibf)
    b = _run_sim(oracle_bin, sim_db, str(tmp_path / "ora"), hibf=hibf)
    _check_sim_against_oracle_level(sim_db, a, hibf)
    for ext in (".all", ".one", ".unc", ".rep", ".sta"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext


@pytest.mark.gpu
@pytest.mark.parametrize("hibf", [False, True])
@pytest.mark.parametrize("thresholds", [[], ["--rel-cutoff", "0.05", "--rel-filter", "0.3", "--fpr-query", "0.2"],
                                        ["--rel-cutoff", "0", "--rel-filter", "1", "--fpr-query", "1e-3"],
                                        ["--rel-cutoff", "0.1", "--rel-filter", "0", "--fpr-query", "1"]])
def test_device_prefilter_changes_no_output_byte(sim_db, tmp_path, monkeypatch, hibf, thresholds):
    # the device-side pre-pass of filter_matches (one-filter levels) vs the host doing all of it: identical files,
    # including the discarded-match totals of the .sta; [] = the thresholds `ganon classify` passes by default
    # (src/ganon/config.py: --rel-cutoff 0.2 --rel-filter 0.1 --fpr-query 1e-5; the C++ binary's own are 0.2 / 0 / 1)
    extra = thresholds if thresholds else ["--rel-cutoff", "0.2", "--rel-filter", "0.1", "--fpr-query", "1e-5"]
    os.makedirs(tmp_path / "pre")
    os.makedirs(tmp_path / "host")
    a = _run_sim(cu.BIN_HIP, sim_db, str(tmp_path / "pre"), hibf=hibf, thresholds=extra)
    monkeypatch.setenv("GANON_HOST_NO_PREFILTER", "1")
    b = _run_sim(cu.BIN_HIP, sim_db, str(tmp_path / "host"), hibf=hibf, thresholds=extra)
    for ext in (".all", ".one", ".unc", ".rep", ".sta"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext
    assert os.path.getsize(a + ".all") > 0


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[0]: 64-bin IBF (k=19, w=31, h=3), 10 k synthetic 150 bp reads -- the plumbing case
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config1(tmp_path_factory):
    import struct

    import numpy as np
    import oracle
    d = str(tmp_path_factory.mktemp("config1"))
    rng = np.random.default_rng(2024)
    genomes = ["".join("ACGT"[x] for x in rng.integers(0, 4, size=2000)) for _ in range(64)]
    ibf = oracle.Ibf(64, 40009, 3)
    hc = []
    for b, g in enumerate(genomes):
        hv = np.unique(oracle.minimiser_hash(oracle.to_ranks(g), 19, 31))
        ibf.emplace_many(hv, b)
        hc.append((f"G{b}", len(hv)))
    built = gf.BuiltIbf()
    built.ibf = ibf
    built.config = dict(n_bins=64, max_hashes_bin=max(c for _, c in hc), hash_functions=3, kmer_size=19, window_size=31,
                        bin_size_bits=40009, max_fp=0.05, true_max_fp=0.05, true_avg_fp=0.05)
    built.hashes_count = hc
    built.bin_map = [(b, f"G{b}") for b in range(64)]
    gf.write_ibf(os.path.join(d, "c1.ibf"), built)
    recs = []
    for i in range(10000):
        if i % 2:
            g = genomes[i % 64]
            p = int(rng.integers(0, 1850))
            recs.append((f"r{i}", g[p:p + 150]))
        else:
            recs.append((f"r{i}", "".join("ACGT"[x] for x in rng.integers(0, 4, size=150))))
    gf.write_fastq(os.path.join(d, "reads.fq"), recs)
    return dict(dir=d, ibf=os.path.join(d, "c1.ibf"), fq=os.path.join(d, "reads.fq"), built=built, recs=recs)


def _run_config1(binary, c1, out):
    cu.run(binary, ["--ibf", c1["ibf"], "--single-reads", c1["fq"], "-o", out, "--output-all", "--output-unclassified",
                    "--output-stats", "--quiet"])  # reference defaults: rel-cutoff 0.2, rel-filter 0, fpr-query 1
    return out


def test_config1_oracle_backend(oracle_bin, config1, tmp_path):
    import oracle
    p = _run_config1(oracle_bin, config1, str(tmp_path / "c1"))
    res = cu.Res(p, lca_file=False)
    res.sanity_check(output_lca=False)
    assert res.total_classified + res.total_unclassified == 10000 and res.total_classified >= 5000
    lvl = oracle.Level([config1["built"].as_filter(0.2)], 19, 31, rel_filter=0.0, fpr_query=1.0)
    for rid, seq in config1["recs"][:400]:
        rr = lvl.classify(oracle.to_ranks(seq))
        assert res.all.get(rid, {}) == rr.kept, rid


@pytest.mark.gpu
def test_config1_hip_equals_oracle_backend(oracle_bin, config1, tmp_path):
    a = _run_config1(cu.BIN_HIP, config1, str(tmp_path / "hip"))
    b = _run_config1(oracle_bin, config1, str(tmp_path / "ora"))
    for ext in (".all", ".unc", ".rep", ".sta"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext


# ---------------------------------------------------------------------------------------------------------------
# multi-worker pipeline: one classify worker per listed device, results consumed in input order
# ---------------------------------------------------------------------------------------------------------------
def _run_workers(binary, c1, out, devices, batch_reads="500"):
    env = dict(os.environ, GANON_HOST_BATCH_READS=batch_reads)
    p = subprocess.run([binary, "--ibf", c1["ibf"], "--single-reads", c1["fq"], "-o", out, "--output-all", "--output-unclassified",
                        "--output-stats", "--quiet", "--device", devices], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr
    return out


def test_multi_worker_outputs_equal_single_worker_oracle_backend(oracle_bin, config1, tmp_path):
    # 10 000 reads in 20 batches over 1, 2 and 5 workers: every output file byte-identical (GanonClassify.cpp:1579-1614:
    # the reference's workers share one reader and their counters are summed; here the order is kept as well)
    one = _run_workers(oracle_bin, config1, str(tmp_path / "w1"), "0")
    big = _run_config1(oracle_bin, config1, str(tmp_path / "big"))  # one batch
    for devs in ("0,0", "0,0,0,0,0"):
        many = _run_workers(oracle_bin, config1, str(tmp_path / ("w" + str(len(devs)))), devs)
        for ext in (".all", ".unc", ".rep", ".sta"):
            assert open(many + ext, "rb").read() == open(one + ext, "rb").read() == open(big + ext, "rb").read(), (devs, ext)


@pytest.mark.gpu
def test_multi_worker_outputs_equal_single_worker_hip(oracle_bin, config1, tmp_path):
    # the same with HIP backends: --device 0,0 = two workers, two replicas of the filter on one GPU
    ref = _run_workers(oracle_bin, config1, str(tmp_path / "ora"), "0")
    for devs in ("0", "0,0", "all"):
        got = _run_workers(cu.BIN_HIP, config1, str(tmp_path / ("hip" + devs.replace(",", "_"))), devs)
        for ext in (".all", ".unc", ".rep", ".sta"):
            assert open(got + ext, "rb").read() == open(ref + ext, "rb").read(), (devs, ext)


def test_device_argument_errors(oracle_bin, config1, tmp_path):
    for bad in ("x", "0,,1", "-1"):
        p = subprocess.run([oracle_bin, "--ibf", config1["ibf"], "--single-reads", config1["fq"], "-o", str(tmp_path / "e"), "--device", bad],
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 1 and "ERROR" in p.stderr


# ---------------------------------------------------------------------------------------------------------------
# .ibf loader: the sdsl bit_vector header is not pinned by any reference fixture -> plausible variants are accepted,
# anything else is refused; the payload is streamed in chunks
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bv_header", ["wgb", "b", "wb", "gb", "wgq", "q"])
def test_ibf_loader_accepts_bit_vector_header_variants(oracle_bin, config1, tmp_path, bv_header):
    ref = _run_config1(oracle_bin, config1, str(tmp_path / "ref"))
    path = str(tmp_path / f"v_{bv_header}.ibf")
    gf.write_ibf(path, config1["built"], bv_header=bv_header)
    c = dict(config1, ibf=path)
    got = _run_config1(oracle_bin, c, str(tmp_path / "got"))
    for ext in (".all", ".rep"):
        assert open(got + ext, "rb").read() == open(ref + ext, "rb").read()


def test_ibf_loader_refuses_damaged_files(oracle_bin, config1, tmp_path):
    raw = open(config1["ibf"], "rb").read()
    payload = 40009 * 8
    cases = {"truncated": raw[:-4096], "trailing": raw + b"\0" * 16, "bad_size": raw[:-payload - 8] + struct_pack_q(12345) + raw[-payload:],
             "bad_width": raw[:-payload - 13] + b"\x02" + raw[-payload - 12:]}
    for name, blob in cases.items():
        path = str(tmp_path / f"{name}.ibf")
        open(path, "wb").write(blob)
        p = cu.run(oracle_bin, ["--ibf", path, "--single-reads", config1["fq"], "-o", str(tmp_path / name), "--quiet"], check=False)
        assert p.returncode == 1 and "ERROR: loading ibf or tax files" in p.stderr, (name, p.stderr)


def struct_pack_q(v):
    import struct
    return struct.pack("<Q", v)


def test_host_lca_on_reference_vectors(tmp_path):
    # /root/reference/tests/utils/LCA.test.cpp:17-107 against the PRODUCT's ganon_amd/host/lca.hpp (tests/test_oracle_kat.py
    # runs the same vectors against the oracle's LCA)
    subprocess.check_call(["make", "-C", os.path.join(cu.ROOT, "tests", "host_oracle"), "-s", "lca_check"])
    exe = os.path.join(cu.ROOT, "tests", "host_oracle", "lca_check")
    gold = os.path.join(cu.ROOT, "tests", "golden")
    tree = [("D0", "E0,E1"), ("C3", "C3,F4"), ("A0", "G0,C3,D5"), ("1", "G0,G5"), ("B1", "B1,C2"), ("B1", "C2,B1"), ("B0", "C0,E1,F2"),
            ("B0", "F2,E1,C0"), ("B0", "E1,C0,F2")]
    ncbi = [("1224", "366602,470"), ("2", "366602,470,1406"), ("2290931", "2223,51589"), ("10239", "2025595,491893"),
            ("1", "366602,470,1406,2223,51589,2025595,491893")]
    for fn, cases in (("lca_tree.tax", tree), ("lca_ncbi.tax", ncbi)):
        out = subprocess.run([exe, os.path.join(gold, fn), "1"] + [q for _, q in cases], capture_output=True, text=True, check=True, timeout=900)
        assert out.stdout.split() == [w for w, _ in cases]
    # robustness: a self-parent row cannot hang the walk; a node outside the rooted tree resolves to the root
    bad = tmp_path / "bad.tax"
    bad.write_text("1\t1\tno rank\troot\nA\t1\tx\tA\nB\tA\tx\tB\nC\tA\tx\tC\nZ\tY\tx\tdetached\n")
    out = subprocess.run([exe, str(bad), "1", "B,C", "B,Z", "B,unknown"], capture_output=True, text=True, check=True, timeout=20)
    assert out.stdout.split() == ["A", "1", "1"]


# ---------------------------------------------------------------------------------------------------------------
# filters wider than one row group of the flat kernels (> 65 536 technical bins): the binary cuts them into column parts
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def wide_db(tmp_path_factory):
    import numpy as np
    import oracle
    d = str(tmp_path_factory.mktemp("wide"))
    rng = np.random.default_rng(77)
    bins, rows, h = 150_000, 311, 3        # W = 2344 words -> three column parts (cuts moved to target boundaries)
    ibf = gf.random_ibf(bins, rows, h, 0.05, seed=9)
    bin_map, hc = [], []
    b, t = 0, 0
    while b < bins:                        # targets own 1..3 consecutive bins
        run = int(rng.choice([1, 1, 1, 2, 3]))
        run = min(run, bins - b)
        for x in range(run):
            bin_map.append((b + x, f"t{t}"))
        hc.append((f"t{t}", 10 * run))
        b += run
        t += 1
    genomes = {}
    for gi in range(60):
        tb = int(rng.integers(0, bins))
        name = bin_map[tb][1]
        g = "".join("ACGT"[x] for x in rng.integers(0, 4, size=1200))
        genomes[name] = g
        ibf.emplace_many(np.unique(oracle.minimiser_hash(oracle.to_ranks(g), 19, 31)), tb)
    built = gf.BuiltIbf()
    built.ibf = ibf
    built.config = dict(n_bins=bins, max_hashes_bin=10, hash_functions=h, kmer_size=19, window_size=31, bin_size_bits=rows,
                        max_fp=0.05, true_max_fp=0.05, true_avg_fp=0.05)
    built.hashes_count = hc
    built.bin_map = bin_map
    path = os.path.join(d, "wide.ibf")
    gf.write_ibf(path, built)
    recs = []
    names = sorted(genomes)
    for i in range(300):
        if i % 3:
            g = genomes[names[i % len(names)]]
            p = int(rng.integers(0, 1000))
            recs.append((f"r{i}", g[p:p + 150]))
        else:
            recs.append((f"r{i}", "".join("ACGT"[x] for x in rng.integers(0, 4, size=150))))
    fq = os.path.join(d, "reads.fq")
    gf.write_fastq(fq, recs)
    return dict(ibf=path, fq=fq, n_targets=t)


def _run_wide(binary, db, out, extra=()):
    p = cu.run(binary, ["--ibf", db["ibf"], "--single-reads", db["fq"], "-o", out, "--output-all", "--output-unclassified", "--rel-cutoff",
                        "0.5", "--quiet"] + list(extra))
    _run_wide.last_stderr = p.stderr
    return out


def test_wide_filter_oracle_backend(oracle_bin, wide_db, tmp_path):
    res = cu.Res(_run_wide(oracle_bin, wide_db, str(tmp_path / "w")), lca_file=False)
    res.sanity_check(output_lca=False)
    assert res.total_classified >= 190 and res.total_classified + res.total_unclassified == 300


@pytest.mark.gpu
def test_wide_filter_hip_column_parts_equal_oracle_backend(oracle_bin, wide_db, tmp_path, monkeypatch):
    # 150 000 technical bins = 2344 words per row: three device filters behind one --ibf
    a = _run_wide(cu.BIN_HIP, wide_db, str(tmp_path / "hip"))
    b = _run_wide(oracle_bin, wide_db, str(tmp_path / "ora"))
    for ext in (".all", ".unc", ".rep"):
        assert open(a + ext, "rb").read() == open(b + ext, "rb").read(), ext
    # with thresholds that make filter_matches drop things: the column parts take part in ONE joint device pre-pass
    # (their targets are disjoint: parts are cut at target boundaries), and nothing changes in the output
    thr = ["--rel-filter", "0.3", "--fpr-query", "1e-3", "--output-stats"]
    monkeypatch.setenv("GANON_HOST_TIMING", "1")
    c = _run_wide(cu.BIN_HIP, wide_db, str(tmp_path / "hip_thr"), thr)
    assert "pre-pass on the device on (1 filter(s)" in _run_wide.last_stderr
    monkeypatch.delenv("GANON_HOST_TIMING")
    d = _run_wide(oracle_bin, wide_db, str(tmp_path / "ora_thr"), thr)
    for ext in (".all", ".unc", ".rep", ".sta"):
        assert open(c + ext, "rb").read() == open(d + ext, "rb").read(), ext


@pytest.mark.gpu
def test_long_reads_flag(tmp_path):
    # a read with more than 65535 minimisers: skipped by default (the reference's default build, GanonClassify.cpp:45-49,674),
    # classified with --long-reads (its -DLONGREADS=ON build) with a count no 16-bit counter holds
    import numpy as np
    rng = np.random.default_rng(8)
    genome = "".join("ACGT"[x] for x in rng.integers(0, 4, size=640_000))
    other = "".join("ACGT"[x] for x in rng.integers(0, 4, size=30_000))
    built = gf.build_ibf({"BIG.1": genome, "OTHER.1": other}, 19, 31, max_fp=0.01, hash_functions=3)
    ibf = str(tmp_path / "long.ibf")
    gf.write_ibf(ibf, built)
    fa = str(tmp_path / "reads.fa")
    gf.write_fasta(fa, [("long_read", genome[2000:632_000]), ("short_read", genome[100:250]), ("other_read", other[500:700])])
    outs = {}
    for tag, extra in (("default", []), ("long", ["--long-reads"])):
        prefix = str(tmp_path / tag)
        cu.run(cu.BIN_HIP, ["--ibf", ibf, "--single-reads", fa, "-o", prefix, "--output-all", "--output-unclassified", "--skip-lca",
                            "--rel-cutoff", "0.5", "--quiet"] + extra)
        rows = [line.rstrip("\n").split("\t") for line in open(prefix + ".all")]
        outs[tag] = ({r[0]: (r[1], int(r[2])) for r in rows}, open(prefix + ".unc").read().split())
    assert set(outs["default"][0]) == {"short_read", "other_read"} and outs["default"][1] == ["long_read"]
    assert set(outs["long"][0]) == {"long_read", "short_read", "other_read"} and outs["long"][1] == []
    assert outs["long"][0]["long_read"][0] == "BIG.1" and outs["long"][0]["long_read"][1] > 65535
    for r in ("short_read", "other_read"):
        assert outs["long"][0][r] == outs["default"][0][r]


@pytest.mark.gpu
def test_shared_targets_with_more_matches_than_the_device_merges(oracle_bin, tmp_path, monkeypatch):
    # two dense filters that share most of their target names on one level: the device replays the level's merge per read, in a
    # wave up to 512 matches over the level's filters, in a block up to 4096; the 150/250-base reads here match nearly every bin
    # of both filters (4400) and come back untouched (bit 31 of the read's max_count), so the host runs merge and rules on them;
    # the short reads (a few minimisers) match a third to two thirds of the bins and take the block path
    import re
    import numpy as np
    rng = np.random.default_rng(5)
    paths = []
    for fi, (lo, rows, h) in enumerate([(0, 4099, 2), (400, 6007, 2)]):
        bins = 2200
        ibf = gf.random_ibf(bins, rows, h, [0.55, 0.6][fi], seed=20 + fi)
        built = gf.BuiltIbf()
        built.ibf = ibf
        built.config = dict(n_bins=bins, max_hashes_bin=50, hash_functions=h, kmer_size=19, window_size=31, bin_size_bits=rows,
                            max_fp=0.3, true_max_fp=0.3, true_avg_fp=0.3)
        built.hashes_count = [(f"t{lo + b}", 50) for b in range(bins)]
        built.bin_map = [(b, f"t{lo + b}") for b in range(bins)]
        p = str(tmp_path / f"dense{fi}.ibf")
        gf.write_ibf(p, built)
        paths.append(p)
    recs = [(f"r{i}", "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.choice([31, 33, 40, 150, 250])))))
            for i in range(300)]
    fq = str(tmp_path / "reads.fq")
    gf.write_fastq(fq, recs)
    outs = {}
    for tag, binary in (("hip", cu.BIN_HIP), ("oracle", oracle_bin)):
        prefix = str(tmp_path / tag)
        if tag == "hip":
            monkeypatch.setenv("GANON_HOST_TIMING", "1")
        p = cu.run(binary, ["--ibf", ",".join(paths), "--single-reads", fq, "-o", prefix, "--output-all", "--output-unclassified",
                            "--output-stats", "--skip-lca", "--rel-cutoff", "0.2,0.2", "--rel-filter", "0.4", "--fpr-query", "0.6", "--quiet"])
        monkeypatch.delenv("GANON_HOST_TIMING", raising=False)
        if tag == "hip":
            assert "pre-pass on the device on (2 filter(s), targets shared between filters)" in p.stderr, p.stderr[-400:]
            handed_back = int(re.search(r"reads the pre-pass handed back whole (\d+)", p.stderr).group(1))
            assert 10 < handed_back < 290, handed_back  # both kinds of reads are there
        outs[tag] = {ext: open(prefix + ext, "rb").read() for ext in (".all", ".unc", ".rep", ".sta")}
    assert outs["hip"] == outs["oracle"]
    assert outs["hip"][".all"].count(b"\n") > 10000


@pytest.mark.gpu
def test_long_reads_flag_with_an_hibf(tmp_path):
    # the same switch for an HIBF (the level kernels take it: sums do not wrap, reads over 65535 minimisers are counted)
    import numpy as np
    import oracle
    rng = np.random.default_rng(18)
    genome = "".join("ACGT"[x] for x in rng.integers(0, 4, size=640_000))
    other = "".join("ACGT"[x] for x in rng.integers(0, 4, size=30_000))
    names = [f"U{u}" for u in range(40)]
    uh = {5: np.unique(oracle.minimiser_hash(oracle.to_ranks(genome.encode()), 19, 31)),
          21: np.unique(oracle.minimiser_hash(oracle.to_ranks(other.encode()), 19, 31))}
    hb = gf.random_hibf(40, 16, 2, seed=3, density=0.1, hash_funs=2, rows=(70000, 90000), user_hashes=uh)
    path = str(tmp_path / "long.hibf")
    gf.write_hibf(path, hb, [[f"/x/{n}.minimiser"] for n in names], 19, 31, 0.05)
    fa = str(tmp_path / "reads.fa")
    gf.write_fasta(fa, [("long_read", genome[2000:632_000]), ("short_read", genome[100:250]), ("other_read", other[500:700])])
    outs = {}
    for tag, extra in (("default", []), ("long", ["--long-reads"])):
        prefix = str(tmp_path / tag)
        cu.run(cu.BIN_HIP, ["--ibf", path, "--hibf", "--single-reads", fa, "-o", prefix, "--output-all", "--output-unclassified", "--skip-lca",
                            "--rel-cutoff", "0.5", "--quiet"] + extra)
        rows = [line.rstrip("\n").split("\t") for line in open(prefix + ".all")]
        outs[tag] = ({(r[0], r[1]): int(r[2]) for r in rows}, open(prefix + ".unc").read().split())
    assert outs["default"][1] == ["long_read"] and outs["long"][1] == []
    assert {k for k in outs["default"][0]} == {k for k in outs["long"][0] if k[0] != "long_read"}
    assert outs["long"][0][("long_read", "U5")] > 65535
    for key, c in outs["default"][0].items():
        assert outs["long"][0][key] == c
