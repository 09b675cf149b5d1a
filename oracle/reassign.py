"""CPU restatement of `ganon reassign`: the EM over a classification's .all file and the rewrite of its .rep.

TEST INFRASTRUCTURE ONLY -- imported by tests/ (and nothing in ganon_amd/): the checker of the ganon-reassign binary and of
the gn_reassign_* calls.  Pinned by tests/golden/reassign/*, which the reference's own reassign.py produced
(scripts/make_reassign_golden.py, build container only).

What the reference does (/root/reference/src/ganon/reassign.py):
  :35-59   the hierarchies named in column 1 of the .rep's data rows, in first-appearance order, each with
           `<prefix>.<hierarchy>.all`; if that file is missing or empty but `<prefix>.all` exists, ONE table for all rows
           (--output-single); otherwise failure.  `#` rows are kept to be appended to the new .rep.
  :76-92   the table: reads in first-appearance order of their id (a read listed in two places is one read), per read its
           (target, count) entries in file order; targets numbered by first appearance.
  :96-107  weights: a read with exactly one entry is unique; prob[t] = unique[t] / max(1, number of unique reads).
  :110-145 EM: every iteration starts from the unique counts, every other read adds one to the entry get_top_match picks;
           prob[t] = count[t] / number of reads; diff = sum over targets (numbering order, IEEE double, left to right) of
           |old - new|; stop when diff <= threshold, or after --max-iter iterations (0: no limit).
  :226-241 get_top_match: the FIRST listed entry whose prob is strictly larger than every earlier one's and than 0; all zero
           (or nothing unique at all): the first entry.
  :148-181 .one: reads in table order; a unique read keeps its entry, any other gets get_top_match under the probabilities
           of the LAST update (the .rep below holds the counts that update was made from).
  :189-219 .rep: the old data rows whose hierarchy is the table's (any, for the single table) and whose target occurs in
           the table: `hierarchy, target, direct matches, unique, count - unique, rank, name` -- always seven tab-separated
           fields, missing rank / name empty; then the `#` rows, right-stripped.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

_WS = " \t\n\r\x0b\x0c"


@dataclass
class Table:
    """one .all file as arrays: CSR over reads, entries in file order"""
    read_ids: List[str]
    target_names: List[str]
    off: np.ndarray        # int64[n_reads + 1]
    target: np.ndarray     # int64[n_entries]
    count: np.ndarray      # int64[n_entries]  (the k-mer count column, carried to .one)
    index: Dict[str, int] = field(default_factory=dict)  # target name -> number


def read_table(path: str) -> Table:
    """:76-92"""
    reads: Dict[str, List[Tuple[int, int]]] = {}
    tindex: Dict[str, int] = {}
    with open(path, "r") as f:
        for line in f:
            parts = line.rstrip(_WS).split("\t")
            if len(parts) != 3:
                raise ValueError(f"{path}: a line of {len(parts)} fields")
            rid, tname, c = parts
            t = tindex.setdefault(tname, len(tindex))
            reads.setdefault(rid, []).append((t, int(c)))
    ids = list(reads)
    lens = np.fromiter((len(reads[r]) for r in ids), dtype=np.int64, count=len(ids))
    off = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = [e for r in ids for e in reads[r]]
    tgt = np.fromiter((e[0] for e in flat), dtype=np.int64, count=len(flat))
    cnt = np.fromiter((e[1] for e in flat), dtype=np.int64, count=len(flat))
    return Table(ids, list(tindex), off, tgt, cnt, tindex)


def unique_counts(tb: Table) -> np.ndarray:
    """:96-103 -- per target the reads that list it and nothing else"""
    single = np.nonzero(np.diff(tb.off) == 1)[0]
    return np.bincount(tb.target[tb.off[single]], minlength=len(tb.target_names)).astype(np.int64)


def top_entries(tb: Table, prob: np.ndarray) -> np.ndarray:
    """get_top_match (:226-241) for every read at once -> index of the chosen entry in tb.target / tb.count.
    The reference walks a read's entries and moves on a strictly larger probability, starting from 0: that is the first
    entry holding the read's maximum if the maximum is positive, the first entry otherwise."""
    n = len(tb.read_ids)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    p = prob[tb.target]
    starts = tb.off[:-1]
    seg_max = np.maximum.reduceat(p, starts)
    owner = np.repeat(np.arange(n), np.diff(tb.off))
    at_max = p == seg_max[owner]
    pos = np.where(at_max, np.arange(len(p)), len(p))
    first_at_max = np.minimum.reduceat(pos, starts)
    return np.where(seg_max > 0.0, first_at_max, starts)


@dataclass
class EmResult:
    iterations: int
    diffs: List[float]
    counts: np.ndarray     # reassigned_matches of the last iteration (:113-121), int64[n_targets]
    prob: np.ndarray       # after the last update (:126-129)
    choice: np.ndarray     # entry index per read under `prob` (:170-181)


def em(tb: Table, max_iter: int = 10, threshold: float = 0) -> EmResult:
    """:96-145"""
    nt = len(tb.target_names)
    n_reads = len(tb.read_ids)
    uniq = unique_counts(tb)
    multi = np.nonzero(np.diff(tb.off) > 1)[0]
    n_unique = int(uniq.sum())
    denom = n_unique if n_unique else 1
    prob = np.array([int(u) / denom for u in uniq], dtype=np.float64)
    diffs: List[float] = []
    it = 0
    while True:
        chosen = top_entries(tb, prob)
        counts = uniq + np.bincount(tb.target[chosen[multi]], minlength=nt).astype(np.int64)
        diff = 0.0
        new = np.empty(nt, dtype=np.float64)
        for t in range(nt):  # left to right, as the reference's loop over its dict (:125-129)
            new[t] = int(counts[t]) / n_reads
            diff += abs(float(prob[t]) - float(new[t]))
        prob = new
        diffs.append(diff)
        if diff <= threshold:
            break
        if max_iter > 0 and it == max_iter - 1:
            break
        it += 1
    return EmResult(it + 1, diffs, counts, prob, top_entries(tb, prob))


def one_text(tb: Table, res: EmResult) -> str:
    """:153-181"""
    out = []
    for r, rid in enumerate(tb.read_ids):
        e = int(tb.off[r]) if tb.off[r + 1] - tb.off[r] == 1 else int(res.choice[r])
        out.append(f"{rid}\t{tb.target_names[int(tb.target[e])]}\t{int(tb.count[e])}\n")
    return "".join(out)


def find_tables(rep_path: str) -> Tuple[Optional[Dict[str, str]], List[str]]:
    """:35-59 -> ({hierarchy: .all path} or None when a table is missing, the `#` rows right-stripped)"""
    prefix = rep_path[:-4] if rep_path.endswith(".rep") else rep_path
    hier: Dict[str, str] = {}
    info: List[str] = []
    with open(rep_path) as f:
        for line in f:
            if line[0] != "#":
                hier[line.split("\t")[0]] = ""
            else:
                info.append(line.rstrip(_WS))

    def present(p):
        return os.path.isfile(p) and os.path.getsize(p) > 0

    for h in list(hier):
        if present(f"{prefix}.{h}.all"):
            hier[h] = f"{prefix}.{h}.all"
        elif present(prefix + ".all"):
            return {"": prefix + ".all"}, info
        else:
            return None, info
    return hier, info


def rep_rows(rep_path: str, hierarchy: str, tb: Table, res: EmResult) -> List[str]:
    """:189-219 for one table"""
    rows = []
    with open(rep_path) as f:
        for line in f:
            if line[0] == "#":
                continue
            fld = line.rstrip(_WS).split("\t")
            if hierarchy != "" and fld[0] != hierarchy:
                continue
            t = tb.index.get(fld[1])
            if t is None:
                continue
            unique = int(fld[3])
            rank = fld[5] if len(fld) >= 6 else ""
            name = fld[6] if len(fld) >= 7 else ""
            rows.append("\t".join([fld[0], fld[1], fld[2], str(unique), str(int(res.counts[t]) - unique), rank, name]) + "\n")
    return rows


def reassign_files(rep_path: str, max_iter: int = 10, threshold: float = 0):
    """-> (new .rep text, {hierarchy: .one text}, [EmResult per hierarchy]) or None where the reference returns False"""
    tables, info = find_tables(rep_path)
    if tables is None:
        return None
    rep, ones, results = [], {}, []
    for h, path in tables.items():
        tb = read_table(path)
        res = em(tb, max_iter, threshold)
        ones[h] = one_text(tb, res)
        rep.extend(rep_rows(rep_path, h, tb, res))
        results.append(res)
    return "".join(rep) + "".join(i + "\n" for i in info), ones, results
