#!/usr/bin/env python3
"""First contact with a real ganon install (closes SURVEY 8 a-11 and f-1 the day anyone has one).

  python scripts/first_contact.py <dir with the reference's ganon-build and ganon-classify> [--work DIR] [--device N]

Nothing in this repo has ever read a byte written by SeqAn3 / cereal / robin_hood: the .ibf layout, the IBF hash constants,
the id-token rule and the hash-map iteration order are restated from the published code.  This script settles all of that in
one run, on a 40-target database and the reference's own 98-pair FASTQ fixture (tests/golden/sim.{1,2}.fq.gz):

  build      the same references through BOTH builders (k=19 w=31 h=4 max-fp 0.05)
  inspect    ours reads THEIR file's header: every field, every redundancy (ganon-classify --inspect-filter)
  verify     every minimiser of every reference is found in THEIR file's bins for its target (--verify-filter: the check of
             the reference's build test, on the device) -- pins the IBF hash constants, hash_shift and the row/bin layout
  cross      THEIR ganon-classify loads OUR file (their reader, our writer)
  classify   both classifiers x both filters, --threads 1; .rep/.all/.one/.unc compared as SORTED lines (semantics) and
             BYTE for byte with our --reference-order (the robin_hood iteration order model, host/robin_order.hpp)
  header     IBFConfig of the two files field by field (sizes are floating-point fragile: informational)

Prints a verdict table and exits 0 only when every required row passes.  `--their-classify-args` exists for this repo's own
test, where our binaries stand in for "theirs" (tests/test_first_contact.py)."""
from __future__ import annotations

import argparse
import gzip
import os
import shlex
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUR_BUILD = os.path.join(ROOT, "ganon_amd", "host", "ganon-build")
OUR_CLASSIFY = os.path.join(ROOT, "ganon_amd", "host", "ganon-classify")
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXTS = (".rep", ".all", ".one", ".unc")


def sim_reads():
    out = []
    for fn in ("sim.1.fq.gz", "sim.2.fq.gz"):
        with gzip.open(os.path.join(GOLDEN, fn), "rt") as f:
            lines = f.read().split("\n")
        out.append([lines[i + 1] for i in range(0, len(lines) - 3, 4)])
    return out


def write_references(d: str):
    """40 synthetic genomes that contain the fixture's reads verbatim (so that the reads classify), one FASTA per target;
    every fifth target is long enough to be split over several technical bins"""
    r1, r2 = sim_reads()
    rng = np.random.default_rng(99)
    os.makedirs(os.path.join(d, "refs"), exist_ok=True)
    lines, tax = [], {}
    for t in range(40):
        parts = []
        for j in range(6):
            parts.append("".join("ACGT"[x] for x in rng.integers(0, 4, size=300 if t % 5 else 4000)))
            idx = (t * 6 + j) % len(r1)
            parts.append(r1[idx] if j % 2 == 0 else r2[idx])
        name = f"T{t}.1"
        path = os.path.join(d, "refs", f"{name}.fna")
        seq = "".join(parts)
        with open(path, "w") as f:
            f.write(f">{name} synthetic\n")
            for i in range(0, len(seq), 80):
                f.write(seq[i:i + 80] + "\n")
        lines.append(f"{path}\t{name}")
        tax[name] = f"G{t % 5}"
    for g in range(5):
        tax[f"G{g}"] = "1"
    with open(os.path.join(d, "input.tsv"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(d, "db.tax"), "w") as f:
        f.write("1\t0\troot\troot\n")
        for node, parent in tax.items():
            f.write(f"{node}\t{parent}\t{'genus' if node.startswith('G') else 'assembly'}\t{node}\n")
    return os.path.join(d, "input.tsv"), os.path.join(d, "db.tax")


def run(cmd, log, timeout=900):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    log.append(f"$ {' '.join(shlex.quote(c) for c in cmd)}\n  rc {p.returncode}\n" + "".join("  | " + ln + "\n" for ln in (p.stdout + p.stderr).splitlines()[-12:]))
    return p


def inspect_fields(text: str) -> dict:
    out = {}
    for ln in text.splitlines():
        if ln.startswith("@") and " = " in ln:
            out[ln[12:].split(" = ")[0].strip()] = ln.split(" = ", 1)[1].split("   ")[0].strip()
    return out


def sorted_lines(path):
    return sorted(open(path, "rb").read().splitlines()) if os.path.exists(path) else None


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("their_dir", help="directory holding the reference's ganon-build and ganon-classify")
    ap.add_argument("--work", default="")
    ap.add_argument("--device", default="0")
    ap.add_argument("--their-classify-args", default="", help="extra arguments for THEIR ganon-classify (this repo's self-test only)")
    args = ap.parse_args()
    their_build, their_classify = (os.path.join(args.their_dir, b) for b in ("ganon-build", "ganon-classify"))
    for b in (their_build, their_classify, OUR_BUILD, OUR_CLASSIFY):
        if not os.access(b, os.X_OK):
            print(f"first_contact: {b} is missing or not executable", file=sys.stderr)
            return 2
    work = args.work or tempfile.mkdtemp(prefix="ganon_first_contact_")
    os.makedirs(work, exist_ok=True)
    log, rows = [], []          # rows: (step, required, verdict, note)
    inp, tax = write_references(work)
    build_flags = ["--kmer-size", "19", "--window-size", "31", "--hash-functions", "4", "--max-fp", "0.05", "--quiet"]
    ibf = {"ours": os.path.join(work, "ours.ibf"), "theirs": os.path.join(work, "theirs.ibf")}

    p = run([OUR_BUILD, "--input-file", inp, "--output-file", ibf["ours"], "--device", args.device] + build_flags, log)
    rows.append(("build: our ganon-build", True, p.returncode == 0 and os.path.exists(ibf["ours"]), ""))
    p = run([their_build, "--input-file", inp, "--output-file", ibf["theirs"], "--threads", "1"] + build_flags
            + (["--device", args.device] if os.path.samefile(their_build, OUR_BUILD) else []), log)
    rows.append(("build: THEIR ganon-build", True, p.returncode == 0 and os.path.exists(ibf["theirs"]), ""))

    pi = {}
    for who in ("ours", "theirs"):
        pi[who] = run([OUR_CLASSIFY, "--inspect-filter", ibf[who]], log)
    rows.append(("inspect: our reader <- THEIR file (every header field and redundancy)", True, pi["theirs"].returncode == 0,
                 (pi["theirs"].stdout.splitlines() or ["no output"])[-1][:110]))
    fo, ft = inspect_fields(pi["ours"].stdout), inspect_fields(pi["theirs"].stdout)
    diff = [k for k in fo if k.startswith(("IBFConfig", "ibf.")) and fo.get(k) != ft.get(k)]
    rows.append(("header: IBFConfig and IBF fields equal in both files (sizes are fp-fragile)", False, not diff and bool(ft),
                 "differs: " + ", ".join(f"{k} {fo[k]} vs {ft.get(k)}" for k in diff[:4]) if diff else ""))
    same_bits = os.path.exists(ibf["ours"]) and os.path.exists(ibf["theirs"]) and open(ibf["ours"], "rb").read() == open(ibf["theirs"], "rb").read()
    rows.append(("bits: the two files are byte-identical (split targets may legitimately differ)", False, same_bits, ""))

    p = run([OUR_CLASSIFY, "--ibf", ibf["theirs"], "--verify-filter", inp, "--device", args.device], log)
    rows.append(("verify: every reference minimiser is in its target's bins of THEIR file (device lookup)", True, p.returncode == 0,
                 (p.stdout.splitlines() or ["no output"])[-1][:110]))
    p = run([OUR_CLASSIFY, "--ibf", ibf["ours"], "--verify-filter", inp, "--device", args.device], log)
    rows.append(("verify: ... of OUR file (control)", True, p.returncode == 0, ""))

    fq = os.path.join(GOLDEN, "sim.1.fq.gz") + "," + os.path.join(GOLDEN, "sim.2.fq.gz")
    common = ["--tax", tax, "--paired-reads", fq, "--output-all", "--output-lca", "--output-unclassified", "--threads", "1", "--quiet",
              "--rel-cutoff", "0.25", "--rel-filter", "0.1"]
    outs = {}
    for flt in ("ours", "theirs"):
        for who, binary, extra in (("ours", OUR_CLASSIFY, ["--device", args.device]),
                                   ("ours_reforder", OUR_CLASSIFY, ["--device", args.device, "--reference-order"]),
                                   ("theirs", their_classify, shlex.split(args.their_classify_args))):
            prefix = os.path.join(work, f"out_{who}_on_{flt}")
            p = run([binary, "--ibf", ibf[flt], "-o", prefix] + common + extra, log)
            outs[(who, flt)] = prefix if p.returncode == 0 and os.path.exists(prefix + ".rep") else None
    rows.append(("cross: THEIR ganon-classify loads OUR file and classifies", True, outs[("theirs", "ours")] is not None, ""))
    rows.append(("cross: our ganon-classify loads THEIR file and classifies", True, outs[("ours", "theirs")] is not None, ""))
    for flt in ("theirs", "ours"):
        a, b, c = outs[("ours", flt)], outs[("theirs", flt)], outs[("ours_reforder", flt)]
        for ext in EXTS:
            if a and b:
                la, lb = sorted_lines(a + ext), sorted_lines(b + ext)
                ok = la is not None and la == lb
                note = "" if ok else f"{0 if la is None else len(la)} vs {0 if lb is None else len(lb)} lines"
            else:
                ok, note = False, "a run failed"
            rows.append((f"classify on {flt}.ibf: {ext} equal as sorted lines (ours vs THEIRS)", True, ok, note))
        for ext in EXTS:
            ok = bool(b and c) and os.path.exists(b + ext) and os.path.exists(c + ext) and open(b + ext, "rb").read() == open(c + ext, "rb").read()
            rows.append((f"classify on {flt}.ibf: {ext} byte-identical with our --reference-order (--threads 1)", False, ok, ""))
    nonempty = bool(outs[("ours", "ours")]) and os.path.getsize(outs[("ours", "ours")] + ".all") > 0
    rows.append(("sanity: the fixture's reads do classify (non-empty .all)", True, nonempty, ""))

    with open(os.path.join(work, "first_contact.log"), "w") as f:
        f.write("\n".join(log))
    width = max(len(r[0]) for r in rows)
    print(f"first contact with {args.their_dir}  (work dir {work}, commands and output tails in first_contact.log)")
    for step, required, ok, note in rows:
        print(f"  {'PASS' if ok else ('FAIL' if required else 'diff'):4}  {'required' if required else 'info    '}  {step.ljust(width)}  {note}")
    failed = [r for r in rows if r[1] and not r[2]]
    info = [r for r in rows if not r[1] and not r[2]]
    print(f"verdict: {'ALL REQUIRED ROWS PASS' if not failed else str(len(failed)) + ' REQUIRED ROW(S) FAIL'}; {len(info)} informational row(s) differ")
    return 0 if not failed else 1


if __name__ == "__main__":
    sys.exit(main())
