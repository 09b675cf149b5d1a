#!/usr/bin/env python3
"""bench_e2e.py -- the product binary end to end (files in, files out), the leg bench.py attaches as `e2e`.

What `ganon-classify` itself times as "classifying+printing elapsed" (/root/reference/src/ganon-classify/GanonClassify.cpp:1047,
1095-1099): FASTQ text from /dev/shm -> parse / tokenise -> GPU -> filter_matches -> .all / .rep written, with the filter load
reported beside it.  Four inputs, every one built so that about half of the reads CLASSIFY (reads / pairs cut from the
genomes the filter holds, mate 2 the reverse strand of the same fragment):

  fastq    plain single-end FASTQ against a 1 GiB flat IBF (4096 bins, h = 4)
  paired   two FASTQ files, 2 x 150 bp ends of 400 bp fragments, same filter
  gz       single-end .fq.gz -- ONE gzip member whose deflate blocks reference the 32 KiB before them, as gzip / pigz write it
           (written here in parallel with zlib's preset-dictionary interface, which produces exactly that), binned qualities
  fasta    the same reads as a FASTA file (two lines per record), same filter
  hibf     the plain FASTQ against a two-level HIBF (16 384 user bins) that holds the same genomes: level 1 is visited

Every input runs `--runs` times (default 5); median, min and max are reported with `#total_classified` of the .rep.
usage: python bench_e2e.py [--runs 5] [--reads 16000000] [--dir /dev/shm] [--budget 150]   -> one JSON line
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import json
import os
import re
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EXE = os.path.join(ROOT, "ganon_amd", "host", "ganon-classify")
# what `ganon classify` passes to the binary by default (/root/reference/src/ganon/config.py: --rel-cutoff 0.75 --rel-filter 0.1
# --fpr-query 1e-5)
THRESHOLDS = ["--rel-cutoff", "0.75", "--rel-filter", "0.1", "--fpr-query", "1e-5"]
QUALS = np.frombuffer(b"FFFFFFFFFFFF:FFF,FFFF#", dtype=np.uint8)  # binned qualities as sequencers write them


def fastq_matrix(bases: np.ndarray, n: int, L: int, first_id: int = 0, quals: bool = False) -> np.ndarray:
    """n records `@r%09d \\n bases \\n + \\n qualities \\n` as one byte matrix (fixed width: ids are zero-padded)"""
    rec = np.empty((n, 2 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0], rec[:, 1] = ord("@"), ord("r")
    idx = np.arange(first_id, first_id + n, dtype=np.int64)
    for p in range(9):
        rec[:, 2 + p] = (idx // 10 ** (8 - p)) % 10 + ord("0")
    rec[:, 11] = ord("\n")
    rec[:, 12:12 + L] = bases.reshape(n, L)
    rec[:, 12 + L:15 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    if quals:
        rec[:, 15 + L:15 + 2 * L] = QUALS[np.random.default_rng(7).integers(0, len(QUALS), size=(n, L), dtype=np.uint8)]
    else:
        rec[:, 15 + L:15 + 2 * L] = ord("I")
    rec[:, -1] = ord("\n")
    return rec


def write_gzip_one_member(path: str, data: np.ndarray, level: int = 6, chunk: int = 8 << 20, threads: int = 16) -> None:
    """`data` as ONE gzip member: raw deflate pieces compressed in parallel, every piece primed with the 32 KiB before it
    (back-references cross the pieces as in a stream written in one go), Z_SYNC_FLUSH between them, Z_FINISH at the end"""
    mv = memoryview(data.reshape(-1))
    n = len(mv)
    starts = list(range(0, n, chunk)) or [0]

    def piece(i):
        a = starts[i]
        b = min(n, a + chunk)
        kw = dict(zdict=bytes(mv[max(0, a - 32768):a])) if a else {}
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, **kw)
        return co.compress(mv[a:b]) + co.flush(zlib.Z_FINISH if b == n else zlib.Z_SYNC_FLUSH)

    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(piece, range(len(starts))))
    crc = 0
    for a in starts:
        crc = zlib.crc32(mv[a:min(n, a + chunk)], crc)
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00" + struct.pack("<I", 0) + b"\x00\x03")
        for p in parts:
            f.write(p)
        f.write(struct.pack("<II", crc & 0xFFFFFFFF, n & 0xFFFFFFFF))


EXTRA_ENV = {}   # --env K=V


def cgroup_throttle():
    """(periods in which this container's CPU quota stopped it, microseconds it was stopped for) so far -- cgroup v2 cpu.stat"""
    n = us = 0
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = ln.partition(" ")
            if k == "nr_throttled":
                n = int(v)
            elif k == "throttled_usec":
                us = int(v)
    except OSError:
        pass
    return n, us


def run_binary(args, n_units, runs, label, deadline, env_extra=None):
    """-> summary of `runs` runs of the binary (fewer when the time budget runs out: the number is reported)"""
    secs, loads, walls = [], [], []
    last = {}
    env = dict(os.environ, GANON_HOST_TIMING="1")
    env.update(EXTRA_ENV)
    env.update(env_extra or {})
    prefix = args[args.index("-o") + 1]
    per_run = []
    for i in range(runs):
        if i >= 1 and time.time() > deadline:
            break
        thr0 = cgroup_throttle()
        t0 = time.time()
        p = subprocess.run([EXE] + args, capture_output=True, text=True, env=env, timeout=600)
        walls.append(time.time() - t0)
        thr1 = cgroup_throttle()
        if p.returncode != 0:
            return {"error": f"rc {p.returncode}: {p.stderr[-300:]}"}
        m = re.search(r"classifying\+printing elapsed \(s\): ([0-9.eE+-]+)", p.stderr)
        if not m:
            return {"error": "no timing line in the binary's --verbose output"}
        secs.append(float(m.group(1)))
        m = re.search(r"loading filter\(s\)\s+elapsed \(s\): ([0-9.eE+-]+)", p.stderr)
        loads.append(float(m.group(1)) if m else 0.0)
        last = {"stderr": p.stderr}
        # per run: the timed seconds, what the cgroup's CPU quota throttled meanwhile, and the stall line -- a slow run names its stall
        st = re.findall(r"\[host stalls\] level [^:]*: (.*?); reads the pre-pass", p.stderr)
        per_run.append({"s": round(secs[-1], 4), "wall_s": round(walls[-1], 3),
                        "throttled_periods": thr1[0] - thr0[0], "throttled_ms": round((thr1[1] - thr0[1]) / 1000.0, 1),
                        "stalls": (st[-1] if st else "")[:260]})
    rep = open(prefix + ".rep").read().splitlines()
    classified = next((int(l.split("\t")[1]) for l in rep if l.startswith("#total_classified")), 0)
    unclassified = next((int(l.split("\t")[1]) for l in rep if l.startswith("#total_unclassified")), 0)
    with open(prefix + ".all", "rb") as f:
        all_lines = sum(buf.count(b"\n") for buf in iter(lambda: f.read(1 << 24), b""))
    stalls = re.findall(r"\[host stalls\] (.*)", last.get("stderr", ""))
    cpu = re.findall(r"\[host cpu\] seconds user \+ system: (.*)", last.get("stderr", ""))
    rates = [n_units / s / 1e6 for s in secs]
    out = {"input": label, "units": n_units, "runs": len(secs),
           "classify_print_s": {"median": round(float(np.median(secs)), 4), "min": round(min(secs), 4), "max": round(max(secs), 4)},
           "rate": {"median": round(float(np.median(rates)), 2), "min": round(min(rates), 2), "max": round(max(rates), 2)},
           "load_filter_s_median": round(float(np.median(loads)), 3), "process_wall_s_median": round(float(np.median(walls)), 2),
           "total_classified": classified, "total_unclassified": unclassified,
           "classified_frac": round(classified / max(1, classified + unclassified), 4), "all_lines": all_lines,
           "host_stalls": "; ".join(stalls)[:400]}
    if cpu:   # CPU seconds (user + system) per group of host threads over the LAST run, and per million units
        groups = {}
        for name, u, sy in re.findall(r"([a-z ]+?) ([0-9.eE+-]+) \+ ([0-9.eE+-]+)(?:,|;| on)", cpu[-1]):
            groups[name.strip().replace(" ", "_")] = round(float(u) + float(sy), 4)
        out["host_cpu_s"] = groups
        out["host_cpu_s_per_munit"] = {k: round(v / (n_units / 1e6), 5) for k, v in groups.items()}
    out["per_run"] = per_run
    if os.environ.get("E2E_DIAG"):  # every timing line of the last run (GANON_HOST_TIMING=1)
        out["timing_lines"] = [l[:600] for l in last.get("stderr", "").splitlines() if l.startswith("[")]
    for ext in (".all", ".rep"):
        if os.path.exists(prefix + ext):
            os.remove(prefix + ext)
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--reads", type=int, default=16_000_000, help="single-end reads (pairs and .gz reads: half of it)")
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--budget", type=float, default=200.0, help="seconds; inputs that no longer fit are left out and named")
    ap.add_argument("--keep-gz", default="", help="copy the filter and the single-end .fq.gz into this directory before they are removed (profiling runs)")
    ap.add_argument("--only", default="", help="comma-separated subset of fastq,paired,gz,fasta,hibf")
    ap.add_argument("--env", action="append", default=[], help="K=V for the binary's environment, e.g. GANON_HIP_ABLATE=fake_count (scripts/host_ceiling.py)")
    ap.add_argument("--devices", default="", help="passed to the binary as --device (e.g. `all`, `0,1,2,3`, `0,0` = two workers on one GPU): one "
                                                  "classify worker per entry over ONE reader, as the reference runs N threads over one parser "
                                                  "(GanonClassify.cpp:1436-1441,1579-1597)")
    args = ap.parse_args()

    import bench_workload as bw
    import ganon_amd
    from ganon_amd import ibf_file

    EXTRA_ENV.update(dict(kv.split("=", 1) for kv in args.env))
    t_start = time.time()
    deadline = t_start + args.budget
    want = [w for w in (args.only.split(",") if args.only else ["fastq", "paired", "gz", "fasta", "hibf"]) if w]
    d = os.path.join(args.dir, f"ganon_e2e_{os.getpid()}")
    os.makedirs(d, exist_ok=True)
    out = {"thresholds": " ".join(THRESHOLDS), "dir": args.dir, "runs_requested": args.runs, "inputs": {}, "skipped": []}
    bins, rows, h, L, n_genomes = 4096, 1 << 21, 4, 150, 4096
    n = args.reads
    try:
        # ---- the flat filter (1 GiB) with the genomes planted, written as a ganon-build .ibf
        wl = bw.make_device_flat_workload("e2e", bins, rows, h, n, paired=False, seed=42)
        flt, _ = bw.device_filter(ganon_amd, wl)
        ibf = os.path.join(d, "e2e.ibf")
        # Bernoulli(0.5) bits = every bin a Bloom filter at its optimal load for h = 4: the header declares the number of
        # minimisers per bin that gives that rate (n = S ln2 / h), so --fpr-query sees the bits that are there
        per_bin = int(rows * 0.6931471805599453 / h)
        cfg = dict(n_bins=bins, max_hashes_bin=per_bin, hash_functions=h, kmer_size=wl.k, window_size=wl.w, bin_size_bits=rows, max_fp=0.0625,
                   true_max_fp=0.0625, true_avg_fp=0.0625)
        ibf_file.save_ibf(ibf, flt, cfg, [(f"T{b}", per_bin) for b in range(bins)], [(b, f"T{b}") for b in range(bins)], bins, rows, h)
        flt.free()
        out["filter"] = {"kind": "flat IBF", "gib": round(os.path.getsize(ibf) / 2**30, 2), "bins": bins, "hash_funs": h, "planted_genomes": n_genomes}

        fq = os.path.join(d, "single.fq")
        fastq_matrix(wl.bases, n, L).tofile(fq)
        out["fastq_gib"] = round(os.path.getsize(fq) / 2**30, 2)
        with open(fq, "rb") as fh:  # (the first reader of a file just written pays for it; not the runs)
            while fh.read(1 << 26):
                pass
        common = ["--output-all", "--verbose"] + THRESHOLDS + (["--device", args.devices] if args.devices else [])
        if args.devices:
            out["devices"] = args.devices
        if EXTRA_ENV:
            out["env"] = dict(EXTRA_ENV)
        if "fastq" in want:
            out["inputs"]["fastq"] = run_binary(["--ibf", ibf, "--single-reads", fq, "-o", os.path.join(d, "o_fastq")] + common, n, args.runs,
                                                f"{n} reads x {L} bp, plain FASTQ", deadline)
        if "fasta" in want and time.time() < deadline - 15:
            fa = os.path.join(d, "single.fa")
            m = fastq_matrix(wl.bases, n, L)[:, :12 + L + 1].copy()   # `@id \n letters \n` of every record, '@' -> '>'
            m[:, 0] = ord(">")
            m.tofile(fa)
            del m
            out["inputs"]["fasta"] = run_binary(["--ibf", ibf, "--single-reads", fa, "-o", os.path.join(d, "o_fasta")] + common, n, args.runs,
                                                f"{n} reads x {L} bp, FASTA (one line of letters per record)", deadline)
            os.remove(fa)
        elif "fasta" in want:
            out["skipped"].append("fasta")
        if "gz" in want and time.time() < deadline - 25:
            ng = n // 2
            gz = os.path.join(d, "single.fq.gz")
            t0 = time.time()
            write_gzip_one_member(gz, fastq_matrix(wl.bases[: ng * L], ng, L, quals=True))
            r = run_binary(["--ibf", ibf, "--single-reads", gz, "-o", os.path.join(d, "o_gz")] + common, ng, args.runs,
                           f"{ng} reads x {L} bp, one-member .fq.gz ({os.path.getsize(gz) / 2**30:.2f} GiB, level 6, written in {time.time() - t0:.1f} s)", deadline)
            out["inputs"]["gz"] = r
            if args.keep_gz:
                import shutil
                os.makedirs(args.keep_gz, exist_ok=True)
                shutil.copy(gz, os.path.join(args.keep_gz, "single.fq.gz"))
                shutil.copy(ibf, os.path.join(args.keep_gz, "e2e.ibf"))
            m = re.search(r"\[host input\] .*(device inflate: .*)", "\n".join(r.get("timing_lines", [])) or "")
            # the same file through the host's parallel inflater (pgzip.cpp), as up to round 4
            if time.time() < deadline - 10:
                out["inputs"]["gz_host_inflate"] = run_binary(["--ibf", ibf, "--single-reads", gz, "-o", os.path.join(d, "o_gzh")] + common, ng, max(1, args.runs // 2),
                                                              "the same file, inflated by the host's threads ($GANON_HOST_DEVICE_INFLATE=0)", deadline,
                                                              {"GANON_HOST_DEVICE_INFLATE": "0"})
            os.remove(gz)
        elif "gz" in want:
            out["skipped"].append("gz")
        if "paired" in want and time.time() < deadline - 20:
            # as many PAIRS as single-end reads (round 5: half): the timed part of 8 M pairs was 0.10-0.15 s, a fifth of which is how far the
            # reader got ahead while the filter loaded -- too short a window for a stable figure
            npair = n if args.reads >= 16_000_000 else n // 2
            wp = bw.make_device_flat_workload("e2e", bins, rows, h, npair, paired=True, seed=42, shard=1)
            f1, f2 = os.path.join(d, "pair.1.fq"), os.path.join(d, "pair.2.fq")
            fastq_matrix(wp.bases[: npair * L], npair, L).tofile(f1)
            fastq_matrix(wp.bases[npair * L:], npair, L).tofile(f2)
            for fw in (f1, f2):   # (as for the single-end file: the first reader of a file just written pays for it, not the runs --
                with open(fw, "rb") as fh:   # round 5's "25.8-83 Mpairs/s" had the first run of every series in it)
                    while fh.read(1 << 26):
                        pass
            nz = min(npair // 2, n // 4)
            zb1, zb2 = wp.bases[: nz * L].copy(), wp.bases[npair * L: (npair + nz) * L].copy()  # (the first half of the pairs, for the .fq.gz leg)
            del wp
            out["inputs"]["paired"] = run_binary(["--ibf", ibf, "--paired-reads", f1 + "," + f2, "-o", os.path.join(d, "o_paired")] + common, npair,
                                                 args.runs, f"{npair} pairs 2 x {L} bp (ends of 400 bp fragments, mate 2 reverse strand), two FASTQ files", deadline)
            if "gz" in want and time.time() < deadline - 40:
                # the same pairs as two one-member .fq.gz files (half of them: the compression is this script's, and slow)
                z1, z2 = os.path.join(d, "pair.1.fq.gz"), os.path.join(d, "pair.2.fq.gz")
                t0 = time.time()
                write_gzip_one_member(z1, fastq_matrix(zb1, nz, L, quals=True))
                write_gzip_one_member(z2, fastq_matrix(zb2, nz, L, quals=True))
                lab = f"{nz} pairs 2 x {L} bp, two one-member .fq.gz files (level 6, written in {time.time() - t0:.1f} s)"
                out["inputs"]["paired_gz"] = run_binary(["--ibf", ibf, "--paired-reads", z1 + "," + z2, "-o", os.path.join(d, "o_pgz")] + common, nz, args.runs, lab, deadline)
                if time.time() < deadline - 15:
                    out["inputs"]["paired_gz_host_inflate"] = run_binary(["--ibf", ibf, "--paired-reads", z1 + "," + z2, "-o", os.path.join(d, "o_pgzh")] + common, nz,
                                                                         max(1, args.runs // 2), "the same files, inflated by the host's threads", deadline,
                                                                         {"GANON_HOST_DEVICE_INFLATE": "0"})
                os.remove(z1)
                os.remove(z2)
            os.remove(f1)
            os.remove(f2)
        elif "paired" in want:
            out["skipped"].append("paired")
        os.remove(ibf)
        if "hibf" in want and time.time() < deadline - 25:
            # two-level HIBF holding the SAME genomes (same seed): 128 merged bins -> 128 children x 128 user bins
            ub, tmax, rows_top, rows_child, hh = 16384, 128, 1 << 20, 1 << 18, 3
            hw, hf = bw.make_hibf_device_workload(ganon_amd, "e2e_hibf", ub, tmax, rows_top, rows_child, hh, 1024, seed=42)
            hp = os.path.join(d, "e2e.hibf")
            ibf_file.save_hibf(hp, hf, [(b, r, x) for (_, b, r, x) in hw.ibfs], hw.next_ibf_id, hw.bin_to_user, [f"U{u}.1" for u in range(ub)],
                               hw.k, hw.w, 0.125)
            hf.free()
            r = run_binary(["--ibf", hp, "--hibf", "--single-reads", fq, "-o", os.path.join(d, "o_hibf")] + common, n, args.runs,
                           f"{n} reads x {L} bp, plain FASTQ, against a 2-level HIBF of {ub} user bins ({os.path.getsize(hp) / 2**30:.2f} GiB)", deadline)
            out["inputs"]["hibf"] = r
            os.remove(hp)
        elif "hibf" in want:
            out["skipped"].append("hibf")
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    out["wall_s"] = round(time.time() - t_start, 1)
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
