"""End-to-end timing of the product binary on self-hosted files: a device-generated filter is saved as a ganon-build .ibf
(ganon_amd.ibf_file), synthetic reads are written as FASTQ, then `ganon-classify` runs start to finish (streaming filter
load -> FASTQ parse -> GPU -> .all/.rep written) on one device and with two workers (`--device 0,0`; on a multi-GPU node
use `--device all`).  Prints one JSON object.   usage: e2e_cli.py [n_reads=16000000] [rows_log2=21] [dir=/dev/shm]"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import bench_workload as bw  # noqa: E402
import ganon_amd  # noqa: E402
from ganon_amd import ibf_file  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
rows = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 21)
d = sys.argv[3] if len(sys.argv) > 3 else "/dev/shm"
bins = 4096
# classification thresholds: E2E_ARGS="" runs the binary's own defaults (--rel-cutoff 0.2 --rel-filter 0 --fpr-query 1); what `ganon classify`
# passes is E2E_ARGS="--rel-cutoff 0.75 --rel-filter 0.1 --fpr-query 1e-5"; the demanding mix for filter_matches is
# E2E_ARGS="--rel-cutoff 0.2 --rel-filter 0.1 --fpr-query 1e-5"
EXTRA = os.environ.get("E2E_ARGS", "--rel-cutoff 0.75").split()
out = {"reads": n, "filter_gib": rows * 512 / 2**30, "args": " ".join(EXTRA)}

wl = bw.make_device_flat_workload("e2e", bins, rows, 4, n, seed=42)
if os.environ.get("E2E_NO_PLANT"):  # the filter holds none of the genomes the reads were cut from: chance matches only
    flt = ganon_amd.HipFilter.ibf(None, wl.bins, wl.rows, wl.hash_funs, None, None, device=0)
    flt.fill_random(wl.seed, 1, wl.word_lo, wl.row_words_total)
    out["planted"] = False
else:
    flt, _ = bw.device_filter(ganon_amd, wl)
ibf = os.path.join(d, "ganon_e2e.ibf")
# the filter's rows are Bernoulli(0.5) bits, i.e. every bin is a Bloom filter at its optimal load for h = 4: per-hash false
# positive rate 0.5^4 = 0.0625 (a database built with --max-fp 0.0625; ganon-build's default is 0.05).  The header
# declares the number of minimisers per bin that gives exactly that rate (n = S ln2 / h), so that classify's per-target
# fpr (GanonClassify.cpp:940-947,968-982) -- what --fpr-query works with -- describes the bits that are really there.
per_bin = int(rows * 0.6931471805599453 / 4)
cfg = dict(n_bins=bins, max_hashes_bin=per_bin, hash_functions=4, kmer_size=wl.k, window_size=wl.w, bin_size_bits=rows, max_fp=0.0625,
           true_max_fp=0.0625, true_avg_fp=0.0625)
ibf_file.save_ibf(ibf, flt, cfg, [(f"T{b}", per_bin) for b in range(bins)], [(b, f"T{b}") for b in range(bins)], bins, rows, 4)
flt.free()
IBFS = ibf
if os.environ.get("E2E_SHARED"):  # a second filter on the same level that shares half of its target names with the first
    wl2 = bw.make_device_flat_workload("e2e_b", bins, rows, 4, 1024, seed=43)
    flt2, _ = bw.device_filter(ganon_amd, wl2)
    ibf2 = os.path.join(d, "ganon_e2e_b.ibf")
    ibf_file.save_ibf(ibf2, flt2, cfg, [(f"T{b + bins // 2}", per_bin) for b in range(bins)], [(b, f"T{b + bins // 2}") for b in range(bins)],
                      bins, rows, 4)
    flt2.free()
    del wl2
    IBFS = ibf + "," + ibf2
    out["shared_targets"] = "two filters on one level, 2048 of 4096 target names in both"

# FASTQ with fixed-width ids, assembled as one byte matrix
L = wl.read_len
rec = np.empty((n, 2 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
rec[:, 0], rec[:, 1] = ord("@"), ord("r")
idx = np.arange(n, dtype=np.int64)
for p in range(9):
    rec[:, 2 + p] = (idx // 10 ** (8 - p)) % 10 + ord("0")
rec[:, 11] = ord("\n")
rec[:, 12:12 + L] = wl.bases.reshape(n, L)
rec[:, 12 + L:15 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 15 + L:15 + 2 * L] = ord("I")
if os.environ.get("E2E_GZ"):  # binned qualities as sequencers write them (a constant line would inflate unrealistically fast)
    qs = np.frombuffer(b"FFFFFFFFFFFF:FFF,FFFF#", dtype=np.uint8)
    rec[:, 15 + L:15 + 2 * L] = qs[np.random.default_rng(7).integers(0, len(qs), size=(n, L), dtype=np.uint8)]
rec[:, -1] = ord("\n")
fq = os.path.join(d, "ganon_e2e.fq")
rec.tofile(fq)
out["fastq_gib"] = round(os.path.getsize(fq) / 2**30, 2)
READS = ["--single-reads", fq]
if os.environ.get("E2E_GZ"):  # the same file as ordinary gzip (one member, level 6): what read sets usually look like
    t0 = time.time()
    subprocess.check_call(["gzip", "-6", "-k", "-f", fq])
    READS = ["--single-reads", fq + ".gz"]
    out["gzip"] = {"file_gib": round(os.path.getsize(fq + ".gz") / 2**30, 2), "compress_s": round(time.time() - t0, 1)}
if os.environ.get("E2E_FASTA"):  # the same reads as a FASTA file (sequential reader: no four-line structure to cut slabs at)
    fa = os.path.join(d, "ganon_e2e.fa")
    rec2 = rec[:, :12 + L + 1].copy()
    rec2[:, 0] = ord(">")
    rec2.tofile(fa)
    READS = ["--single-reads", fa]
    out["fasta"] = True
    del rec2
if os.environ.get("E2E_PAIRED"):  # a mate file: the same records, bases of the neighbouring read (content does not matter here)
    rec[:, 12:12 + L] = np.roll(wl.bases.reshape(n, L), 1, axis=0)
    fq2 = os.path.join(d, "ganon_e2e.2.fq")
    rec.tofile(fq2)
    READS = ["--paired-reads", fq + "," + fq2]
    out["paired"] = True
del rec

if os.environ.get("E2E_HIBF"):  # a two-level raptor-style HIBF (4096 user bins) written by the tests' fixture writer instead of the flat filter
    import ganon_fixtures as gf
    t0 = time.time()
    hb = gf.random_hibf(4096, 64, 2, seed=5, density=0.3, hash_funs=3, rows=(60000, 65537))
    os.remove(ibf)
    ibf = os.path.join(d, "ganon_e2e.hibf")
    gf.write_hibf(ibf, hb, [[f"/x/U{u}.minimiser"] for u in range(4096)], wl.k, wl.w, 0.05)
    EXTRA = EXTRA + ["--hibf"]
    out["hibf"] = {"user_bins": 4096, "file_mib": round(os.path.getsize(ibf) / 2**20, 1), "built_s": round(time.time() - t0, 1)}
# (the first reader of a file numpy has just written pays for it: 10 s of system time on 19 GiB in /dev/shm; a reader, then the runs)
for path in READS[1].split(","):
    with open(path, "rb") as fh:
        while fh.read(1 << 26):
            pass
exe = os.path.join(ROOT, "ganon_amd", "host", "ganon-classify")
runs = [("one_worker", "0", None), ("two_workers_one_gpu", "0,0", None), ("three_workers_one_gpu", "0,0,0", None), ("default_no_device_flag", None, None)]
if os.environ.get("E2E_GZ"):  # ... and with the parallel inflate switched off (one zlib stream, as before) / other thread counts
    runs += [("gz_sequential_inflate", "0,0", "seq"), ("gz_inflate_8", "0,0", "i8"), ("gz_inflate_12", "0,0", "i12"), ("gz_inflate_24", "0,0", "i24")]
if not (os.environ.get("E2E_GZ") or os.environ.get("E2E_FASTA") or os.environ.get("E2E_PAIRED")):
    # the same file parsed on the host (slab parsers) instead of tokenised on the device, on 8 and on 12 parser threads; one context per worker
    runs += [("host_slab_parser", None, "h8"), ("host_slab_parser_12_threads", None, "h12"), ("one_lane_per_worker", None, "l1")]
if os.environ.get("E2E_SWEEP"):  # parser threads x device workers, to see which stage limits the pipeline on this host
    runs += [(f"sweep_parse{pt}_workers{len(dev.split(','))}", dev, pt) for pt in (4, 6, 8, 10, 12) for dev in ("0", "0,0", "0,0,0")]
for label, dev, parse_threads in runs:
    prefix = os.path.join(d, "ganon_e2e_out_" + label)
    t0 = time.time()
    env = dict(os.environ, GANON_HOST_TIMING="1")
    if parse_threads == "seq":
        env["GANON_HOST_NO_PGZIP"] = "1"
    elif isinstance(parse_threads, str) and parse_threads.startswith("h"):
        env["GANON_HOST_DEVICE_FASTQ"] = "0"
        env["GANON_HOST_PARSE_THREADS"] = parse_threads[1:]
    elif parse_threads == "l1":
        env["GANON_HOST_LANES"] = "1"
    elif isinstance(parse_threads, str) and parse_threads.startswith("i"):
        env["GANON_HOST_INFLATE_THREADS"] = parse_threads[1:]
    elif parse_threads:
        env["GANON_HOST_PARSE_THREADS"] = str(parse_threads)
    p = subprocess.run([exe, "--ibf", IBFS if not os.environ.get("E2E_HIBF") else ibf] + READS + ["-o", prefix, "--output-all", "--verbose"] + (["--device", dev] if dev else []) + EXTRA,
                       capture_output=True, text=True, env=env, timeout=900)
    r = {"rc": p.returncode, "wall_s": round(time.time() - t0, 2)}
    for key, pat in (("load_s", r"loading filter\(s\)\s+elapsed \(s\): ([0-9.eE+-]+)"),
                     ("classify_print_s", r"classifying\+printing elapsed \(s\): ([0-9.eE+-]+)"),
                     ("total_s", r"total\s+elapsed \(s\): ([0-9.eE+-]+)")):
        m = re.search(pat, p.stderr)
        if m:
            r[key] = float(m.group(1))
    m = re.search(r"\[host timing\] (.*)", p.stderr)
    if m:
        r["host_timing"] = m.group(1)
    bt = re.findall(r"\[backend timing\] (.*)", p.stderr)
    if bt:
        r["backend_timing"] = bt
    m = re.search(r"\[prefilter\] (.*)", p.stderr)
    if m:
        r["prefilter"] = m.group(1)
    m = re.search(r"\[host stalls\] (.*)", p.stderr)
    if m:
        r["host_stalls"] = m.group(1)
    for key in ("host cpu", "host input", "host pipeline", "pinned pool"):
        m = re.search(r"\[" + key + r"\] (.*)", p.stderr)
        if m:
            r[key.replace(" ", "_")] = m.group(1)
    if "classify_print_s" in r:
        r["mreads_per_s_classify_print"] = round(n / r["classify_print_s"] / 1e6, 2)
    if p.returncode == 0 and parse_threads is None:
        r["all_lines"] = sum(1 for _ in open(prefix + ".all"))
        r["rep_tail"] = open(prefix + ".rep").read().splitlines()[-2:]
    elif p.returncode == 0:
        os.remove(prefix + ".all")
    else:
        r["stderr"] = p.stderr[-400:]
    out[label] = r
if all(out[x].get("rc") == 0 for x in ("one_worker", "two_workers_one_gpu", "three_workers_one_gpu", "default_no_device_flag")):
    a = os.path.join(d, "ganon_e2e_out_one_worker")
    out["outputs_identical"] = all(open(a + e, "rb").read() == open(os.path.join(d, "ganon_e2e_out_" + x) + e, "rb").read()
                                   for e in (".all", ".rep") for x in ("two_workers_one_gpu", "three_workers_one_gpu", "default_no_device_flag"))
if os.environ.get("E2E_KEEP"):  # leave filter and reads behind for follow-up runs (A/B scripts)
    os.rename(ibf, os.path.join(d, os.environ["E2E_KEEP"] + ".ibf"))
    os.rename(fq, os.path.join(d, os.environ["E2E_KEEP"] + ".fq"))
    if os.environ.get("E2E_SHARED"):
        os.rename(ibf2, os.path.join(d, os.environ["E2E_KEEP"] + "_b.ibf"))
    if os.environ.get("E2E_PAIRED"):
        os.rename(fq2, os.path.join(d, "ganon_keep2.fq"))
for f in os.listdir(d):
    if f.startswith("ganon_e2e"):
        os.remove(os.path.join(d, f))
print(json.dumps(out))
