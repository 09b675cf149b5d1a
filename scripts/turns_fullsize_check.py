#!/usr/bin/env python3
"""Inflater turns at the sizes the product uses (32 KiB chunks, 128 MiB steps, three decode sets ahead): an 8 M-read single-end .fq.gz and
4 M pairs from two .fq.gz files through ganon-classify with 1, 2 and 3 inflaters per file ($GANON_HOST_DEVICE_INFLATE_TURNS; one GPU) --
.all and .rep must not differ by a byte.  (The tests use tiny steps; this is the full-size twin.)  -> one JSON line"""
import hashlib, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_e2e as be  # noqa: E402
import bench_workload as bw  # noqa: E402
import ganon_amd  # noqa: E402
from ganon_amd import ibf_file  # noqa: E402

d = f"/dev/shm/ganon_turns_{os.getpid()}"
os.makedirs(d, exist_ok=True)
out = {"runs": []}
try:
    bins, rows, h, L = 4096, 1 << 21, 4, 150
    n = 8_000_000
    wl = bw.make_device_flat_workload("e2e", bins, rows, h, n, paired=False, seed=42)
    flt, _ = bw.device_filter(ganon_amd, wl)
    ibf = os.path.join(d, "t.ibf")
    per_bin = int(rows * 0.6931471805599453 / h)
    cfg = dict(n_bins=bins, max_hashes_bin=per_bin, hash_functions=h, kmer_size=wl.k, window_size=wl.w, bin_size_bits=rows, max_fp=0.0625, true_max_fp=0.0625, true_avg_fp=0.0625)
    ibf_file.save_ibf(ibf, flt, cfg, [(f"T{b}", per_bin) for b in range(bins)], [(b, f"T{b}") for b in range(bins)], bins, rows, h)
    flt.free()
    gz = os.path.join(d, "s.fq.gz")
    be.write_gzip_one_member(gz, be.fastq_matrix(wl.bases, n, L, quals=True))
    npair = 4_000_000
    wp = bw.make_device_flat_workload("e2e", bins, rows, h, npair, paired=True, seed=42, shard=1)
    z1, z2 = os.path.join(d, "p.1.fq.gz"), os.path.join(d, "p.2.fq.gz")
    be.write_gzip_one_member(z1, be.fastq_matrix(wp.bases[: npair * L], npair, L, quals=True))
    be.write_gzip_one_member(z2, be.fastq_matrix(wp.bases[npair * L:], npair, L, quals=True))
    ok = True
    for label, inp in (("single", ["--single-reads", gz]), ("paired", ["--paired-reads", z1 + "," + z2])):
        ref = None
        for turns in ("1", "2", "3"):
            pre = os.path.join(d, f"o_{label}_{turns}")
            env = dict(os.environ, GANON_HOST_DEVICE_INFLATE_TURNS=turns, GANON_HOST_TIMING="1")
            t0 = time.time()
            p = subprocess.run([be.EXE, "--ibf", ibf, "-o", pre, "--output-all", "--verbose"] + be.THRESHOLDS + inp, capture_output=True, text=True, env=env, timeout=600)
            digest = {}
            for ext in (".all", ".rep"):
                hsh = hashlib.sha256()
                with open(pre + ext, "rb") as f:
                    for blk in iter(lambda: f.read(1 << 24), b""):
                        hsh.update(blk)
                digest[ext] = hsh.hexdigest()[:16]
                os.remove(pre + ext)
            said = [ln for ln in p.stderr.splitlines() if "device inflate:" in ln]
            rec = {"input": label, "turns": turns, "rc": p.returncode, "wall_s": round(time.time() - t0, 2), "digest": digest,
                   "inflaters_line": (said[-1][said[-1].index("device inflate:"):][:90] if said else "")}
            ref = ref or digest
            rec["same_as_one_inflater"] = digest == ref
            ok = ok and p.returncode == 0 and digest == ref and bool(said)
            out["runs"].append(rec)
    out["ok"] = ok
finally:
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
print(json.dumps(out), flush=True)
sys.exit(0 if out.get("ok") else 1)
