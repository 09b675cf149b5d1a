"""Throughput of the product's ganon-build on synthetic genomes: N FASTA files of L random bases each (70-column lines) are
written to a tmpfs, then `ganon-build` runs start to finish (parse -> device minimisers + sort/unique -> sizing -> device
insert -> .ibf written).  Prints one JSON object.   usage: bench_build.py [n_files=512] [len=4000000] [threads=16] [dir=/dev/shm]"""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
d = os.path.join(sys.argv[4] if len(sys.argv) > 4 else "/dev/shm", "ganon_build_bench")
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
cols = 70
rows = (L + cols - 1) // cols
with open(os.path.join(d, "in.tsv"), "w") as tsv:
    for i in range(n_files):
        body = np.full((rows, cols + 1), ord("\n"), dtype=np.uint8)
        body[:, :cols] = lut[rng.integers(0, 4, size=(rows, cols), dtype=np.uint8)]
        f = os.path.join(d, f"g{i}.fna")
        with open(f, "wb") as o:
            o.write(f">genome{i} synthetic\n".encode())
            o.write(body.tobytes()[: L + L // cols + 1])
        tsv.write(f"{f}\tT{i}\n")
out = {"files": n_files, "bases_per_file": L, "total_gbp": round(n_files * L / 1e9, 3), "threads": threads}
exe = os.path.join(ROOT, "ganon_amd", "host", "ganon-build")
t0 = time.time()
p = subprocess.run([exe, "-i", os.path.join(d, "in.tsv"), "-o", os.path.join(d, "db.ibf"), "-t", str(threads), "--verbose", "-p", "0.05"],
                   capture_output=True, text=True)
out["rc"], out["wall_s"] = p.returncode, round(time.time() - t0, 2)
for key, pat in (("count_hashes_s", r"Count/save hashes start:.*\n.*\n\s*elapsed \(s\): ([0-9.eE+-]+)"),
                 ("sizing_s", r"Estimate params   start:.*\n.*\n\s*elapsed \(s\): ([0-9.eE+-]+)"),
                 ("fill_s", r"Building filter   start:.*\n.*\n\s*elapsed \(s\): ([0-9.eE+-]+)"),
                 ("write_s", r"Saving filer      start:.*\n.*\n\s*elapsed \(s\): ([0-9.eE+-]+)"),
                 ("total_s", r"ganon-build       start:.*\n.*\n\s*elapsed \(s\): ([0-9.eE+-]+)")):
    m = re.search(pat, p.stderr)
    if m:
        out[key] = float(m.group(1))
m = re.search(r"ganon-build processed .*", p.stderr)
out["summary"] = m.group(0) if m else p.stderr[-300:]
for key in ("n_bins", "max_hashes_bin", "bin_size_bits", "hash_functions"):
    m = re.search(key + r"\s+(\d+)", p.stderr)
    if m:
        out[key] = int(m.group(1))
if p.returncode == 0:
    out["ibf_gib"] = round(os.path.getsize(os.path.join(d, "db.ibf")) / 2**30, 3)
    out["mbp_per_s"] = round(n_files * L / 1e6 / out.get("total_s", out["wall_s"]), 1)
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
print(json.dumps(out))
