"""Filter load rate from disk into HBM: a device-generated flat IBF is saved as a
ganon-build .ibf (ganon_amd.ibf_file.save_ibf) on a tmpfs, then loaded back by
  (a) the ganon-classify binary  -- streaming loader of ganon_amd/host/filter_io.cpp, its own "loading filter(s)" clock,
  (b) ganon_amd.ibf_file.load_ibf -- whole filter and one of eight column slices (what one rank of config 5 loads).
usage: bench_load.py [rows_log2=25] [bins=32768] [dir=/dev/shm]"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import ganon_amd  # noqa: E402
from ganon_amd import ibf_file  # noqa: E402

rows = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
d = sys.argv[3] if len(sys.argv) > 3 else "/dev/shm"
path = os.path.join(d, "ganon_bench_load.ibf")
W = (bins + 63) >> 6
gib = rows * W * 8 / 2**30
out = {"filter_gib": round(gib, 2), "bins": bins, "rows": rows, "dir": d}

flt = ganon_amd.HipFilter.ibf(None, bins, rows, 4)
flt.fill_random(7, 1)
cfg = dict(n_bins=bins, max_hashes_bin=1000, hash_functions=4, kmer_size=19, window_size=31, bin_size_bits=rows, max_fp=0.05,
           true_max_fp=0.05, true_avg_fp=0.05)
t0 = time.time()
ibf_file.save_ibf(path, flt, cfg, [(f"t{b}", 1000) for b in range(bins)], [(b, f"t{b}") for b in range(bins)], bins, rows, 4)
out["save_s"] = round(time.time() - t0, 2)
probe = np.array([0, 1, rows // 2, rows - 1], dtype=np.uint64)
want = flt.download_row_list(probe, W)
flt.free()

t0 = time.time()
f2, m = ibf_file.load_ibf(path)
out["python_load_s"] = round(time.time() - t0, 2)
out["python_load_gbs"] = round(gib * 2**30 / 1e9 / (time.time() - t0), 2)
assert np.array_equal(f2.download_row_list(probe, W), want)
f2.free()
t0 = time.time()
f3, _ = ibf_file.load_ibf(path, word_lo=5 * W // 8, word_hi=6 * W // 8, bin2target=np.arange(bins // 8, dtype=np.uint32), n_targets=bins // 8)
out["python_slice_load_s"] = round(time.time() - t0, 2)   # reads the whole file, keeps 1/8 of every row
assert np.array_equal(f3.download_row_list(probe, W // 8), want[:, 5 * W // 8: 6 * W // 8])
f3.free()

fq = os.path.join(d, "ganon_bench_load.fq")
open(fq, "w").write("@r0\n" + "ACGT" * 40 + "\n+\n" + "I" * 160 + "\n")
p = subprocess.run([os.path.join(ROOT, "ganon_amd", "host", "ganon-classify"), "--ibf", path, "--single-reads", fq, "-o",
                    os.path.join(d, "ganon_bench_load_out"), "--verbose"], capture_output=True, text=True)
mt = re.search(r"loading filter\(s\)\s+elapsed \(s\): ([0-9.eE+-]+)", p.stderr)
out["binary_rc"] = p.returncode
if mt:
    out["binary_load_s"] = float(mt.group(1))
    out["binary_load_gbs"] = round(gib * 2**30 / 1e9 / float(mt.group(1)), 2)
else:
    out["binary_stderr"] = p.stderr[-500:]
for f in (path, fq):
    os.remove(f)
print(json.dumps(out))
