"""f-4 measurement: the EM of gn_reassign_* on a seeded table (device time, GB/s against the algorithmic bytes of an iteration:
8 per read + 4 per entry of a read with several), and the ganon-reassign binary on a generated .all/.rep (where the time goes).
usage: python scripts/bench_reassign.py [n_reads] [n_targets]  -> one JSON line"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ganon_amd  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_targets = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
rng = np.random.default_rng(11)
deg = np.where(rng.random(n_reads) < 0.4, 1, rng.integers(2, 9, size=n_reads)).astype(np.int64)
off = np.zeros(n_reads + 1, dtype=np.uint64)
np.cumsum(deg, out=off[1:])
w = rng.random(n_targets) ** 3 + 1e-3
tgt = rng.choice(n_targets, size=int(off[-1]), p=w / w.sum()).astype(np.uint32)
out = dict(n_reads=n_reads, n_entries=int(off[-1]), n_targets=n_targets)
for label, mi in (("em_10", 10), ("em_unbounded", 0)):
    t0 = time.time()
    g = ganon_amd.HipReassign(off, tgt, n_targets)
    t1 = time.time()
    diffs, counts, unique, prob, choice = g.run(mi, 0.0)
    t2 = time.time()
    info = g.info()
    g.free()
    passes = len(diffs) + 1  # + the final choice
    out[label] = dict(iterations=len(diffs), upload_s=round(t1 - t0, 3), run_and_fetch_s=round(t2 - t1, 3), device_ms=round(info["ms"], 3),
                      bytes_per_iteration=info["bytes_per_iteration"], gbps=round(info["bytes_per_iteration"] * passes / (info["ms"] * 1e6), 1),
                      frac_of_8tbs=round(info["bytes_per_iteration"] * passes / (info["ms"] * 1e6) / 8000, 3), wave_reads=info["wave_reads"])
# the binary on text (2 M reads: Python-side text generation is the slow part of this script)
m = min(n_reads, 2_000_000)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    names = [f"T{t}.1" for t in range(n_targets)]
    with open(os.path.join(d, "x.all"), "w") as f:
        o = off[: m + 1].astype(np.int64)
        lines = []
        for r in range(m):
            for e in range(o[r], o[r + 1]):
                lines.append(f"read{r}\t{names[tgt[e]]}\t{10 + (e & 63)}\n")
            if len(lines) > 1 << 20:
                f.write("".join(lines))
                lines = []
        f.write("".join(lines))
    used = np.unique(tgt[: int(o[m])])
    with open(os.path.join(d, "x.rep"), "w") as f:
        for t in used:
            f.write(f"H1\t{names[t]}\t1\t0\t0\tspecies\tname {t}\n")
        f.write(f"#total_classified\t{m}\n#total_unclassified\t0\n")
    t0 = time.time()
    p = subprocess.run([os.path.join(ROOT, "ganon_amd", "host", "ganon-reassign"), "-i", os.path.join(d, "x"), "-o", os.path.join(d, "y"), "--verbose"],
                       capture_output=True, text=True)
    out["binary"] = dict(reads=m, lines=int(o[m]), all_bytes=os.path.getsize(os.path.join(d, "x.all")), wall_s=round(time.time() - t0, 3), rc=p.returncode,
                         verbose=[l for l in p.stderr.split("\n") if l.startswith("[reassign]")])
print(json.dumps(out))
