"""The split-bin kernel at the low-cutoff thresholds, alone (for rocprofv3 passes): split32k workload, --rel-cutoff 0.2 with the
filter_matches pre-pass set (--rel-filter 0.1 --fpr-query 1e-5), N steps.  usage: split_lowcut.py [steps=5] [reads=2000000] [cutoff=0.2]
MIX=1: targets of 1 .. 4 consecutive bins (seeded draw from 1,1,1,2,2,3,4) instead of two bins each; MIX=2: one in a hundred of 5 .. 200 bins."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import bench_workload as bw  # noqa: E402
import ganon_amd  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
cutoff = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
bins, rows, h, bpt = 32768, 1 << 21, 4, 2
wl = bw.make_device_flat_workload("split32k", bins, rows, h, n, False, rel_cutoff=cutoff, seed=42)
if os.environ.get("MIX"):
    sizes = np.random.default_rng(7).choice(np.array([1, 1, 1, 2, 2, 3, 4]), size=bins)
    if os.environ["MIX"] == "2":  # ... and one target in a hundred of 5 .. 200 bins
        r7 = np.random.default_rng(8)
        sizes = np.where(r7.random(bins) < 0.01, r7.choice(np.array([5, 9, 40, 70, 200]), size=bins), sizes)
    sizes = sizes[: int(np.searchsorted(np.cumsum(sizes), bins, side="right"))]
    b2t = np.repeat(np.arange(sizes.size, dtype=np.uint32), sizes)
    b2t = np.concatenate([b2t, np.full(bins - b2t.size, sizes.size, dtype=np.uint32)])  # the remainder: one last target
    lens = np.bincount(b2t).astype(np.float64)
    nt = int(lens.size)
else:
    b2t, nt, lens = (np.arange(bins, dtype=np.uint32) // bpt).astype(np.uint32), bins // bpt, np.full(bins // bpt, float(bpt))
flt, _ = bw.device_filter(ganon_amd, wl, 0, b2t, nt)
st = ganon_amd.HipStream(flt, n, wl.bases.size, max_matches=n * 2)
st.upload(wl.bases, wl.off, None)
if not os.environ.get("NO_PREPASS"):
    st.set_postfilter(0.1, 1e-5, 1.0 - (1.0 - 0.5 ** h) ** lens)
ms = []
for _ in range(steps + 1):
    st.classify(wl.k, wl.w, cutoff)
    st.sync()
    ms.append(st.timings()["ms_count"])
mo, m = st.fetch()[2:4]
import zlib
print("count+select ms per step:", [round(x, 2) for x in ms[1:]], "matches", st.timings()["n_matches"],
      "crc", zlib.crc32(np.ascontiguousarray(m).tobytes()), zlib.crc32(np.ascontiguousarray(mo).tobytes()))
