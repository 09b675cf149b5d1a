"""The split-bin kernel at the low-cutoff thresholds, alone (for rocprofv3 passes): split32k workload, --rel-cutoff 0.2 with the
filter_matches pre-pass set (--rel-filter 0.1 --fpr-query 1e-5), N steps.  usage: split_lowcut.py [steps=5] [reads=2000000] [cutoff=0.2]"""
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
flt, _ = bw.device_filter(ganon_amd, wl, 0, (np.arange(bins, dtype=np.uint32) // bpt).astype(np.uint32), bins // bpt)
st = ganon_amd.HipStream(flt, n, wl.bases.size, max_matches=n * 2)
st.upload(wl.bases, wl.off, None)
if not os.environ.get("NO_PREPASS"):
    st.set_postfilter(0.1, 1e-5, np.full(bins // bpt, 1.0 - (1.0 - 0.5 ** h) ** bpt, dtype=np.float64))
ms = []
for _ in range(steps + 1):
    st.classify(wl.k, wl.w, cutoff)
    st.sync()
    ms.append(st.timings()["ms_count"])
print("count+select ms per step:", [round(x, 2) for x in ms[1:]], "matches", st.timings()["n_matches"])
