"""Cross-check of the three clocks behind the bench numbers on the flat8g workload, without torch in the process:
CPU wall-clock around classify+sync, the library's hipEvents, and (when run under `rocprofv3 --kernel-trace`) the
tracer's kernel durations.  Used for profiles/r01_clock_check.txt."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import ganon_amd, bench_workload as bw
n = 10_000_000
wl = bw.make_flat_workload("flat8g", 4096, 1 << 24, 4, n, seed=42)
flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs)
bw.plant_genomes(flt, wl)
st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
for i in range(2):
    st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync()
for i in range(4):
    a = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync()
    b = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    t = st.timings()
    print(f"step {i}: wall {(b-a)/1e6:.3f} ms  [{a} .. {b}]  events: min {t['ms_minimiser']:.3f} count {t['ms_count']:.3f} total {t['ms_total']:.3f}", flush=True)
