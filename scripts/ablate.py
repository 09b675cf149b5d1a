"""Perf ablation of the count kernel (GANON_HIP_ABLATE flags) on the flat8g shape with fewer reads."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import ganon_amd, bench_workload as bw
n_reads = int(os.environ.get("N_READS", 4_000_000))
rows = int(os.environ.get("ROWS", 1 << 24))
wl = bw.make_flat_workload("ablate", 4096, rows, 4, n_reads, seed=42)
if os.environ.get("MAP") == "split2":   # every target owns two contiguous technical bins -> generic (CSR) kernel
    b2t = (np.arange(wl.bins) // 2).astype(np.uint32)
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs, b2t, wl.bins // 2)
else:
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs)
bw.plant_genomes(flt, wl)
st = ganon_amd.HipStream(flt, n_reads, wl.bases.size, n_reads * 2)
st.upload(wl.bases, wl.off, None)
for flags in [int(x) for x in os.environ.get("FLAGS", "0,1,2,6").split(",")]:
    os.environ["GANON_HIP_ABLATE"] = str(flags)
    ms = []
    for i in range(4):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync()
        t = st.timings(); ms.append(t["ms_count"])
    gbs = t["algo_bytes"] / (min(ms[1:]) * 1e-3) / 1e9
    print(f"flags {flags}: count ms {['%.2f' % m for m in ms]}  -> {gbs:.0f} GB/s algorithmic; minimiser {t['ms_minimiser']:.2f} ms; total {t['ms_total']:.2f} ms", flush=True)
