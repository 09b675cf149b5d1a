"""Quick timing of the other BASELINE.json shapes (wide rows, HIBF) -- exploration, not the driver bench."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import ganon_amd, bench_workload as bw
which = os.environ.get("SHAPES", "flat32k,hibf64k").split(",")
if "flat32k" in which:
    n = 2_000_000
    wl = bw.make_flat_workload("flat32k", 32768, 1 << 21, 4, n, seed=42)   # 2^21 rows x 4 KiB = 8 GiB
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs)
    bw.plant_genomes(flt, wl)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
    for i in range(3):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
    print(f"flat32k (W=512, 4 KiB rows, 8 GiB): count {t['ms_count']:.2f} ms, minimiser {t['ms_minimiser']:.2f} ms, "
          f"{t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s algorithmic, {n/t['ms_total']/1e3:.1f} Mreads/s, matches {t['n_matches']}", flush=True)
    st.destroy(); flt.free(); del wl
if "hibf64k" in which:
    n = 2_000_000
    t0 = time.time()
    wl = bw.make_hibf_workload(ganon_amd, "hibf64k", 65536, 256, 1 << 20, 1 << 20, 3, n)
    print(f"hibf built in {time.time()-t0:.1f}s, {wl.filter_bytes/2**30:.2f} GiB", flush=True)
    flt = ganon_amd.HipFilter.hibf(wl.ibfs, wl.next_ibf_id, wl.bin_to_user, wl.n_user_bins)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
    for i in range(3):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
    print(f"hibf64k: count {t['ms_count']:.2f} ms, minimiser {t['ms_minimiser']:.2f} ms, {t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s "
          f"algorithmic ({t['algo_bytes']/n:.0f} B/read), {n/t['ms_total']/1e3:.1f} Mreads/s, matches {t['n_matches']}", flush=True)
if "split" in which:
    # split bins: every target owns 2 technical bins -> generic (LDS-counter) kernel
    n = 2_000_000
    wl = bw.make_flat_workload("flat1g", 4096, 1 << 21, 4, n, seed=42)
    b2t = (np.arange(wl.bins, dtype=np.uint32) // 2)
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs, b2t.tolist(), wl.bins // 2)
    bw.plant_genomes(flt, wl)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
    for i in range(3):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
    print(f"split (4096 bins -> 2048 targets, 1 GiB): count {t['ms_count']:.2f} ms, {t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s "
          f"algorithmic, {n/t['ms_total']/1e3:.1f} Mreads/s, matches {t['n_matches']}", flush=True)
    st.destroy(); flt.free(); del wl
if "split32k" in which:
    # 32768 bins, every target owns 2 technical bins -> generic kernel with 4 column slices per read
    n = 2_000_000
    wl = bw.make_flat_workload("flat32k", 32768, 1 << 21, 4, n, seed=42)
    b2t = (np.arange(wl.bins, dtype=np.uint32) // 2)
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs, b2t.tolist(), wl.bins // 2)
    bw.plant_genomes(flt, wl)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
    for i in range(3):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
    print(f"split32k (32768 bins -> 16384 targets, 8 GiB): count {t['ms_count']:.2f} ms, {t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s "
          f"algorithmic, {n/t['ms_total']/1e3:.1f} Mreads/s, matches {t['n_matches']}", flush=True)
    st.destroy(); flt.free(); del wl
if "split40k" in which:
    # 40960 bins (640 words per row: 5 column slices of a wave, 8 waves per read), 2 bins per target: the shape of a
    # species-level RefSeq index
    n = 1_000_000
    wl = bw.make_flat_workload("flat40k", 40960, 1 << 20, 4, n, seed=42)   # 2^20 rows x 5 KiB = 5 GiB
    b2t = (np.arange(wl.bins, dtype=np.uint32) // 2)
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs, b2t.tolist(), wl.bins // 2)
    bw.plant_genomes(flt, wl)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
    for i in range(3):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
    print(f"split40k (40960 bins -> 20480 targets, 5 GiB): count {t['ms_count']:.2f} ms, {t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s "
          f"algorithmic, {n/t['ms_total']/1e3:.1f} Mreads/s, matches {t['n_matches']}", flush=True)
    st.destroy(); flt.free()
    flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs)
    bw.plant_genomes(flt, wl)
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2); st.upload(wl.bases, wl.off, None)
    for i in range(3):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
    print(f"flat40k identity (fast kernel): count {t['ms_count']:.2f} ms, {t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s "
          f"algorithmic, fetched {t['fetched_bytes']/t['algo_bytes']:.3f}, {n/t['ms_total']/1e3:.1f} Mreads/s, matches {t['n_matches']}", flush=True)
    st.destroy(); flt.free(); del wl
