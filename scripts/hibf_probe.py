"""Where does a level of the skewed HIBF spend its time?  One workload (bench.py hibf64k_skew by default), the count path under a
list of switch settings (gn_ablate), per-level hipEvent times and line rates for each; checksums must agree.
  python scripts/hibf_probe.py [workload] [reads] -- prints one JSON line per variant"""
import json, os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import ganon_amd, bench_workload as bw, bench

name = sys.argv[1] if len(sys.argv) > 1 else "hibf64k_skew"
spec = bench.WORKLOADS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else spec["reads"]
fill = ganon_amd.FILL_3_OF_16 if spec.get("fill") == "3/16" else 0
if spec.get("skew"):
    wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, name, spec["user_bins"], spec["h"], n, seed=42, rows_scale=spec.get("rows_scale", 1.0), fill=fill)
else:
    wl, flt = bw.make_hibf_device_workload(ganon_amd, name, spec["user_bins"], spec["tmax"], spec.get("rows_top", spec["rows"]), spec["rows"], spec["h"], n, seed=42, fill=fill or 1)
st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2)
st.upload(wl.bases, wl.off, None)
ref = None
variants = [x for x in os.environ.get("VARIANTS", "|hibf_persistent|hibf_bpc=2|hibf_bpc=4|hibf_bpc=6|hibf_bpc=8").split("|")]
for v in variants:
    ganon_amd.set_ablation(v)
    ms, lv = [], None
    for i in range(4):
        st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync()
        t = st.timings()
        if i:
            ms.append((t["ms_count"], t["ms_total"]))
    lv = st.hibf_levels()
    ck = bw.checksum_matches(st.fetch()[3])
    ref = ref or ck
    print(json.dumps({"variant": v or "(product)", "count_ms": round(float(np.mean([a for a, _ in ms])), 3), "total_ms": round(float(np.mean([b for _, b in ms])), 3),
                      "mreads_s": round(n / float(np.mean([b for _, b in ms])) / 1e3, 1), "same_result": ck == ref,
                      "levels": [{"ms": round(l["ms"], 3), "line_gbs": round(l["line_bytes"] / max(l["ms"], 1e-6) / 1e6, 1),
                                  "algo_gb": round(l["algo_bytes"] / 1e9, 2), "line_gb": round(l["line_bytes"] / 1e9, 2)} for l in lv]}), flush=True)
ganon_amd.set_ablation("")
