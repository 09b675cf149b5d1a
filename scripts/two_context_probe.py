"""Does a second batch context on the same device hide the minimiser kernel (VALU-bound, 2.2 ms per 10 M reads) behind the count
kernels (memory-bound)?  ganon-classify gives every worker two contexts; bench.py's steps run on one.  This probe runs a workload's
resident batch (a) K times on one context, (b) K times split over two contexts driven by two host threads, and prints both rates.
Both contexts hold the same reads; the checksums of their results must agree.
  python scripts/two_context_probe.py [workload] [steps]"""
import json, os, sys, threading, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
import ganon_amd, bench_workload as bw, bench

name = sys.argv[1] if len(sys.argv) > 1 else "flat8g"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
spec = bench.WORKLOADS[name]
n = spec["reads"]
if spec["kind"] == "hibf":
    fill = ganon_amd.FILL_3_OF_16 if spec.get("fill") == "3/16" else 0
    if spec.get("skew"):
        wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, name, spec["user_bins"], spec["h"], n, seed=42, rows_scale=spec.get("rows_scale", 1.0), fill=fill)
    else:
        wl, flt = bw.make_hibf_device_workload(ganon_amd, name, spec["user_bins"], spec["tmax"], spec.get("rows_top", spec["rows"]), spec["rows"], spec["h"], n,
                                               seed=42, fill=fill or 1)
    off2 = None
else:
    wl = bw.make_device_flat_workload(name, spec["bins"], spec["rows"], spec["h"], n, spec["paired"], seed=42)
    flt, _ = bw.device_filter(ganon_amd, wl, 0)
    off2 = wl.off2
ctx = []
for _ in range(2):
    st = ganon_amd.HipStream(flt, n, wl.bases.size, n * 2)
    st.upload(wl.bases, wl.off, off2)
    st.sync()
    ctx.append(st)


def run(st, k):
    for _ in range(k):
        st.classify(wl.k, wl.w, wl.rel_cutoff)
        st.sync()


for st in ctx:
    run(st, 2)
t0 = time.perf_counter()
run(ctx[0], steps)
one = time.perf_counter() - t0
tm1 = ctx[0].timings()
th = [threading.Thread(target=run, args=(st, steps // 2)) for st in ctx]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
two = time.perf_counter() - t0
tm2 = ctx[0].timings()
ck = [bw.checksum_matches(st.fetch()[3]) for st in ctx]
k2 = 2 * (steps // 2)
print(json.dumps({"workload": name, "reads": n, "steps": steps,
                  "one_context": {"ms_per_step": round(one / steps * 1e3, 3), "mreads_s": round(n * steps / one / 1e6, 1),
                                  "ms_minimiser": round(tm1["ms_minimiser"], 3), "ms_count": round(tm1["ms_count"], 3)},
                  "two_contexts": {"ms_per_step": round(two / k2 * 1e3, 3), "mreads_s": round(n * k2 / two / 1e6, 1),
                                   "ms_minimiser": round(tm2["ms_minimiser"], 3), "ms_count": round(tm2["ms_count"], 3)},
                  "same_result": ck[0] == ck[1]}), flush=True)
