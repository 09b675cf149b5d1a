"""Device timeline of one run of the product binary (rocprofv3 --kernel-trace --memory-copy-trace, csv): how long the copy engines and
the compute units were busy, at what rate the copies ran, and how much of the wall time nothing ran.   timeline_e2e.py <dir>"""
import csv
import glob
import json
import sys

d = sys.argv[1]


def rows(pat):
    out = []
    for f in glob.glob(d + "/**/*" + pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


k = rows("kernel_trace.csv")
m = rows("memory_copy_trace.csv")
kiv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k]
res = {}
if kiv:
    t0, t1 = min(s for s, _ in kiv), max(e for _, e in kiv)
    res["span_ms"] = (t1 - t0) / 1e6
    res["kernels_busy_ms"] = union(kiv) / 1e6
    res["kernels_sum_ms"] = sum(e - s for s, e in kiv) / 1e6
    res["n_kernels"] = len(kiv)
    by = {}
    for r in k:
        n = r["Kernel_Name"].split("(")[0][:60]
        by.setdefault(n, [0, 0])
        by[n][0] += 1
        by[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res["kernels"] = {n: {"calls": c, "ms": round(t / 1e6, 2)} for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:14]}
for direction in sorted({r["Direction"] for r in m}):
    sel = [r for r in m if r["Direction"] == direction]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel]
    nbytes = sum(int(r.get("Size", 0) or 0) for r in sel)
    big = [r for r in sel if int(r.get("Size", 0) or 0) >= (8 << 20)]
    res["copy_" + direction] = {"n": len(sel), "GB": round(nbytes / 1e9, 3), "busy_ms": round(union(iv) / 1e6, 2), "sum_ms": round(sum(e - s for s, e in iv) / 1e6, 2),
                                "GBps_while_busy": round(nbytes / max(union(iv), 1), 2),
                                "large_copies_GBps_each": round(sum(int(r["Size"]) for r in big) / max(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big), 1), 2) if big else None}
if kiv and m:
    alliv = kiv + [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in m]
    res["anything_busy_ms"] = union(alliv) / 1e6
print(json.dumps(res, indent=1))
