"""Steady-state window of a device timeline (rocprofv3 kernel + memory-copy trace of the product binary): busy time of the copy engine and
of the compute units per batch, gaps between uploads, and a printout of a few milliseconds.   timeline_window.py <dir> [print_ms]"""
import csv, glob, statistics, sys
d = sys.argv[1]
show = float(sys.argv[2]) if len(sys.argv) > 2 else 0
k, m = [], []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True): k += list(csv.DictReader(open(f)))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True): m += list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:34], r.get('Queue_Id', ''), r.get('Stream_Id', '')) for r in k]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M', r['Direction'][-14:], '', r.get('Stream_Id', '')) for r in m]
ev.sort()
mark = sorted(e for e in ev if e[3].startswith('gn_fq_count') or e[3].startswith('void gn_ibf_count_fast'))
per = 2 if any(e[3].startswith('gn_fq_count') for e in mark) else 1
a, b, t0 = mark[8 * per][0], mark[-8 * per][1], mark[0][0]
def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)
win = [e for e in ev if e[0] >= a and e[1] <= b]
nb = len([e for e in win if e[3].startswith('void gn_ibf_count_fast')])
H = sorted((e[0], e[1]) for e in win if e[2] == 'M' and e[3].startswith('HOST_TO') and e[1] - e[0] > 300000)
gaps = [(H[i + 1][0] - H[i][1]) / 1e6 for i in range(len(H) - 1)]
print({"batches_total": len(mark) // per, "window_ms": round((b - a) / 1e6, 2), "batches": nb, "ms_per_batch": round((b - a) / 1e6 / max(nb, 1), 3),
       "h2d_busy_ms": round(union(H) / 1e6, 2), "h2d_copy_ms_mean": round(statistics.mean((e - s) / 1e6 for s, e in H), 3),
       "h2d_gap_ms_median": round(statistics.median(gaps), 3), "h2d_gaps_over_0.3ms": sum(g > 0.3 for g in gaps),
       "d2h_busy_ms": round(union([(e[0], e[1]) for e in win if e[2] == 'M' and e[3].startswith('DEVICE_TO_H')]) / 1e6, 2),
       "kernels_busy_ms": round(union([(e[0], e[1]) for e in win if e[2] == 'K']) / 1e6, 2),
       "kernels_sum_ms": round(sum(e[1] - e[0] for e in win if e[2] == 'K') / 1e6, 2),
       "anything_busy_ms": round(union([(e[0], e[1]) for e in win]) / 1e6, 2),
       "first_upload_ms": 0.0, "span_ms": round((mark[-1][1] - t0) / 1e6, 2)})
if show:
    mid = (a + b) // 2
    for e in ev:
        if mid <= e[0] < mid + show * 1e6 and e[1] - e[0] > 25000:
            print(f"{(e[0]-t0)/1e6:9.3f} {(e[1]-e[0])/1e6:7.3f} ms {e[2]} {e[3]:36s} q={e[4]} s={e[5]}")
