"""Idle holes of a device timeline (rocprofv3 kernel + memory-copy trace): intervals in which neither a kernel nor a copy ran, with what ran
before and after.   timeline_holes.py <dir> [min_ms=0.5]"""
import csv, glob, sys
d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
k, m = [], []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True): k += list(csv.DictReader(open(f)))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True): m += list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:30], r.get('Stream_Id', '')) for r in k]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'][-14:], r.get('Stream_Id', '')) for r in m]
ev.sort()
mark = [e for e in ev if e[2].startswith('void gn_ibf_count_fast')]
a, b = mark[10][0], mark[-10][1]
ev = [e for e in ev if e[0] >= a and e[1] <= b]
t0 = a
end = ev[0][1]; last = ev[0]
holes = []
for e in ev[1:]:
    if e[0] > end:
        holes.append((e[0] - end, end, last, e))
    if e[1] > end:
        end = e[1]; last = e
tot = sum(h[0] for h in holes)
big = [h for h in holes if h[0] > min_ms * 1e6]
print({"window_ms": (b - a) / 1e6, "idle_ms": tot / 1e6, "holes": len(holes), "holes_over_min": len(big), "idle_in_those_ms": sum(h[0] for h in big) / 1e6})
import collections
hist = collections.Counter(min(int(h[0] / 1e5), 30) for h in holes)
print("hole length histogram (0.1 ms bins: count):", sorted(hist.items()))
for h in big[:25]:
    print(f"  hole {h[0]/1e6:6.3f} ms at {(h[1]-t0)/1e6:8.3f}: after {h[2][2]:30s} s={h[2][3]:3s} before {h[3][2]:30s} s={h[3][3]}")
