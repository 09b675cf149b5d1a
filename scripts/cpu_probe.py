import os, sys, time, json
sys.path[:0]=['/root/repo','/root/repo/tests']
print("cpu.max:", open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)
print("affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
for f in ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    if os.path.exists(f): print(f, open(f).read().strip())
import numpy as np, bench_workload as bw, bench_cpu, oracle, ganon_amd
wl = bw.make_device_flat_workload("p", 4096, 1<<21, 4, 400_000, seed=42)
flt,_ = bw.device_filter(ganon_amd, wl)
arr, keep, note = bench_cpu._host_filter_buffer(wl.rows*wl.bin_words); wl.filter_rows = arr.reshape(wl.rows, wl.bin_words); bw.download_filter(flt, wl); bench_cpu._reset_mempolicy()
bench_cpu._oracle_native()
ofl, ibf = bw.oracle_filter(wl)
ranks = oracle.to_ranks(wl.bases)
for th in (1, 4, 16, 64, 128, 256):
    n = min(wl.n_reads, 4000*th)
    t0=time.perf_counter(); oracle.baseline_classify(ofl, ranks[:int(wl.off[n])], wl.off[:n+1], wl.k, wl.w, th); dt=time.perf_counter()-t0
    print(f"threads {th}: {n} reads {dt:.2f}s -> {n/dt/1e6:.4f} Mreads/s, {dt*th/n*1e6:.1f} us/read/thread", flush=True)
