"""End-to-end timing of the ganon-classify binary (FASTQ parse -> GPU -> .all/.rep written) on a synthetic 1 GiB filter,
and device-only timing of paired reads on the same filter.  Exploration script (numbers quoted in DESIGN.md)."""
import os, struct, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganon_amd, bench_workload as bw

n = int(os.environ.get("N_READS", 2_000_000))
wl = bw.make_flat_workload("e2e", 4096, 1 << 21, 4, n, seed=42)
flt = ganon_amd.HipFilter.ibf(wl.filter_rows.reshape(-1), wl.bins, wl.rows, wl.hash_funs)
bw.plant_genomes(flt, wl)
bw.download_filter(flt, wl)

# ---- paired reads, device only
half = n // 2
off1 = wl.off[: half + 1]
off2 = wl.off[half: 2 * half + 1]
st = ganon_amd.HipStream(flt, half, wl.bases.size, half * 2)
st.upload(wl.bases, off1, off2)
for i in range(3):
    st.classify(wl.k, wl.w, wl.rel_cutoff); st.sync(); t = st.timings()
print(f"paired 2x150 ({half} pairs): count {t['ms_count']:.2f} ms, minimiser {t['ms_minimiser']:.2f} ms, "
      f"{t['algo_bytes']/t['ms_count']/1e6:.0f} GB/s algorithmic, {half/t['ms_total']/1e3:.1f} Mpairs/s, "
      f"{t['n_hashes']/half:.1f} minimisers/pair", flush=True)
st.destroy(); flt.free()

# ---- files
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
def wstr(s):
    b = s.encode(); return struct.pack("<Q", len(b)) + b
t0 = time.time()
with open(f"{d}/db.ibf", "wb") as f:
    f.write(struct.pack("<3i", 2, 1, 1))
    f.write(struct.pack("<QQBBHQddd", wl.bins, 500, wl.hash_funs, wl.k, wl.w, wl.rows, 0.05, 0.05, 0.05))
    f.write(struct.pack("<Q", wl.bins))
    for i in range(wl.bins):
        f.write(wstr(f"T{i}") + struct.pack("<Q", 400))
    f.write(struct.pack("<Q", wl.bins))
    for i in range(wl.bins):
        f.write(struct.pack("<Q", i) + wstr(f"T{i}"))
    W = wl.bin_words
    f.write(struct.pack("<6Q", wl.bins, W * 64, wl.rows, 64 - int(wl.rows).bit_length(), W, wl.hash_funs))
    f.write(struct.pack("<BfQ", 1, 1.5, W * 64 * wl.rows))
    f.write(wl.filter_rows.tobytes())
reads = wl.bases.reshape(n, wl.read_len)
with open(f"{d}/reads.fq", "wb") as f:
    qual = b"I" * wl.read_len
    chunk = []
    for i in range(n):
        chunk.append(b"@r%d\n%s\n+\n%s\n" % (i, reads[i].tobytes(), qual))
        if len(chunk) == 100000:
            f.write(b"".join(chunk)); chunk = []
    f.write(b"".join(chunk))
print(f"files written in {time.time()-t0:.1f}s: db.ibf {os.path.getsize(d+'/db.ibf')/2**30:.2f} GiB, reads.fq {os.path.getsize(d+'/reads.fq')/2**20:.0f} MiB", flush=True)
t0 = time.time()
p = subprocess.run([os.path.join(ROOT, "ganon_amd/host/ganon-classify"), "--ibf", f"{d}/db.ibf", "--single-reads", f"{d}/reads.fq",
                    "-o", f"{d}/out", "--output-all", "--rel-cutoff", "0.75", "--verbose"], capture_output=True, text=True,
                   env=dict(os.environ, GANON_HOST_TIMING="1"))
dt = time.time() - t0
print("rc", p.returncode, f"wall {dt:.2f}s")
print("\n".join(l for l in p.stderr.splitlines() if "elapsed" in l or "host timing" in l or "processed" in l or "classified" in l or "ERROR" in l))
print("all lines:", sum(1 for _ in open(f"{d}/out.all")), open(f"{d}/out.rep").read().splitlines()[-2:])

# ---- gzip input: one file, and the same file as both mates (inflate-bound reader)
if os.environ.get("E2E_GZ"):
    subprocess.run(["gzip", "-1", "-k", "-f", f"{d}/reads.fq"], check=True)
    import zlib
    t0 = time.time()
    with open(f"{d}/reads.fq", "rb") as fi, open(f"{d}/readsb.fq.gz", "wb") as fo:   # blocked gzip (bgzip / Illumina style)
        while True:
            chunk = fi.read(65280)
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            fo.write(struct.pack("<4BI2BH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6) + struct.pack("<2BHH", 66, 67, 2, 12 + 6 + len(comp) + 8 - 1))
            fo.write(comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
            if not chunk:
                break
    print(f"bgzf written in {time.time()-t0:.1f}s")
    for label, args in (("single gz", ["--single-reads", f"{d}/reads.fq.gz"]),
                        ("paired gz", ["--paired-reads", f"{d}/reads.fq.gz,{d}/reads.fq.gz"]),
                        ("single bgzf", ["--single-reads", f"{d}/readsb.fq.gz"]),
                        ("paired bgzf", ["--paired-reads", f"{d}/readsb.fq.gz,{d}/readsb.fq.gz"])):
        t0 = time.time()
        p = subprocess.run([os.path.join(ROOT, "ganon_amd/host/ganon-classify"), "--ibf", f"{d}/db.ibf", *args, "-o", f"{d}/outz",
                            "--output-all", "--rel-cutoff", "0.75", "--verbose"], capture_output=True, text=True,
                           env=dict(os.environ, GANON_HOST_TIMING="1"))
        print(label, "rc", p.returncode, f"wall {time.time()-t0:.2f}s")
        print("\n".join(l for l in p.stderr.splitlines() if "host timing" in l or "processed" in l or "ERROR" in l))
