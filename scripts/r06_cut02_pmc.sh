#!/bin/bash
# HBM bytes of the fast count kernel at the binary's --rel-cutoff 0.2 (108 matches a read written): FETCH_SIZE and WRITE_SIZE, one pass each
set -u
R=$PWD
OUT=$R/gpurun_out/prof_r06_cut02_pmc
mkdir -p $OUT
export TMPDIR=/tmp
B="python $R/bench.py --rel-cutoff 0.2 --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 0 --steps 5 --warmup 1"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
cd $R
python - <<'P' | tee gpurun_out/r06_flat8g_cutoff0.2_pmc.txt
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "prof_r06_cut02_pmc")
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    vals = []
    for p in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p, newline="")):
            if "gn_ibf_count_fast_kernel" in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    vals = vals[1:] if len(vals) > 1 else vals          # (the first launch is the one whose match buffer overflowed and was run again)
    kib = sum(vals) / max(1, len(vals))
    print(name, "launches", len(vals), "KiB per launch", round(kib), "-> bytes x1024:", round(kib * 1024 / 1e9, 2), "GB; x1024x2 (gfx950 read tally):", round(kib * 2048 / 1e9, 2), "GB")
P
