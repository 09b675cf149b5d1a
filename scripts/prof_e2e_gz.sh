#!/bin/bash
# rocprofv3 kernel trace + stats of ganon-classify on a .fq.gz file inflated on the device (the files of bench_e2e.py's gz leg)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D=/dev/shm/ganon_prof_gz
OUT=$ROOT/gpurun_out/prof_e2e_gz
rm -rf $D $OUT; mkdir -p $D $OUT
cd $ROOT
python bench_e2e.py --only gz --runs 1 --budget 300 --keep-gz $D > $OUT/bench_e2e.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
GANON_HOST_TIMING=1 GANON_HOST_FULL_TEARDOWN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o e2e_gz -- $ROOT/ganon_amd/host/ganon-classify \
  --ibf $D/e2e.ibf --single-reads $D/single.fq.gz -o $D/out --output-all --rel-cutoff 0.75 --verbose > $OUT/run.log 2>&1
grep -E "classifying|device inflate" $OUT/run.log | cut -c1-400
python - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
rm -rf $D
