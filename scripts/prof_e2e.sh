# Device timeline of the product binary on 32 M reads (kernel + memory-copy trace; no counters): bash scripts/prof_e2e.sh [env...]
set -u
ROOT=$PWD
D=/dev/shm
export TMPDIR=/tmp
[ -f $D/keep.fq ] || E2E_KEEP=keep python scripts/e2e_cli.py 32000000 21 $D > /dev/null 2>&1
cd /tmp
for mode in raw; do
  rm -rf $ROOT/gpurun_out/prof_e2e_$mode
  if [ $mode = host ]; then export GANON_HOST_DEVICE_FASTQ=0; fi
  GANON_HOST_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/prof_e2e_$mode -- $ROOT/ganon_amd/host/ganon-classify --ibf $D/keep.ibf --single-reads $D/keep.fq -o $D/ab_out --output-all --rel-cutoff 0.75 --verbose --device 0,0,0 2>&1 | grep -E "classifying|loading|host cpu|backend timing|host stalls" | cut -c1-300
  python $ROOT/scripts/timeline_e2e.py $ROOT/gpurun_out/prof_e2e_$mode > $ROOT/gpurun_out/timeline_e2e_$mode.json
  cat $ROOT/gpurun_out/timeline_e2e_$mode.json
  find $ROOT/gpurun_out/prof_e2e_$mode -name "*.csv" -size +20M -delete
done
rm -f $D/keep.* $D/ab_out.*
