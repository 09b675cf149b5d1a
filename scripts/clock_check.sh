#!/bin/bash
# bash scripts/clock_check.sh  (through gpurun) -> gpurun_out/clock_check.txt
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/clock_check.txt; mkdir -p $ROOT/gpurun_out; : > $OUT
cd /tmp
for ee in 1 0; do
  if [ $ee = 0 ]; then export GANON_HIP_ABLATE=early_exit; else unset GANON_HIP_ABLATE; fi
  echo "### early exit $ee, untraced" >> $OUT
  python $ROOT/scripts/clock_check.py 2>&1 | grep step >> $OUT
  echo "### early exit $ee, under rocprofv3 --kernel-trace --stats" >> $OUT
  rm -rf /tmp/cc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cc -- python $ROOT/scripts/clock_check.py 2>&1 | grep step >> $OUT
  grep count_fast /tmp/cc/*/*kernel_stats.csv | cut -d, -f1-4,6,7 >> $OUT
done
cat $OUT
