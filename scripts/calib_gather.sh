#!/bin/bash
# Calibration of short-row random gathers on the GPU box (through gpurun, from the repo root):
#   bash scripts/calib_gather.sh
# timing sweep -> gpurun_out/calib/timing.jsonl ; counter passes (each its own rocprofv3 run, kernel trace only beside
# them) -> gpurun_out/calib/pmc_<counters>_<table>_<row>/ ; scripts/calib_summary.py turns both into profiles/r03_calib_gather.json
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/calib
mkdir -p $OUT
export TMPDIR=/tmp
B=$ROOT/scripts/calib_gather
cd /tmp
: > $OUT/timing.jsonl
for T in 32 128 1024 8192; do
  for R in 32 64 128 512; do
    $B $T $R 800 8 8 >> $OUT/timing.jsonl
  done
done
# how many loads in flight / how many resident blocks a 32-byte-row gather needs
for U in 1 2 4 8 16; do for BPC in 2 4 8; do $B 8192 32 400 $U $BPC >> $OUT/timing.jsonl; done; done
for U in 2 4 8 16; do $B 32 32 400 $U 8 >> $OUT/timing.jsonl; done
for cfg in "32 32" "8192 32" "8192 64" "8192 128" "8192 512" "32 512"; do
  set -- $cfg
  for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
    tag=$(echo $C | tr ' ' '+')
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${tag}_$1_$2 -- $B $1 $2 800 8 8 > $OUT/pmc_${tag}_$1_$2.log 2>&1
  done
done
cd $ROOT
python scripts/calib_summary.py
