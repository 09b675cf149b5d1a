#!/bin/bash
# Round profile of one bench workload on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r02 flat8g
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>_<workload>/ ; scripts/summarize_profiles.py copies the summaries
# that DESIGN.md quotes into profiles/.  Counter passes are separate runs (never combined with tracing domains other than
# the kernel trace), in the order: timing (bench line) -> kernel trace/stats -> FETCH_SIZE -> FETCH_SIZE without the early
# exit (calibration of the counter on this access pattern) -> WRITE_SIZE -> SQ counters.
set -u
TAG=${1:-r03}
WL=${2:-flat8g}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-extra --no-variants --no-every-row --check 0"
cd /tmp
python $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --no-extra > $OUT/bench.json 2> $OUT/bench.log
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
if [ "$WL" = flat8g ] || [ "$WL" = flat128g ]; then
  GANON_HIP_ABLATE=early_exit timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_noee -- $BENCH > $OUT/pmc_fetch_noee.log 2>&1
fi
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 420 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $ROOT
python scripts/summarize_profiles.py $TAG $WL
