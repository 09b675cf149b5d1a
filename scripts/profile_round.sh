#!/bin/bash
# Round profile of the driver bench on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py copies the summaries that
# DESIGN.md quotes into profiles/.  Counter passes are separate runs (never combined with tracing).
set -u
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
cd /tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
GANON_HIP_NO_EARLY_EXIT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_noee -- $BENCH > $OUT/pmc_fetch_noee.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $ROOT
python scripts/summarize_profiles.py $TAG
