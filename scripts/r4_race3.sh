#!/bin/bash
# round 4, item 1: which HIP runtime is in the process decides whether hipMemset(null stream) is ordered before a non-blocking stream's copy
mkdir -p gpurun_out/race
OLD=$PWD/scripts/_ab/libganon_hip_r3memset.so
for t in "" 1; do for lib in $OLD ""; do
  GANON_HIP_LIB=$lib PROBE_IMPORT_TORCH=$t python scripts/memset_race_probe.py
done; done > gpurun_out/race/probe_runtime.jsonl 2> gpurun_out/race/probe_runtime.err
GANON_HIP_LIB=$OLD python -m pytest tests/test_upload_order.py -q -m gpu -k "zero_fill" > gpurun_out/race/pytest_zero_fill_old.log 2>&1
python -m pytest tests/test_upload_order.py -q -m gpu -k "zero_fill" > gpurun_out/race/pytest_zero_fill_new.log 2>&1
cat gpurun_out/race/probe_runtime.jsonl; tail -5 gpurun_out/race/pytest_zero_fill_old.log gpurun_out/race/pytest_zero_fill_new.log
