#!/bin/bash
# round 4, item 1: the memset race with round 3's library and with this one
mkdir -p gpurun_out/race
OLD=$PWD/scripts/_ab/libganon_hip_r3memset.so
for i in 1 2 3; do GANON_HIP_LIB=$OLD python scripts/memset_race_probe.py; done > gpurun_out/race/probe_old.jsonl 2>gpurun_out/race/probe_old.err
for i in 1 2 3; do python scripts/memset_race_probe.py; done > gpurun_out/race/probe_new.jsonl 2>gpurun_out/race/probe_new.err
GANON_HIP_LIB=$OLD GANON_TEST_FRESH_RUNS=120 python -m pytest tests/test_upload_order.py -q -m gpu -k fresh > gpurun_out/race/fresh_old.log 2>&1
python -m pytest tests/test_upload_order.py tests/test_build_gpu.py -q -m gpu > gpurun_out/race/new.log 2>&1
tail -3 gpurun_out/race/*.jsonl gpurun_out/race/*.log
