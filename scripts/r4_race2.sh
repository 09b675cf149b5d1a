#!/bin/bash
mkdir -p gpurun_out/race
OLD=$PWD/scripts/_ab/libganon_hip_r3memset.so
for i in 1 2 3 4; do GANON_HIP_LIB=$OLD timeout 600 python scripts/inproc_stress.py 120; done > gpurun_out/race/inproc_old.jsonl 2> gpurun_out/race/inproc_old.err
for i in 1 2; do timeout 600 python scripts/inproc_stress.py 120; done > gpurun_out/race/inproc_new.jsonl 2> gpurun_out/race/inproc_new.err
cat gpurun_out/race/inproc_old.jsonl gpurun_out/race/inproc_new.jsonl | cut -c1-1200
