#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite with durations, the default bench line, the two-rank shared-GPU dry run
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=80 > gpurun_out/r06_pytest1.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest1.log
python bench.py > gpurun_out/r06_bench1.out 2> gpurun_out/r06_bench1.err
echo "bench rc $?" >> gpurun_out/r06_bench1.err
cp bench_detail.json gpurun_out/r06_bench1_detail.json
GANON_BENCH_ALLOW_SHARED_GPU=1 GANON_BENCH_EXTRAS=tiny,slice_tiny GANON_BENCH_E2E_READS=8000000 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06_bench_2rank_dry.out 2> gpurun_out/r06_bench_2rank_dry.err
echo "bench2 rc $?" >> gpurun_out/r06_bench_2rank_dry.err
tail -c 3000 gpurun_out/r06_bench_2rank_dry.out
