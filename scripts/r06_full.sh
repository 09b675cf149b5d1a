#!/bin/bash
# the driver's round-end sequence: the whole -m gpu suite, smoke, the default bench line
mkdir -p gpurun_out
TAG=${1:-run2}
timeout 1400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06_pytest_full_$TAG.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest_full_$TAG.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06_pytest_full_$TAG.log | tail -40
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r06_bench_default_$TAG.out 2> gpurun_out/r06_bench_default_$TAG.err
echo "bench rc $?"
cp bench_detail.json gpurun_out/r06_bench_default_${TAG}_detail.json
tail -1 gpurun_out/r06_bench_default_$TAG.out
