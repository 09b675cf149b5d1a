#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_devgzip.py tests/test_gpu_inflate.py -m gpu -x -q > gpurun_out/r06_pytest9.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest9.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06_pytest9.log | tail -40
