#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_segmented.py tests/test_gpu_parity.py tests/test_gpu_gather.py tests/test_partition_cli.py tests/test_host_tunables.py tests/test_cli_kat.py -m gpu -x -q > gpurun_out/r06_pytest3.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest3.log
tail -4 gpurun_out/r06_pytest3.log
python scripts/host_ceiling.py --workers 8 --post-threads 16 --runs 2 --reads 48000000 --only fastq,gz --skip-real > gpurun_out/r06_host_ceiling_48m.json 2> gpurun_out/r06_host_ceiling_48m.err
echo "ceiling rc $?"
nproc; cat /sys/fs/cgroup/cpu.max; free -g | head -2
