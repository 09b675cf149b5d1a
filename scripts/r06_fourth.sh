#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_segmented.py tests/test_partition_cli.py tests/test_upload_order.py -m gpu -x -q > gpurun_out/r06_pytest4.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest4.log
tail -4 gpurun_out/r06_pytest4.log
bash scripts/prof_cutoff.sh > gpurun_out/r06_prof_cutoff.log 2>&1
cat gpurun_out/r06_cutoff_counters.txt
python bench_e2e.py --only paired --runs 10 --budget 300 > gpurun_out/r06_e2e_paired_10runs.json 2> gpurun_out/r06_e2e_paired_10runs.err
echo "paired rc $?"
