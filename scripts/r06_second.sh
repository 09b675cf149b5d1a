#!/bin/bash
# round 6, second GPU call: segmented results (parity + numbers), a kernel trace at the binary's --rel-cutoff 0.2, the host ceiling
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_segmented.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_gather.py tests/test_gpu_fullsize.py tests/test_partition_cli.py tests/test_cli_kat.py -m gpu -x -q > gpurun_out/r06_pytest2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest2.log
tail -5 gpurun_out/r06_pytest2.log
python bench.py --no-extra --no-e2e --no-cpu-baseline --steps 5 > gpurun_out/r06_bench2.out 2> gpurun_out/r06_bench2.err
cp bench_detail.json gpurun_out/r06_bench2_detail.json
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06_cut02 -- python $R/bench.py --rel-cutoff 0.2 --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 0 --steps 5 > $R/gpurun_out/r06_cut02_trace.log 2>&1)
find gpurun_out/prof_r06_cut02 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_flat8g_cutoff0.2_kernel_stats.csv
python scripts/host_ceiling.py --workers 8 --post-threads 16 --runs 3 > gpurun_out/r06_host_ceiling.json 2> gpurun_out/r06_host_ceiling.err
echo "ceiling rc $?"
