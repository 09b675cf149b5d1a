#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r06_pytest5.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest5.log
tail -4 gpurun_out/r06_pytest5.log
timeout 300 python bench.py --no-extra --no-e2e --no-cpu-baseline --steps 5 > gpurun_out/r06_bench5.out 2> gpurun_out/r06_bench5.err
cp bench_detail.json gpurun_out/r06_bench5_detail.json
python - <<'P'
import json
d=json.load(open("gpurun_out/r06_bench5_detail.json"))
print(d["value"], d["config"]["kernel_ms"], json.dumps(d.get("variants"))[:1500])
P
timeout 400 python bench_e2e.py --only paired --runs 10 --budget 300 > gpurun_out/r06_e2e_paired_10runs.json 2> gpurun_out/r06_e2e_paired_10runs.err
echo "paired rc $?"
