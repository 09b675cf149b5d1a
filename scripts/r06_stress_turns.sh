#!/bin/bash
# the inflater-turn tests five times over (a race shows up as a sporadic byte difference)
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_devgzip.py tests/test_gpu_inflate.py -m gpu -x -q -k "turns" 2>&1 | grep -E "passed|failed" | sed "s/^/round $i: /"
done | tee gpurun_out/r06_stress_turns.txt
