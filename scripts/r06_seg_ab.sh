#!/bin/bash
# one box, flat8g at --rel-cutoff 0.2: the product (segments, LDS list) against the per-batch copy of round 5 (switch seg_result), alternating
mkdir -p gpurun_out
for i in 1 2 3; do
 for sw in "" "seg_result"; do
  env GANON_HIP_ABLATE=$sw timeout 200 python bench.py --rel-cutoff 0.2 --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('switches=[$sw]', 'Mreads/s', d['value'], 'ms/step', d['ms_per_step'], 'count+select ms', d['config'].get('count_select_ms'))"
 done
done | tee gpurun_out/r06_seg_ab.txt
