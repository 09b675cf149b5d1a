#!/bin/bash
# flat8g at --rel-cutoff 0.2 ran in two modes on one box (count+select 66.8 or 73.7 ms, same code): does it follow from how the 14 GB match
# buffers came to be (grown after an overflow, or made large at once)?
mkdir -p gpurun_out
for i in 1 2 3 4; do
 for mpr in 2 130; do
  env GANON_BENCH_MATCHES_PER_READ=$mpr timeout 200 python bench.py --rel-cutoff 0.2 --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('matches_per_read=$mpr', 'Mreads/s', d['value'], 'count+select ms', d['config'].get('count_select_ms'))"
 done
done | tee gpurun_out/r06_alloc_mode.txt
