#!/bin/bash
# round 4: the measurements DESIGN quotes for reassign and the HIBF low-cutoff variants (run through gpurun from the repo root)
python scripts/bench_reassign.py > gpurun_out/r04_reassign_bench.jsonl 2>/dev/null
python scripts/bench_reassign.py 50000000 4000 >> gpurun_out/r04_reassign_bench.jsonl 2>/dev/null
cut -c1-250 gpurun_out/r04_reassign_bench.jsonl; grep -o "\[reassign\][^\"]*" gpurun_out/r04_reassign_bench.jsonl
for wl in hibf64k_skew hibf64k; do
  GANON_BENCH_HIBF_LOW_CUTOFF=1 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_${wl}_low_cutoff.json 2> gpurun_out/r04_${wl}_low.err
  tail -2 gpurun_out/r04_${wl}_low.err
  python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_${wl}_low_cutoff.json').read().strip().splitlines()[-1])
print('$wl', d['value'], d['config'].get('oracle_spot_check')); print(json.dumps(d.get('variants'))[:1200])"
done
