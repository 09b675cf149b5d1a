#!/bin/bash
# where do the 8 ms of the fast kernel's low-cutoff epilogue go?  the same batch at --rel-cutoff 0.2: product / matches listed but not stored / nothing emitted
mkdir -p gpurun_out
for probe in 0 32 0 32; do
  if [ $probe = 0 ]; then E=""; else E="GANON_HIP_ABLATE=emit_probe=$probe"; fi
  env $E timeout 200 python bench.py --rel-cutoff 0.2 --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emit_probe=$probe', d['value'], d['ms_per_step'], d['config'].get('count_select_ms'))"
done > gpurun_out/r06_emit_probe.txt
cat gpurun_out/r06_emit_probe.txt
