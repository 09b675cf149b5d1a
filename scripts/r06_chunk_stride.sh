#!/bin/bash
# Does the stride of the wave-private chunks of the match buffer (8192 records = 96 KiB) camp on HBM channels?  The same kernel with chunks of
# 8533 and 24571 records (strides that are no multiple of 4 KiB), match buffers grown / made at once, flat8g at --rel-cutoff 0.2.
mkdir -p gpurun_out
cp ganon_amd/csrc/libganon_hip.so /tmp/lib_base.so
for v in base 8533 24571 base 8533 24571; do
  if [ $v = base ]; then cp /tmp/lib_base.so ganon_amd/csrc/libganon_hip.so; else cp ganon_amd/csrc/libganon_hip_chunk$v.so ganon_amd/csrc/libganon_hip.so; fi
  for mpr in 2 130; do
    env GANON_BENCH_MATCHES_PER_READ=$mpr timeout 200 python bench.py --rel-cutoff 0.2 --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 200 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunk_max=$v matches_per_read=$mpr', 'Mreads/s', d['value'], 'count+select ms', d['config'].get('count_select_ms'), 'mismatching', d['config'].get('oracle_mismatching_reads'))"
  done
done | tee gpurun_out/r06_chunk_stride.txt
cp /tmp/lib_base.so ganon_amd/csrc/libganon_hip.so
