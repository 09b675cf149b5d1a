#!/bin/bash
# round 4: the 16-byte-lane fast kernel at three waves a SIMD (-DGN_FAST_WPE_LW2=3, scripts/_ab/libganon_hip_wpe3.so) against the default two
for lib in "" "GANON_HIP_LIB=$PWD/scripts/_ab/libganon_hip_wpe3.so"; do
  for wl in flat32k flat128g; do
    env $lib python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; v=d.get('variants') or {}
print('$wl', '${lib:0:14}', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], d['config'].get('oracle_spot_check'), {k:(x.get('mreads_per_s'), x.get('frac')) for k,x in v.items()})"
  done
done
