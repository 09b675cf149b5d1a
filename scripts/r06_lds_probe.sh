#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_devgzip.py -m gpu -x -q 2>&1 | grep -E "passed|failed" 
(cd scripts && timeout 300 python inflate_turns_probe.py --turns 1 --reps 5)
(cd scripts && timeout 300 python inflate_probe.py --reads 1000000 --tile 8 --reps 3 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['best']; print('probe wall GB/s', round(b['text_GBps_wall'],1), 'decode ms', round(b['ms_decode'],1), 'equal', r['bytes_equal'])")
timeout 400 python bench_e2e.py --only gz --runs 5 --budget 200 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['inputs']['gz']; print('e2e gz', v['rate'], [r['s'] for r in v['per_run']])"
