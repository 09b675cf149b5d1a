#!/bin/bash
# A/B of the device inflater's scheduling switches on the end-to-end .fq.gz run (bench_e2e.py --only gz): decodes launched ahead or not,
# decode waves per CU; five runs each (the first run of a process pays for the page cache)
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/r05_e2e_gz_ab.txt}
: > $OUT
for sw in "" "inflate_ahead"; do
  echo "### GANON_HIP_ABLATE=$sw" >> $OUT
  GANON_HIP_ABLATE=$sw E2E_DIAG=1 timeout 600 python bench_e2e.py --only gz --runs 5 --budget 200 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read())
for k,v in r['inputs'].items():
    print(k, v.get('rate'), v.get('classify_print_s'), v.get('process_wall_s_median'), v.get('error'))
    for l in v.get('timing_lines', []):
        if 'device inflate' in l: print('   ', l[:300])
" >> $OUT
done
cat $OUT
