#!/bin/bash
# A/B runs of the product binary over ONE kept data set (scripts/e2e_cli.py with E2E_KEEP=keep): which stage limits the pipeline.
#   bash scripts/e2e_ab.sh [reads=64000000]
N=${1:-64000000}
ROOT=$PWD
D=/dev/shm
E2E_KEEP=keep python scripts/e2e_cli.py $N 21 $D > /dev/null 2>&1
EXE=$ROOT/ganon_amd/host/ganon-classify
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$( env GANON_HOST_TIMING=1 "${envs[@]}" timeout 120 $EXE --ibf $D/keep.ibf --single-reads $D/keep.fq -o $D/ab_out --output-all --rel-cutoff 0.75 --verbose "$@" 2>&1 )
  t=$(echo "$out" | grep -o "classifying+printing elapsed (s): [0-9.e+-]*" | grep -o "[0-9.e+-]*$")
  echo "$label: classify+print $t s = $(python -c "print(round($N/$t/1e6,1))") Mreads/s"
  echo "$out" | grep -E "host stalls|backend timing" | sed 's/^/      /' | cut -c1-330
}
for mode in spin block yield; do
  run "2 workers, sync $mode" GANON_HIP_SYNC=$mode -- --device 0,0
  run "3 workers, sync $mode" GANON_HIP_SYNC=$mode -- --device 0,0,0
  run "4 workers, sync $mode" GANON_HIP_SYNC=$mode -- --device 0,0,0,0
done
run "3 workers, sync block, 10 parsers" GANON_HIP_SYNC=block GANON_HOST_PARSE_THREADS=10 -- --device 0,0,0
run "4 workers, sync block, 10 parsers, 4 post" GANON_HIP_SYNC=block GANON_HOST_PARSE_THREADS=10 GANON_HOST_POST_THREADS=4 -- --device 0,0,0,0
rm -f $D/keep.ibf $D/keep.fq $D/ab_out.*
