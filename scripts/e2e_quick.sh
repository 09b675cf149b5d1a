#!/bin/bash
# quick look: bash scripts/e2e_quick.sh [reads] -- one kept data set, the variants named in $VARIANTS (label:ENV=..,ENV=..:args)
N=${1:-64000000}
ROOT=$PWD
D=/dev/shm
[ -f $D/keep.fq ] || E2E_KEEP=keep python scripts/e2e_cli.py $N 21 $D > /dev/null 2>&1
[ -f $D/nop.fq ] || { E2E_NO_PLANT=1 E2E_KEEP=nop python scripts/e2e_cli.py 1000 21 $D > /dev/null 2>&1; rm -f $D/nop.fq; }
EXE=$ROOT/ganon_amd/host/ganon-classify
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$( env GANON_HOST_TIMING=1 "${envs[@]}" timeout 90 $EXE --ibf ${IBF:-$D/keep.ibf} --single-reads $D/keep.fq -o $D/ab_out ${NOOUT:+--quiet} $( [ -z "${NOOUT:-}" ] && echo --output-all ) --rel-cutoff 0.75 --verbose "$@" 2>&1 )
  t=$(echo "$out" | grep -o "classifying+printing elapsed (s): [0-9.e+-]*" | grep -o "[0-9.e+-]*$")
  echo "$label: classify+print $t s = $(python -c "print(round($N/$t/1e6,1))") Mreads/s"
  echo "$out" | grep -E "host stalls|backend timing|host cpu|host timing|host input|host input|pinned pool|host cpu|ERROR|rror" | sed 's/^/      /' | cut -c1-420
}
for rep in 1 2; do
run "text, one copy per piece" --
run "text, 2 copies per piece" GANON_HIP_SPLIT_UPLOAD=2 --
run "text, 4 copies per piece" GANON_HIP_SPLIT_UPLOAD=4 --
done
run "one worker, one copy" -- --device 0
run "one worker, 4 copies" GANON_HIP_SPLIT_UPLOAD=4 -- --device 0
run "two workers, 4 copies" GANON_HIP_SPLIT_UPLOAD=4 -- --device 0,0
