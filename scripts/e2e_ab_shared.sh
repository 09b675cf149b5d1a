#!/bin/bash
# two 1 GiB filters on one hierarchy level: one upload + one minimiser pass per batch (gn_stream_classify_shared) against one per filter
N=${1:-32000000}
ROOT=$PWD
D=/dev/shm
E2E_SHARED=1 E2E_KEEP=keep python scripts/e2e_cli.py $N 21 $D > /dev/null 2>&1
EXE=$ROOT/ganon_amd/host/ganon-classify
for mode in shared separate shared separate; do
  if [ $mode = separate ]; then export GANON_HOST_NO_SHARED_HASHES=1; else unset GANON_HOST_NO_SHARED_HASHES; fi
  out=$( $EXE --ibf $D/keep.ibf,$D/keep_b.ibf --single-reads $D/keep.fq -o $D/ab_$mode --output-all --rel-cutoff 0.75 --verbose 2>&1 )
  t=$(echo "$out" | grep -o "classifying+printing elapsed (s): [0-9.e+-]*" | grep -o "[0-9.e+-]*$")
  echo "$mode: classify+print $t s = $(python -c "print(round($N/$t/1e6,1))") Mreads/s"
done
cmp $D/ab_shared.all $D/ab_separate.all && cmp $D/ab_shared.rep $D/ab_separate.rep && echo "outputs identical"
rm -f $D/keep* $D/ab_*
