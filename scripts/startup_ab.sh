#!/bin/bash
# Start-up A/B of the product binary on 16 M reads + a 1 GiB filter in /dev/shm: where does the time before the first batch go, and what
# do the pinning mode, the fast exit and the gap between two runs change?  bash scripts/startup_ab.sh  (through gpurun) -> gpurun_out/r05_startup_ab.txt
set -u
ROOT=$PWD; D=/dev/shm; OUT=$ROOT/gpurun_out/r05_startup_ab.txt; mkdir -p $ROOT/gpurun_out; : > $OUT
[ -f $D/keep.fq ] || E2E_KEEP=keep python scripts/e2e_cli.py 16000000 21 $D > /dev/null 2>&1
run() { # label, env...
  local label=$1; shift
  local t0=$(date +%s.%N)
  env "$@" $ROOT/ganon_amd/host/ganon-classify --ibf $D/keep.ibf --single-reads $D/keep.fq -o $D/ab_out --output-all --rel-cutoff 0.75 --rel-filter 0.1 --fpr-query 1e-5 --verbose 2> $D/ab_err > /dev/null
  local t1=$(date +%s.%N)
  echo "== $label: wall $(echo "$t1 - $t0" | bc) s" >> $OUT
  grep -E "into HBM|runtime up|main\(\) returns|batch pool|warm-up batch \(all|classifying\+printing|loading filter" $D/ab_err | cut -c1-260 >> $OUT
}
for i in 1 2 3; do run "product, back to back #$i"; done
sleep 3; run "product after 3 s of rest"
for i in 1 2; do run "hipHostMalloc pinning #$i" GANON_HIP_ABLATE=pinned_malloc; done
sleep 3; run "hipHostMalloc pinning after rest" GANON_HIP_ABLATE=pinned_malloc
for i in 1 2; do run "full teardown #$i" GANON_HOST_FULL_TEARDOWN=1; done
for i in 1 2; do run "full teardown + hipHostMalloc #$i" GANON_HOST_FULL_TEARDOWN=1 GANON_HIP_ABLATE=pinned_malloc; done
rm -f $D/keep.* $D/ab_out.* $D/ab_err
cat $OUT
