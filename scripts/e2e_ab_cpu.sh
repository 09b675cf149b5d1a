#!/bin/bash
# Where the host's CPU seconds go ($GANON_HOST_TIMING's [host cpu] line) on ONE kept data set, and what reading the page cache in place
# (GANON_HOST_READ=mmap) instead of copying it first (pread) changes.   bash scripts/e2e_ab_cpu.sh [reads=64000000]
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
  echo "$out" | grep -E "host stalls|backend timing|host cpu|host timing" | sed 's/^/      /' | cut -c1-400
}
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|NUMA node\(s\)"
for rep in 1 2; do
  run "default (pread)" -- 
  run "mmap" GANON_HOST_READ=mmap --
done
run "pread, 12 parsers" GANON_HOST_PARSE_THREADS=12 --
run "mmap, 12 parsers" GANON_HOST_READ=mmap GANON_HOST_PARSE_THREADS=12 --
run "mmap, 6 parsers" GANON_HOST_READ=mmap GANON_HOST_PARSE_THREADS=6 --
run "mmap, 4 workers" GANON_HOST_READ=mmap -- --device 0,0,0,0
rm -f $D/keep.ibf $D/keep.fq $D/ab_out.*
