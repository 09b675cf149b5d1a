#!/bin/bash
# Device-tokenised FASTQ (default) against the host's slab parser (GANON_HOST_DEVICE_FASTQ=0) on ONE kept data set, with the CPU
# seconds per thread group.   bash scripts/e2e_ab_raw.sh [reads=64000000]
N=${1:-64000000}
ROOT=$PWD
D=/dev/shm
E2E_KEEP=keep python scripts/e2e_cli.py $N 21 $D > /dev/null 2>&1
EXE=$ROOT/ganon_amd/host/ganon-classify
run() { # label, env..., -- args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$( env GANON_HOST_TIMING=1 "${envs[@]}" timeout 120 $EXE --ibf $D/keep.ibf --single-reads $D/keep.fq -o $D/ab_out_$TAG --output-all --rel-cutoff 0.75 --verbose "$@" 2>&1 )
  t=$(echo "$out" | grep -o "classifying+printing elapsed (s): [0-9.e+-]*" | grep -o "[0-9.e+-]*$")
  echo "$label: classify+print $t s = $(python -c "print(round($N/$t/1e6,1))") Mreads/s"
  echo "$out" | grep -E "host stalls|backend timing|host cpu|host timing|host input|ERROR|rror" | sed 's/^/      /' | cut -c1-400
}
TAG=a run "host slab parser" GANON_HOST_DEVICE_FASTQ=0 --
TAG=b run "device tokeniser" --
cmp $D/ab_out_a.all $D/ab_out_b.all && cmp $D/ab_out_a.rep $D/ab_out_b.rep && echo "outputs identical"
for rep in 1 2; do
TAG=b run "device tokeniser" --
TAG=b run "device tokeniser, 4 workers" -- --device 0,0,0,0
TAG=b run "device tokeniser, 2 workers" -- --device 0,0
TAG=b run "device tokeniser, 4 readers" GANON_HOST_PARSE_THREADS=4 --
TAG=b run "device tokeniser, 12 readers" GANON_HOST_PARSE_THREADS=12 --
TAG=b run "device tokeniser, 96 MiB pieces" GANON_HOST_SLAB_BYTES=100663296 --
TAG=b run "device tokeniser, 24 MiB pieces" GANON_HOST_SLAB_BYTES=25165824 --
TAG=b run "device tokeniser, 4 workers 5 post" GANON_HOST_POST_THREADS=5 -- --device 0,0,0,0
TAG=a run "host slab parser" GANON_HOST_DEVICE_FASTQ=0 --
done
rm -f $D/keep.ibf $D/keep.fq $D/ab_out_*
