set -u
ROOT=$PWD
D=/dev/shm
export TMPDIR=/tmp
E2E_KEEP=keep python scripts/e2e_cli.py 32000000 21 $D > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_e2e -- $ROOT/ganon_amd/host/ganon-classify --ibf $D/keep.ibf --single-reads $D/keep.fq -o $D/ab_out --output-all --rel-cutoff 0.75 --verbose --device 0,0,0 2>&1 | grep -E "classifying|loading"
rm -f $D/keep.* $D/ab_out.*
