# device timelines of (a) the product binary on FASTQ text and (b) scripts/lane_probe (the same device calls without the rest of the host
# pipeline), side by side: bash scripts/prof_compare.sh
set -u
ROOT=$PWD
D=/dev/shm
export TMPDIR=/tmp
[ -f $D/keep.fq ] || E2E_KEEP=keep python scripts/e2e_cli.py 32000000 21 $D > /dev/null 2>&1
[ -f $D/nop.ibf ] || { E2E_NO_PLANT=1 E2E_KEEP=nop python scripts/e2e_cli.py 1000 21 $D > /dev/null 2>&1; rm -f $D/nop.fq; }
cd /tmp
rm -rf $ROOT/gpurun_out/prof_cmp_*
GANON_HOST_DEVICE_FASTQ=1 GANON_HOST_NO_PREFILTER=1 GANON_HOST_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/prof_cmp_binary -- $ROOT/ganon_amd/host/ganon-classify --ibf $D/nop.ibf --single-reads $D/keep.fq -o $D/ab_out --output-all --rel-cutoff 0.75 --verbose 2>&1 | grep -E "classifying" | cut -c1-200
python $ROOT/scripts/timeline_window.py $ROOT/gpurun_out/prof_cmp_binary 6 > $ROOT/gpurun_out/timeline_cmp_binary.txt
PROBE_ONE=3x2 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/gpurun_out/prof_cmp_probe -- $ROOT/scripts/lane_probe 300 2>&1 | grep threads
python $ROOT/scripts/timeline_window.py $ROOT/gpurun_out/prof_cmp_probe 6 > $ROOT/gpurun_out/timeline_cmp_probe.txt
head -1 $ROOT/gpurun_out/timeline_cmp_binary.txt $ROOT/gpurun_out/timeline_cmp_probe.txt
find $ROOT/gpurun_out/prof_cmp_binary $ROOT/gpurun_out/prof_cmp_probe -name "*.csv" -size +30M -delete
rm -f $D/keep.* $D/nop.* $D/ab_out.*
