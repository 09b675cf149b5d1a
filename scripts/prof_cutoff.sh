#!/bin/bash
# What the fast count kernel's epilogue costs at the binary's --rel-cutoff 0.2 (108 matches a read written) against the same kernel
# fetching every row with next to nothing to write (--rel-cutoff 0.75, early exit ablated): SQ / memory counters per launch, separate
# rocprofv3 --pmc passes (never combined with a trace).  -> gpurun_out/r06_cutoff_counters.txt
set -u
R=$PWD
OUT=$R/gpurun_out/prof_r06_cutcmp
mkdir -p $OUT
export TMPDIR=/tmp
B="python $R/bench.py --no-extra --no-e2e --no-variants --no-every-row --no-cpu-baseline --check 0 --steps 4 --warmup 1"
cd /tmp
for cfg in low high; do
  if [ $cfg = low ]; then ARGS="--rel-cutoff 0.2"; ENVV=""; else ARGS="--rel-cutoff 0.75"; ENVV="GANON_HIP_ABLATE=early_exit"; fi
  env $ENVV rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${cfg}_sq1 -- $B $ARGS > $OUT/${cfg}_sq1.log 2>&1
  env $ENVV rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT/${cfg}_sq2 -- $B $ARGS > $OUT/${cfg}_sq2.log 2>&1
  env $ENVV rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/${cfg}_mem -- $B $ARGS > $OUT/${cfg}_mem.log 2>&1
done
cd $R
python - <<'P' > gpurun_out/r06_cutoff_counters.txt
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "prof_r06_cutcmp")
for cfg in ("low", "high"):
    tot = collections.defaultdict(float); n = collections.Counter()
    for sub in ("sq1", "sq2", "mem"):
        for p in glob.glob(os.path.join(out, f"{cfg}_{sub}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(p, newline="")):
                if "gn_ibf_count_fast_kernel" in row["Kernel_Name"]:
                    tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(cfg, {k: round(v / max(1, n[k]), 1) for k, v in sorted(tot.items())}, "launches", dict(n))
P
cat gpurun_out/r06_cutoff_counters.txt
