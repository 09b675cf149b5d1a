#!/bin/bash
mkdir -p gpurun_out
# V6 probe: do a cache-bound level and an HBM-bound level of the HIBF overlap when each launch takes half the wave slots?
for bpc in 0 2 1; do
  if [ $bpc = 0 ]; then E=""; else E="GANON_HIP_ABLATE=hibf_bpc=$bpc"; fi
  env $E timeout 300 python scripts/two_context_probe.py hibf64k_skew 10 2>/dev/null | tail -1 | sed "s/^/bpc=$bpc /"
done > gpurun_out/r06_hibf_overlap_probe.txt
cat gpurun_out/r06_hibf_overlap_probe.txt
timeout 500 python bench_e2e.py --only paired --runs 10 --budget 400 > gpurun_out/r06_e2e_paired_10runs_b.json 2> gpurun_out/r06_e2e_paired_10runs_b.err
echo "paired rc $?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r06_e2e_paired_10runs_b.json"))
for n,v in d["inputs"].items():
    print(n, v.get("rate"), [r["s"] for r in v.get("per_run",[])])
P
