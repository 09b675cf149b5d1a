#!/bin/bash
mkdir -p gpurun_out
for t in 1 2 3; do
  timeout 400 python bench_e2e.py --only gz --runs 3 --budget 200 --env GANON_HOST_DEVICE_INFLATE_TURNS=$t > gpurun_out/r06_e2e_gz_turns$t.json 2> gpurun_out/r06_e2e_gz_turns$t.err
  python - <<P
import json
d=json.load(open("gpurun_out/r06_e2e_gz_turns$t.json"))
v=d["inputs"]["gz"]; print("turns=$t", v.get("rate"), v.get("process_wall_s_median"), v.get("error",""))
P
done
