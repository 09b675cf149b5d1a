#!/bin/bash
# VERDICT r5 item 7, first look: the block search costs a fifth of the decode kernel and its length does not depend on the chunk size
# (it ends at the chunk's first block start) -- larger chunks, fewer searches per byte?
mkdir -p gpurun_out
rm -f gpurun_out/r06_inflate_chunk_sizes.jsonl
for c in 0 65536 131072 262144; do
  timeout 300 python scripts/inflate_probe.py --reads 1000000 --tile 8 --chunk $c --reps 3 --out gpurun_out/r06_inflate_chunk_sizes.jsonl > /dev/null 2>gpurun_out/r06_inflate_chunk_sizes.err
done
python - <<'P'
import json
for ln in open("gpurun_out/r06_inflate_chunk_sizes.jsonl"):
    r=json.loads(ln); b=r["best"]
    print("chunk", r["chunk"], "equal", r["bytes_equal"], "wall GB/s", round(b["text_GBps_wall"],1), "steps", b["steps"], "chunks", b["chunks"], "ms decode/chain/resolve", round(b["ms_decode"],1), round(b["ms_chain"],1), round(b["ms_resolve"],1), "fixups", b["fixups"], "prof", [round(x,1) for x in b["prof_ms"][:4]])
P
