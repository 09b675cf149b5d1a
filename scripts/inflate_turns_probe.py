#!/usr/bin/env python3
"""Several inflaters of one file taking its steps in turn (gn_inflate_set_turns / gn_inflate_handoff), all on ONE device: what the
hand-over and the decodes-ahead cost when nothing is gained (one device does all the work either way).  Sum of the decode kernels' time
equal to the single inflater's = no step was decoded twice (a wrong guess of where the next own step begins would show there).
  python scripts/inflate_turns_probe.py [--reads N] [--tile T] [--turns 1,2,3]"""
import argparse, json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganon_amd import hip  # noqa: E402
from inflate_probe import synth_fastq  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--tile", type=int, default=8)
ap.add_argument("--turns", default="1,2,3")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
text = synth_fastq(a.reads)
co = zlib.compressobj(6, zlib.DEFLATED, 31)
gz = np.frombuffer((co.compress(text) + co.flush()) * a.tile, dtype=np.uint8)
n_text = len(text) * a.tile
for turns in [int(x) for x in a.turns.split(",")]:
    best = None
    for rep in range(a.reps):
        zs = [hip.HipInflate(gz.size) for _ in range(turns)]
        for i, z in enumerate(zs):
            if turns > 1:
                z.set_turns(turns, i)
            z.feed(gz)
        t0 = time.perf_counter()
        k, done, total = 0, False, 0
        while not done:
            z = zs[k % turns]
            n, done = z.step()
            total += n
            if not done and turns > 1:
                z.handoff(zs[(k + 1) % turns])
            k += 1
        dt = time.perf_counter() - t0
        st = [z.stats() for z in zs]
        for z in zs:
            z.close()
        assert total == n_text, (total, n_text)
        rec = {"turns": turns, "steps": k, "wall_ms": round(dt * 1e3, 1), "gb_s_text": round(n_text / dt / 1e9, 1),
               "decode_ms_sum": round(sum(s["ms_decode"] for s in st), 1), "chain_ms_sum": round(sum(s["ms_chain"] for s in st), 1),
               "resolve_ms_sum": round(sum(s["ms_resolve"] for s in st), 1), "step_calls_ms": round(sum(s["ms_step_wall"] for s in st), 1)}
        if best is None or rec["wall_ms"] < best["wall_ms"]:
            best = rec
    print(json.dumps(best), flush=True)
