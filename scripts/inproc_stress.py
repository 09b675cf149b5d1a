"""the suite's failing sequence (ganon-build -> load_ibf -> submit -> fetch -> dense tap) many times in ONE process,
with different filters behind each other as in tests/test_build_gpu.py.  usage: [GANON_HIP_LIB=...] python scripts/inproc_stress.py N"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ganon_amd  # noqa: E402
import test_build_gpu as t  # noqa: E402

ganon_amd.load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
combos = [(19, 32, 4), (19, 32, 2), (21, 23, 4), (27, 27, 4), (19, 32, 0), (19, 32, 3)]
fails = []
with tempfile.TemporaryDirectory() as d:
    for i in range(n):
        k, w, h = combos[i % len(combos)]
        sub = os.path.join(d, f"r{i}")
        os.mkdir(sub)
        inp, _, names = t.write_inputs(sub, t.SEQS)
        out, _ = t.run_build(sub, inp, k=k, w=w, h=h)
        try:
            t.check_filter(ganon_amd, out, t.SEQS, names, k, w, h, 0.05, 0)
        except AssertionError as e:
            fails.append(dict(iteration=i, k=k, w=w, h=h, what=str(e)[:1500]))
print(json.dumps(dict(lib=os.environ.get("GANON_HIP_LIB", "default"), runs=n, failed=len(fails), fails=fails[:5])))
