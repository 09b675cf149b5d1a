"""Which library loses rows written right behind gn_filter_upload_ibf(rows = NULL)?  
usage: GANON_HIP_LIB=<old or new .so> python scripts/memset_race_probe.py -> one JSON line"""
import json
import os
import sys

if os.environ.get("PROBE_IMPORT_TORCH"):  # as under pytest (tests/conftest.py) and bench.py: PyTorch's bundled HIP runtime is in the process first
    import torch
    torch.cuda.is_available()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ganon_amd  # noqa: E402
from test_upload_order import streamed_rows_survive  # noqa: E402

ganon_amd.load_library()
out = {"lib": os.environ.get("GANON_HIP_LIB", "default"), "torch_first": bool(os.environ.get("PROBE_IMPORT_TORCH")),
       "hip_runtime": sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l})}
out["first_call_tiny_bad_of_1"] = streamed_rows_survive(ganon_amd, 1 << 12, 64, 16, 1, settle=0.0)  # first GPU work of the process
out["big_4g_bad_of_9"] = streamed_rows_survive(ganon_amd, 1 << 23, 4096, 256, 9)
out["tiny_bad_of_60"] = streamed_rows_survive(ganon_amd, 1 << 12, 64, 16, 60, settle=0.0)
print(json.dumps(out))
