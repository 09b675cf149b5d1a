"""The full-size workloads under the runtime the PRODUCT uses.  pytest and bench.py run the kernels inside a
PyTorch process: torch's bundled libamdhip64 (HIP 7.0.2) is mapped first and serves libganon_hip.so, while ganon-classify maps
/opt/rocm's 7.2.0 -- the runtime the library is linked against.  scripts/runtime_check.py runs BASELINE configs 2, 3 and 4 (flat8g,
hibf64k, flat128g) in a torch-free process (7.2.0) and in one that imports torch first (7.0.2): same match checksums, and the flat
ones are the values bench.py's lines carry (`match_checksum_rank0`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
# bench.py's match_checksum of the default runs (profiles/r04_bench_default_run*.json / r04_bench_flat128g.json)
KNOWN = {"flat8g": "5f2b96725f9d8712", "flat128g": "c5e6eca18923e019"}


def _run(name, *extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "runtime_check.py"), name, *extra], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.splitlines()[-1])


@pytest.mark.parametrize("name", ["flat8g", "hibf64k", "flat128g"])
def test_same_checksum_under_rocm72_and_under_torchs_runtime(name):
    a = _run(name)                     # the runtime libganon_hip.so is linked against
    b = _run(name, "--with-torch")     # what every other GPU test runs under
    assert a["libamdhip64"] and all(p.startswith("/opt/rocm") for p in a["libamdhip64"]), a["libamdhip64"]
    assert any("/torch/lib/" in p for p in b["libamdhip64"]), b["libamdhip64"]
    assert a["libganon_hip"] == b["libganon_hip"] and a["libganon_hip"][0].endswith("ganon_amd/csrc/libganon_hip.so")
    assert a["matches"] > 0 and (a["checksum"], a["matches"], a["n_hashes"]) == (b["checksum"], b["matches"], b["n_hashes"]), (a, b)
    if name in KNOWN:
        assert a["checksum"] == KNOWN[name], a


def test_the_suite_itself_runs_under_torchs_runtime_and_says_so(request):
    # conftest.py prints the mapped libamdhip64 in the header; here it is asserted for this very process
    import ganon_amd
    ganon_amd.load_library()
    hip = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    assert hip, "no HIP runtime mapped"
    import torch
    assert ("/torch/lib/" in hip[0]) == (torch.version.hip is not None)
