import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """which HIP runtime the in-process kernels run under: PyTorch's bundled libamdhip64 is mapped first and serves libganon_hip.so
    too; the product binaries map /opt/rocm's (tests/test_runtime72.py runs the full-size workloads under both)"""
    try:
        import torch
        libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
        return [f"HIP runtime mapped by this process: {', '.join(libs) or 'none yet'} (torch {torch.__version__}, hip {torch.version.hip}); "
                f"the product binaries use /opt/rocm ({os.path.realpath('/opt/rocm')})"]
    except Exception as e:  # noqa: BLE001
        return [f"HIP runtime: unknown ({e!r})"]


def pytest_collection_modifyitems(config, items):
    # gpu tests are skipped (not failed) when no GPU is visible and the user did not ask for -m gpu
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return  # explicitly requested: let them run and fail loudly
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _product_path_after_each_test():
    """a test that switched a kernel path off (gpu_util.SW) never leaks that into the next test"""
    yield
    import gpu_util
    gpu_util.SW.clear()


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kat_classify.json")) as f:
        return json.load(f)
