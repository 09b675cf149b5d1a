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
        return ["test order: hot-path parity (SURVEY 8a) first, boundary next, f-rows and the bench line last (conftest.TIERS)",
                f"HIP runtime mapped by this process: {', '.join(libs) or 'none yet'} (torch {torch.__version__}, hip {torch.version.hip}); "
                f"the product binaries use /opt/rocm ({os.path.realpath('/opt/rocm')})"]
    except Exception as e:  # noqa: BLE001
        return [f"HIP runtime: unknown ({e!r})"]


# `pytest -x` stops at the first failure, and files are collected alphabetically: test_bench_line, test_build_gpu, test_cli_fuzz,
# test_first_contact, test_gpu_devgzip ... used to run BEFORE test_gpu_parity.  A flaky input-side case would have left every
# hot-path parity row unreached.  Order of the run, by what a failure would cost:
#   0  hot-path parity, SURVEY 8 rows (a): device == oracle on hashes / counts / matches, toy and full size, KATs through the binary
#   1  the boundary: C ABI, CLI, filter files, partition / multi-rank, oracle pins
#   2  f-rows (input side, builder, reassign, first contact) and the bench line
TIERS = (
    ("test_gpu_parity", "test_gpu_fullsize", "test_gpu_fullsize_large", "test_cli_kat", "test_gpu_on_demand", "test_gpu_padded_rows",
     "test_gpu_fuzz", "test_gpu_segmented", "test_oracle_kat"),
    ("test_abi_cpu", "test_gpu_gather", "test_partition_gloo", "test_partition_cli", "test_upload_order", "test_cli_fuzz", "test_ibf_file",
     "test_inspect_filter", "test_verify_filter", "test_reference_order", "test_report_rep", "test_runtime72", "test_reader_formats",
     "test_host_tunables"),
)


def _tier(item) -> int:
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    for t, names in enumerate(TIERS):
        if mod in names:
            return t * 100 + names.index(mod)
    return len(TIERS) * 100


_T0 = [None]


def pytest_sessionstart(session):
    import time
    _T0[0] = time.time()


def pytest_terminal_summary(terminalreporter):
    import time
    if _T0[0] is not None:
        terminalreporter.write_line(f"suite wall time: {time.time() - _T0[0]:.0f} s (budget on the driver's box: 900 s of its 1200 s step limit)")


def pytest_collection_modifyitems(config, items):
    items.sort(key=_tier)   # stable: the order inside a file stays as written
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
