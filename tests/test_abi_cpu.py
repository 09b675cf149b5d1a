"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/ganon_hip.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ganon_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ganon_amd
    from ganon_amd.hip import ABI_SYMBOLS
    L = ganon_amd.load_library()
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/ganon_hip.h but not exported"
    assert sorted(ABI_SYMBOLS) == declared


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import ganon_amd
    with pytest.raises(ganon_amd.GanonHipError):
        ganon_amd.device_count()
    with pytest.raises(ganon_amd.GanonHipError):
        ganon_amd.HipFilter.ibf(np.zeros((10, 1), dtype=np.uint64), 64, 10, 2)


def test_product_does_not_reference_oracle():
    # the oracle is test infrastructure: nothing under ganon_amd/ may import, include or link it
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "ganon_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"ganon_oracle|import oracle|from oracle|libganon_oracle|gno_", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_hibf_row_stride_words():
    # host-side arithmetic of the ABI (no device): the row stride gn_filter_upload_hibf gives an IBF of ceil(bins / 64) words a row --
    # the next power of two up to one 128-byte line, whole lines beyond -- is what ganon-classify's placement counts
    import ganon_amd
    L = ganon_amd.load_library()
    got = [int(L.gn_hibf_row_stride_words(w)) for w in (1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 32, 33, 64, 100)]
    assert got == [1, 2, 4, 4, 8, 8, 16, 16, 16, 32, 32, 48, 64, 112]
    for w in range(1, 200):
        s = int(L.gn_hibf_row_stride_words(w))
        assert s >= w and s < 2 * w + 16 and (128 % (s * 8) == 0 or (s * 8) % 128 == 0)
