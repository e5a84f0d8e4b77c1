"""The C-ABI library loads on CPU and exports exactly the symbols include/rectools_hip.h declares;
the ctypes signature table lists every one of them (no compute calls: no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rectools_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from rectools_amd import build

    return build.build()


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    syms = _declared_symbols()
    assert len(syms) >= 4
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in include/rectools_hip.h but not exported"


def test_ctypes_table_matches_header():
    from rectools_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_host_only_entry_points(lib_path):
    from rectools_amd import _lib

    lib = _lib.load()
    assert lib.rt_version() >= 100
    assert lib.rt_topk_workspace_bytes(64, 26744, 10, 64) > 0


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    from rectools_amd import _lib
    from rectools_amd.rank import HipRanker

    with pytest.raises(_lib.HipLibraryError):
        HipRanker("dot", "cpu", np.zeros((1, 4), np.float32), np.zeros((2, 4), np.float32))
    with pytest.raises(_lib.HipLibraryError):
        HipRanker("dot", "cuda", np.zeros((1, 4), np.float32), np.zeros((2, 4), np.float32))


def test_docs_quote_the_current_entry_point_count():
    """DESIGN.md / INTEGRATION.md / README.md state how many C entry points the library has: keep them honest."""
    n = len(_declared_symbols())
    for name, pattern in (("DESIGN.md", r"\((\d+) `extern \"C\"` entry points"), ("INTEGRATION.md", r"\((\d+) `extern \"C\"` functions"),
                          ("README.md", r"\((\d+) C entry points\)")):
        m = re.search(pattern, open(os.path.join(ROOT, name)).read())
        assert m and int(m.group(1)) == n, f"{name} quotes {m.group(1) if m else None} entry points, the header declares {n}"
