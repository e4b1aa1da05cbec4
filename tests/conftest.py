import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


LIB_OPTIONS = ("UMR_GEMM_TILE", "UMR_NT_SPLITK", "UMR_SPLITK_FENCE", "UMR_NT_ORDER", "UMR_NT256_PERSIST", "UMR_NT256_BM", "UMR_X3_TRACE",
               "UMR_NT256_PH2", "UMR_ATTN_BWD_FUSED", "UMR_BILINEAR_GY", "UMR_HEAD_OUT_BWD_GENERIC")


class _UmrOpts:
    """libumr's debug / A-B options (include/umr.h: umr_set_debug_option).  The library reads its environment ONCE, when it is loaded,
    so tests switch an option between launches through the entry point; same call shapes as pytest's monkeypatch.setenv / delenv,
    every option is put back to what it was when the test ends."""

    def __init__(self):
        self._saved = {}

    def setenv(self, name, value):
        from unmore_amd import ops
        assert name in LIB_OPTIONS, name
        prev = ops.set_debug_option(name, value)
        self._saved.setdefault(name, prev)

    def delenv(self, name, raising=False):
        from unmore_amd import ops
        assert name in LIB_OPTIONS, name
        prev = ops.set_debug_option(name, None)
        self._saved.setdefault(name, prev)

    def undo(self):
        from unmore_amd import ops
        for name, prev in self._saved.items():
            # letter options come back as their character code
            ops.set_debug_option(name, None if prev is None else (chr(prev) if name == "UMR_NT_ORDER" else prev))
        self._saved = {}


@pytest.fixture
def umr_opts():
    o = _UmrOpts()
    yield o
    o.undo()
