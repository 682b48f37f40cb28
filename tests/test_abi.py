"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/xmipp_hip.h declares; no compute without a GPU (it must fail loudly)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "xmipp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from xmipp3_amd import _lib
    L = C.CDLL(_lib.lib_path())
    syms = _declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/xmipp_hip.h but not exported"
    # and the Python binding covers the same set
    assert set(_lib.SIGNATURES) == set(syms)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from xmipp3_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.xh_ctx_create(0, None, C.byref(h))
    assert rc != 0
    assert b"no CPU fallback" in L.xh_last_error()
    import xmipp3_amd as xa
    with pytest.raises(xa.XhError):
        xa.Context(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xmipp3_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"
