"""CPU-only: the C-ABI library loads and exports every symbol include/vdet_hip.h declares, and the
product never imports the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'vdet_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vdet_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    from vdetlib_amd import _lib
    so = _lib.LIB_PATH
    if not os.path.isfile(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names            # the python binding covers the whole header
    _lib.load_library()


def test_version_string():
    from vdetlib_amd import _lib
    L = _lib.load_library()
    assert b"gfx950" in L.vdet_version()


def test_no_gpu_fails_loudly():
    """Without a GPU the product raises -- there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from vdetlib_amd.utils import cython_nms
    with pytest.raises(RuntimeError):
        cython_nms.nms(np.zeros((3, 5), np.float32), 0.3)


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, 'vdetlib_amd')):
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.hpp', '.h', '.cpp', '.sh')):
                txt = open(os.path.join(dp, fn), errors='ignore').read()
                if re.search(r'^\s*(from|import)\s+oracle\b|libvdet_oracle|oracle/', txt, flags=re.M):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
