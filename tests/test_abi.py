"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every
symbol include/rt_hip.h declares, and refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from raytracing_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _ffi.lib()
    names = _declared("rt_hip.h")
    assert sorted(names) == sorted(_ffi.SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_no_device_no_fallback():
    L = _ffi.lib()
    if L.rt_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = L.rt_create(0, C.byref(h))
    assert rc == -2 and not h.value  # RT_ERR_NO_DEVICE
    assert b"no HIP device" in L.rt_last_error_string()
    with pytest.raises(_ffi.RtError):
        _ffi.HipContext(0)
