"""The C-ABI library loads, exports every symbol include/bbt_hip.h declares
and validates its arguments (no compute: runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

import baseband_tasks_amd as bt
from baseband_tasks_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'bbt_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bbt_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 35
    lib = hip.lib()
    for name in names:
        assert hasattr(lib, name), f'{name} declared in bbt_hip.h but not exported'
    bound = set(hip.SIGNATURES) | {'bbt_last_error'}
    assert set(names) == bound, set(names) ^ bound
    assert lib.bbt_version() >= 101


def test_argument_validation_reports_errors():
    lib = hip.lib()
    plan = C.c_void_p()
    assert lib.bbt_chan_plan_create(C.byref(plan), 1000, 2, -1) != 0
    assert b'power of two' in lib.bbt_last_error()
    assert lib.bbt_chan_plan_create(C.byref(plan), 1024, 3, -1) != 0
    assert b'even' in lib.bbt_last_error()
    assert lib.bbt_chan_plan_create(C.byref(plan), 1024, 2, 0) != 0
    taps = np.zeros((2, 1024), np.float32)
    assert lib.bbt_pfb_plan_create(C.byref(plan), 0, 1024, 2,
                                   taps.ctypes.data_as(C.POINTER(C.c_float))) != 0
    assert b'n_tap' in lib.bbt_last_error()
    resp = np.zeros((1, 300), np.complex64)
    assert lib.bbt_osm_plan_create(C.byref(plan), 300, 2, 1, resp.ctypes.data, 0, None) != 0
    assert b'power of two' in lib.bbt_last_error()
    assert lib.bbt_osm_plan_create(C.byref(plan), 2**25, 2, 1, resp.ctypes.data, 0, None) != 0
    idx = np.array([0, 5], np.int32)
    resp = np.zeros((1, 256), np.complex64)
    assert lib.bbt_osm_plan_create(C.byref(plan), 256, 2, 1, resp.ctypes.data, 0,
                                   idx.ctypes.data_as(C.POINTER(C.c_int32))) != 0
    assert b'out of range' in lib.bbt_last_error()
    assert lib.bbt_osm_execute(None, None, None, 0, None, None, None, None, None) != 0
    assert lib.bbt_memcpy2d(None, 0, None, 0, 0, 0, 7, None) != 0
    taps = np.zeros((5, 3), np.complex64)
    assert lib.bbt_fir_plan_create(C.byref(plan), 5, 3, taps.ctypes.data) != 0
    assert b'even' in lib.bbt_last_error()
    assert lib.bbt_fir_plan_create(C.byref(plan), 5000, 2, taps.ctypes.data) != 0
    assert b'n_tap' in lib.bbt_last_error()
    assert lib.bbt_fir_execute(None, None, None, 0, None) != 0
    assert lib.bbt_osm_execute_channelized_detect(None, None, None, 0, None, None, None, None, 1024,
                                                  0, 0, 64, 1, 1, None) != 0
    assert lib.bbt_osm_detect_bins_max(None, 1024, 64) == -1
    assert lib.bbt_free(C.c_void_p(12345)) != 0 and b'not allocated' in lib.bbt_last_error()
    # destroying null plans is harmless
    assert lib.bbt_osm_plan_destroy(None) == 0 and lib.bbt_chan_plan_destroy(None) == 0
    assert lib.bbt_pfb_plan_destroy(None) == 0 and lib.bbt_fir_plan_destroy(None) == 0
    assert lib.bbt_shift_plan_destroy(None) == 0


def test_python_wrappers_raise():
    import pytest
    with pytest.raises(hip.HipError, match='power of two'):
        hip.ChanPlan(100, 2)
    d = hip.DeviceArray.__new__(hip.DeviceArray)      # views without touching the device
    d.shape, d.dtype, d.ptr, d.owner = (10, 4), np.dtype(np.complex64), 1 << 20, None
    v = d[2:5]
    assert v.shape == (3, 4) and v.ptr == (1 << 20) + 2 * 32 and v.nbytes == 96
    assert d.reshape(5, -1).shape == (5, 8)
    with pytest.raises(TypeError):
        d[3]
    with pytest.raises(ValueError):
        d.reshape(7, 7)
