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
    with open(os.path.join(ROOT, 'include', 'bbt_hip.h')) as f:
        text = f.read()
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
    assert lib.bbt_chan_plan_create(C.byref(plan), 1001, 2, -1) != 0      # 7 x 11 x 13
    assert b'power of two' in lib.bbt_last_error()
    assert lib.bbt_chan_plan_create(C.byref(plan), 1024, 3, -1) != 0
    assert b'even' in lib.bbt_last_error()
    assert lib.bbt_chan_plan_create(C.byref(plan), 1024, 2, 0) != 0
    taps = np.zeros((2, 1024), np.float32)
    assert lib.bbt_pfb_plan_create(C.byref(plan), 0, 1024, 2,
                                   taps.ctypes.data_as(C.POINTER(C.c_float))) != 0
    assert b'n_tap' in lib.bbt_last_error()
    resp = np.zeros((1, 311), np.complex64)
    assert lib.bbt_osm_plan_create(C.byref(plan), 311, 2, 1, resp.ctypes.data, 0, None) != 0      # prime
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
    # entry points added in round 2
    assert lib.bbt_osm_plan_defer(None, None) != 0 and b'null plan' in lib.bbt_last_error()
    assert lib.bbt_osm_plan_fusable(None, 1024) == 0
    raw = np.zeros(64, np.uint8)
    assert lib.bbt_unpack(raw.ctypes.data, raw.ctypes.data, 3, 64, 32, 2, 128, 2, 1, 0, None) != 0
    assert b'whole sets' in lib.bbt_last_error()
    assert lib.bbt_unpack(raw.ctypes.data, raw.ctypes.data, 2, 64, 32, 3, 8, 1, 1, 0, None) != 0
    assert b'bits per component' in lib.bbt_last_error()
    assert lib.bbt_unpack(raw.ctypes.data, raw.ctypes.data, 2, 64, 32, 2, 129, 1, 1, 0, None) != 0
    assert b'do not fit' in lib.bbt_last_error()
    assert lib.bbt_unpack(raw.ctypes.data, raw.ctypes.data, 2, 64, 32, 8, 8, 1, 1, 7, None) != 0
    assert lib.bbt_comm_unique_id(None, 128) != 0
    assert lib.bbt_comm_unique_id(raw.ctypes.data, 16) != 0 and b'128' in lib.bbt_last_error()
    comm = C.c_void_p()
    assert lib.bbt_comm_init(C.byref(comm), 2, 5, raw.ctypes.data, 128) != 0 and b'rank 5 of 2' in lib.bbt_last_error()
    assert lib.bbt_comm_init(C.byref(comm), 1, 0, raw.ctypes.data, 8) != 0
    assert lib.bbt_bcast_chirp(None, None, 0, 0, None) != 0 and lib.bbt_gather_output(None, None, None, 0, None) != 0
    assert lib.bbt_comm_destroy(None) == 0
    assert lib.bbt_pool_set_stream(None) == 0
    # destroying null plans is harmless
    assert lib.bbt_osm_plan_destroy(None) == 0 and lib.bbt_chan_plan_destroy(None) == 0
    assert lib.bbt_pfb_plan_destroy(None) == 0 and lib.bbt_fir_plan_destroy(None) == 0
    assert lib.bbt_shift_plan_destroy(None) == 0


def test_python_wrappers_raise():
    import pytest
    with pytest.raises(hip.HipError, match='power of two'):
        hip.ChanPlan(22, 2)
    assert hip.OsmPlan.fusable.__doc__
    d = hip.DeviceArray.__new__(hip.DeviceArray)      # views without touching the device
    d.shape, d.dtype, d._ptr, d.owner = (10, 4), np.dtype(np.complex64), 1 << 20, None
    v = d[2:5]
    assert v.shape == (3, 4) and v.ptr == (1 << 20) + 2 * 32 and v.nbytes == 96
    assert d.reshape(5, -1).shape == (5, 8)
    with pytest.raises(TypeError):
        d[3]
    with pytest.raises(ValueError):
        d.reshape(7, 7)


def test_sanitized_host_build_passes_the_abi_checks():
    """AddressSanitizer + UBSan build of the host side (tools/build_sanitize.sh):
    the symbol and argument-validation tests above run against it in a child
    process with the ASan runtime preloaded; any report aborts the child."""
    import subprocess
    import sys
    import pytest
    if os.environ.get('BBT_HIP_LIB'):
        pytest.skip('already running against an alternative build')
    script = os.path.join(ROOT, 'tools', 'build_sanitize.sh')
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc: cannot make the sanitizer build here')
    lib = os.path.join(ROOT, 'build', 'libbbt_hip_asan.so')
    src = [os.path.join(ROOT, 'baseband-tasks_amd', 'csrc', f)
           for f in ('bbt_hip.hip', 'bbt_kernels.hpp', 'fft_core.hpp', 'fft_generic.hpp', 'gen_kernels.hpp')]
    src = [f for f in src if os.path.exists(f)] + [os.path.join(ROOT, 'include', 'bbt_hip.h')]
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in src):
        subprocess.check_call([script], stdout=subprocess.DEVNULL)
    runtime = subprocess.check_output([script, '--runtime'], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=runtime, BBT_HIP_LIB=lib,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-p', 'no:cacheprovider', __file__,
                        '-k', 'declared_symbol or argument_validation or wrappers_raise'],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert '3 passed' in r.stdout


def build_c_example():
    """gcc build of tests/cabi_example.c against include/bbt_hip.h and the in-tree library."""
    import subprocess
    exe = os.path.join(ROOT, 'build', 'cabi_example')
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, 'baseband-tasks_amd', 'lib')
    hip.lib()                                            # (builds nothing: raises if the library is missing)
    subprocess.check_call(['gcc', '-std=c99', '-D_DEFAULT_SOURCE', '-Wall', '-Wextra', '-Werror',
                           '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'cabi_example.c'),
                           '-L', libdir, '-lbbt_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib',
                           '-lm', '-o', exe])
    return exe


def test_plain_c_program_links_against_the_abi():
    """The header is C (not C++) and the library needs nothing but itself: a gcc
    -Wall -Wextra -Werror build of a C host program links; without a GPU it
    stops at the first call with the library's own error text."""
    import shutil
    import subprocess
    import pytest
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    exe = build_c_example()
    if hip.available():
        pytest.skip('a GPU is present: the program is run by the -m gpu suite')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and 'bbt_device_count' in r.stderr and 'hipGetDeviceCount' in r.stderr
