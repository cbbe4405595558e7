"""The generic-length engine without a GPU: what its host planner decides (csrc/gen2_host.hpp),
that the index maps of those plans give the transform (the model of tools/fft_gen2_model.py
follows every value through them and compares with numpy.fft), and that the kernel source the
library writes for a length compiles through hipRTC for gfx950 (csrc/rtc.hpp's path: the
compiler needs no device).  The lengths are the reference's default block lengths -- the
smallest 2^a 3^b 5^c 7^d >= n of its NumPy engine, baseband_tasks/fourier/numpy.py:99-126 -- and
the ones its tests use (6174: tests/test_base.py:522-529)."""
import ctypes as C
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'baseband-tasks_amd', 'csrc')


@pytest.fixture(scope='module')
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('gen2') / 'gen2_plan_dump')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-I', CSRC, os.path.join(ROOT, 'tests', 'gen2_plan_dump.cpp'),
                           '-o', exe])

    def run(*args):
        out = subprocess.check_output([exe] + [str(a) for a in args], text=True)
        return [json.loads(line) for line in out.splitlines()]
    run.exe = exe
    return run


def model():
    spec = importlib.util.spec_from_file_location('fft_gen2_model', os.path.join(ROOT, 'tools', 'fft_gen2_model.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


LENGTHS = [2, 3, 14, 30, 100, 490, 540, 1000, 1536, 3000, 3087, 3402, 3430, 6174, 6561, 8100, 8192, 3125, 2401]


def test_plans_of_the_host_planner_transform_correctly(dump):
    """Every plan: the stage list multiplies to n, no stage asks a thread for more than its
    registers hold, the exchange area holds every buffer -- and the model, fed the plan's stages,
    threads and pitches, reproduces numpy.fft (forward on the plan, inverse on its reversal)."""
    m = model()
    args = []
    for n in LENGTHS:
        args += [n, 1]
    args += [490, 8, 486, 8, 30, 4]
    for pmax in (20, 12, 10):                # (row pass / column passes / channelizer: gen2_host.hpp g2_pmax)
      for plans in dump('plan', pmax, *args):
        for key, sign in (('forward', -1), ('reversed', +1)):
            g = plans[key]
            n, fac, tj = g['n'], g['fac'], g['tj']
            assert int(np.prod(fac)) == n and all(2 <= r <= min(16, pmax) for r in fac) or n == 1
            assert g['threads'] % 64 == 0 and g['threads'] >= tj * g['ct']
            ns, need = 1, 0
            for s, r in enumerate(fac):
                b = -(-(n // r) // tj)
                assert b <= 4 and b * r <= pmax, (n, fac, tj)        # (BBT_G2_MAXB, points per thread)
                assert g['slots'] >= b * r
                assert g['pitch'][s] >= ns * r
                if s + 1 < len(fac):
                    need = max(need, (n // (ns * r)) * g['pitch'][s])
                ns *= r
            assert g['lds_elems'] >= need * g['ct']
            if n <= 3500 and pmax != 12:                             # (the model is plain Python)
                m.run(n, fac=fac, sign=sign, tj=tj, pitches=g['pitch'][:len(fac) - 1], verbose=False)
        assert plans['forward']['fac'] == plans['reversed']['fac'][::-1]
        assert plans['forward']['tj'] == plans['reversed']['tj']
        assert plans['forward']['slots'] == plans['reversed']['slots']


def test_the_model_and_the_planner_agree_on_stages_and_threads(dump):
    m = model()
    for n, plans in zip(LENGTHS, dump('plan', 20, *sum(([n, 1] for n in LENGTHS), []))):
        fac = m.factorise(n)
        assert len(fac) == len(plans['forward']['fac'])
        assert m.threads(n, fac) == plans['forward']['tj'], n


def test_split_rule_prefers_workgroups_that_pack_a_cu(dump):
    """The reference's default blocks at 800 / 600 MHz and for Resample: both kernels of
    the chosen split have 1, 2, 4 or 8 waves (measured: 29.5 -> 34-35.7 Gsamples/s at 800 MHz)."""
    for n, got in zip((1666980, 3936600, 1049760, 93312), dump('split', 1666980, 3936600, 1049760, 93312)):
        assert got['n1'] * got['n2'] == n and got['n2'] <= 8192 and got['n1'] <= 1024
        for k in ('col', 'row'):
            # (blocks up to 2^17 samples: their workgroups are small, three waves pack as well)
            assert got[k]['threads'] // 64 in ((1, 2, 3, 4, 8) if n <= 2**17 else (1, 2, 4, 8)), (n, got[k])
    assert [(g['n1'], g['n2']) for g in dump('split', 1666980, 3936600, 1049760)] == [(540, 3087), (486, 8100), (240, 4374)]
    assert dump('split', 11059200) == [None]         # (N2 <= 8192 needs N1 >= 1350: the general rule takes over)


def test_short_blocks_fill_their_waves(dump):
    """Blocks of a few 10^4 samples (the defaults at low DM: 31 104 = 2^7 3^5 at 1400 MHz, DM 10) split
    into columns and rows that need a dozen threads each: a column tile takes 16 or 32 columns, a
    row workgroup up to 8 rows, so that neither kernel runs waves that are mostly idle (measured:
    Dedisperse on 31 104-sample blocks 26.7 -> 44.7 Gsamples/s), and no column is shorter than 16
    points.  Reference: the block lengths of base.py:750-758 with fourier/numpy.py:99-126."""
    got, = dump('split', 31104)
    assert (got['n1'], got['n2']) == (36, 864) and got['col']['ct'] == 32
    for n in (8232, 19200, 20000, 23328, 25725, 31104, 39366, 46656, 54432, 65610, 93312, 100000):
        g, = dump('split', n)
        assert g['n1'] * g['n2'] == n and g['n1'] >= 16
        for k in ('col', 'row'):
            fill = g[k]['tj'] * g[k]['ct'] / g[k]['threads']
            # (a 16-point column is one thread's work and a tile has at most 32 columns: half a wave)
            assert fill >= 0.5, (n, k, g[k])


def test_wide_column_tiles_of_short_blocks_are_sound(dump):
    """The plans the split rule picks for short blocks, with their 16- / 32-column tiles and several
    rows per workgroup: registers, threads and exchange area within what the kernels have, and the
    model reproduces numpy.fft on both halves.  10 080 and 10 206 are Resample's blocks on
    10 000-sample frames with pad 16 / 32 and 64 (tests/test_gpu_parity.py,
    test_random_filter_bank_channelizer_and_resampler_geometries; sampling.py:211-220 of the reference)."""
    m = model()
    for n in (8232, 10080, 10206, 19200, 20000, 31104, 46656, 65610, 93312, 100000, 129024):
        g, = dump('split', n)
        assert g is not None and g['n1'] * g['n2'] == n
        for k, pmax, lds_cap in (('col', 20, 64 * 1024), ('row', 20, 160 * 1024)):     # (BBT_G2_PMAX for both)
            p = g[k]
            fac, tj, nn = p['fac'], p['tj'], p['n']
            assert int(np.prod(fac)) == nn and all(2 <= r <= min(16, pmax) for r in fac)
            assert p['threads'] % 64 == 0 and tj * p['ct'] <= p['threads'] <= 1024
            assert p['lds_elems'] * 8 <= lds_cap, (n, k, p)
            ns, need = 1, 0
            for s, r in enumerate(fac):
                b = -(-(nn // r) // tj)
                assert b <= 4 and b * r <= pmax and p['slots'] >= b * r, (n, k, p)
                assert p['pitch'][s] >= ns * r
                if s + 1 < len(fac):
                    need = max(need, (nn // (ns * r)) * p['pitch'][s])
                ns *= r
            assert p['lds_elems'] >= need * p['ct'], (n, k, p)
            if nn <= 3500:
                m.run(nn, fac=fac, sign=-1, tj=tj, pitches=p['pitch'][:len(fac) - 1], verbose=False)


def _hiprtc():
    for name in ('libhiprtc.so.7', 'libhiprtc.so', '/opt/rocm/lib/libhiprtc.so'):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    return None


def _compile(rtc, src):
    """csrc/rtc.hpp's call of hipRTC, with the library's options; returns the code object."""
    prog = C.c_void_p()
    assert rtc.hiprtcCreateProgram(C.byref(prog), src, b'bbt_g2.hip', 0, None, None) == 0
    opts = [b'--offload-arch=gfx950', b'-I' + CSRC.encode(), b'-O3', b'-std=c++17', b'-Wno-unused-value',
            b'-mllvm', b'-simplifycfg-sink-common=false']
    rc = rtc.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*opts))
    n = C.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    log = C.create_string_buffer(n.value + 1)
    rtc.hiprtcGetProgramLog(prog, log)
    assert rc == 0, log.value.decode(errors='replace')[-2000:]
    assert rtc.hiprtcGetCodeSize(prog, C.byref(n)) == 0 and n.value > 10000
    code = C.create_string_buffer(n.value)
    assert rtc.hiprtcGetCode(prog, code) == 0
    rtc.hiprtcDestroyProgram(C.byref(prog))
    return code.raw


def test_generated_source_compiles_through_hiprtc(dump):
    """What csrc/rtc.hpp does at plan time, without a device: source for one length -> hipRTC with
    the library's options -> a code object for gfx950 that holds both entry points."""
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip('libhiprtc.so not found')
    code = _compile(rtc, subprocess.check_output([dump.exe, 'source', '6174'], text=True).encode())
    assert b'k_small' in code and b'k_rows' in code


@pytest.mark.parametrize('n', [10080, 10206, 31104])
def test_two_level_source_of_short_blocks_compiles(dump, n):
    """The translation unit of a short two-level block (row kernel with several rows per workgroup,
    column passes with 16 / 32 columns per tile), as bbt_osm_plan_create writes it: compiles for
    gfx950 with its three entry points.  10 080 / 10 206: Resample's blocks on 10000-sample frames."""
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip('libhiprtc.so not found')
    src = subprocess.check_output([dump.exe, 'source2', str(n)], text=True)
    assert 'BBT_G2_KERNEL_COL(k_first' in src
    code = _compile(rtc, src.encode())
    assert b'k_row' in code and b'k_first' in code and b'k_last' in code
    readelf = '/opt/rocm/lib/llvm/bin/llvm-readelf'
    if os.path.exists(readelf):
        # (what the kernels take: no scratch memory -- nothing spilled --, exchange areas as planned)
        obj = os.path.join(os.path.dirname(dump.exe), f'g2_{n}.co')
        with open(obj, 'wb') as f:
            f.write(code)
        notes = subprocess.check_output([readelf, '--notes', obj], text=True)
        import re
        found = re.findall(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+)', notes, re.S)
        assert sorted(name for _, name, _ in found) == ['k_first', 'k_last', 'k_row']
        for lds, name, scratch in found:
            assert int(scratch) == 0 and int(lds) <= 64 * 1024, (n, name, lds, scratch)
