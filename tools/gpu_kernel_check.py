"""Stand-alone check of libbbt_hip.so against numpy on a GPU box (dev tool).

Usage: python tools/gpu_kernel_check.py [--big]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = C.CDLL(os.environ.get('BBT_LIB') or os.path.join(HERE, '..', 'baseband-tasks_amd', 'lib', 'libbbt_hip.so'))
LIB.bbt_last_error.restype = C.c_char_p


def chk(rc):
    if rc != 0:
        raise RuntimeError(LIB.bbt_last_error().decode())


class Dev:
    def __init__(self, nbytes):
        self.p = C.c_void_p()
        chk(LIB.bbt_malloc(C.byref(self.p), C.c_size_t(nbytes)))
        self.nbytes = nbytes

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.nbytes)
        chk(LIB.bbt_memcpy_h2d(d.p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), None))
        chk(LIB.bbt_device_sync())
        return d

    def to_host(self, shape, dtype):
        out = np.empty(shape, dtype)
        chk(LIB.bbt_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.p, C.c_size_t(out.nbytes), None))
        chk(LIB.bbt_device_sync())
        return out

    def free(self):
        chk(LIB.bbt_free(self.p))


def relerr(a, b):
    return np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()), np.abs(a - b).max() / np.sqrt(np.mean(np.abs(b) ** 2))


def noise(shape, rng):
    return (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)


def check_chan(n, S, nspec, direction, rng):
    x = noise((nspec * n, S), rng)
    plan = C.c_void_p()
    chk(LIB.bbt_chan_plan_create(C.byref(plan), n, S, direction))
    din = Dev.from_host(x)
    dout = Dev(x.nbytes)
    chk(LIB.bbt_chan_execute(plan, din.p, dout.p, C.c_int64(nspec), None))
    got = dout.to_host((nspec, n, S), np.complex64)
    xr = x.reshape(nspec, n, S).astype(np.complex128)
    ref = np.fft.fft(xr, axis=1) if direction < 0 else np.fft.ifft(xr, axis=1)
    e = relerr(got, ref)
    print(f"chan n={n} S={S} nspec={nspec} dir={direction}: relL2={e[0]:.2e} max/rms={e[1]:.2e}")
    chk(LIB.bbt_chan_plan_destroy(plan))
    din.free(), dout.free()
    assert e[0] < 5e-7 and e[1] < 5e-6, e


def osm_ref(x, H, N, hop, valid_start, nblk, resp_index):
    S = x.shape[1]
    out = np.zeros((nblk * hop, S), np.complex128)
    for b in range(nblk):
        blk = x[b * hop:b * hop + N].astype(np.complex128)
        ft = np.fft.fft(blk, axis=0)
        ft *= H[resp_index].T
        r = np.fft.ifft(ft, axis=0)
        out[b * hop:(b + 1) * hop] = r[valid_start:valid_start + hop]
    return out


def check_osm(N, S, nblk, rng, C_resp=1, timing=False):
    pad = N // 5
    hop = N - pad
    valid_start = pad // 2 + 1
    L = (nblk - 1) * hop + N
    x = noise((L, S), rng)
    H = np.exp(2j * np.pi * rng.uniform(size=(C_resp, N))).astype(np.complex64)
    resp_index = (np.arange(S) % C_resp).astype(np.int32)
    plan = C.c_void_p()
    chk(LIB.bbt_osm_plan_create(C.byref(plan), C.c_int64(N), S, C_resp,
                                H.ctypes.data_as(C.c_void_p), 0,
                                resp_index.ctypes.data_as(C.POINTER(C.c_int32))))
    din = Dev.from_host(x)
    dout = Dev(nblk * hop * S * 8)
    chk(LIB.bbt_memset(dout.p, 0xff, C.c_size_t(dout.nbytes), None))
    chk(LIB.bbt_osm_execute_regular(plan, din.p, dout.p, C.c_int64(nblk), C.c_int64(0),
                                    C.c_int64(0), C.c_int64(hop), C.c_int32(valid_start), None))
    chk(LIB.bbt_device_sync())
    got = dout.to_host((nblk * hop, S), np.complex64)
    if N * nblk <= (1 << 22):
        ref = osm_ref(x, H, N, hop, valid_start, nblk, resp_index)
        e = relerr(got, ref)
    else:  # spot check first and last block only
        ref0 = osm_ref(x, H, N, hop, valid_start, 1, resp_index)
        e0 = relerr(got[:hop], ref0)
        refl = osm_ref(x[(nblk - 1) * hop:], H, N, hop, valid_start, 1, resp_index)
        el = relerr(got[(nblk - 1) * hop:], refl)
        e = (max(e0[0], el[0]), max(e0[1], el[1]))
    print(f"osm N={N} S={S} nblk={nblk} C={C_resp}: relL2={e[0]:.2e} max/rms={e[1]:.2e}")
    if timing:
        chk(LIB.bbt_osm_timing_enable(plan, 1))
        for _ in range(5):
            chk(LIB.bbt_osm_execute_regular(plan, din.p, dout.p, C.c_int64(nblk), C.c_int64(0),
                                            C.c_int64(0), C.c_int64(hop), C.c_int32(valid_start), None))
        ms = (C.c_double * 3)()
        nl = C.c_int64()
        chk(LIB.bbt_osm_timing_read(plan, ms, C.byref(nl)))
        tot = sum(ms)
        print(f"   passes ms (A,B,C) per 5 runs: {ms[0]:.3f} {ms[1]:.3f} {ms[2]:.3f} launches={nl.value}"
              f" -> {5 * nblk * hop / tot / 1e3:.1f} Msamples/s valid")
        chk(LIB.bbt_osm_timing_enable(plan, 0))
        chk(LIB.bbt_device_sync())
        t0 = time.perf_counter()
        for _ in range(10):
            chk(LIB.bbt_osm_execute_regular(plan, din.p, dout.p, C.c_int64(nblk), C.c_int64(0),
                                            C.c_int64(0), C.c_int64(hop), C.c_int32(valid_start), None))
        chk(LIB.bbt_device_sync())
        dt = (time.perf_counter() - t0) / 10
        print(f"   wall {dt * 1e3:.3f} ms per {nblk} blocks -> {nblk * hop / dt / 1e6:.1f} Msamples/s")
    chk(LIB.bbt_osm_plan_destroy(plan))
    din.free(), dout.free()
    assert e[0] < 1e-6 and e[1] < 1e-5, e


def check_fused(N, S, nblk, n_chan, rng, C_resp=1, timing=False, first_spec=0, drop_tail=0):
    """bbt_osm_execute_channelized against numpy: channelize(overlap_save(x))."""
    pad = N // 5 + 3
    hop = N - pad
    valid_start = pad // 2 + 1
    L = (nblk - 1) * hop + N
    x = noise((L, S), rng)
    H = np.exp(2j * np.pi * rng.uniform(size=(C_resp, N))).astype(np.complex64)
    resp_index = (np.arange(S) % C_resp).astype(np.int32)
    plan = C.c_void_p()
    chk(LIB.bbt_osm_plan_create(C.byref(plan), C.c_int64(N), S, C_resp,
                                H.ctypes.data_as(C.c_void_p), 0,
                                resp_index.ctypes.data_as(C.POINTER(C.c_int32))))
    nspec_all = (nblk * hop) // n_chan
    nspec = nspec_all - first_spec - drop_tail
    din = Dev.from_host(x)
    dout = Dev(nspec * n_chan * S * 8)
    chk(LIB.bbt_memset(dout.p, 0xff, C.c_size_t(dout.nbytes), None))
    io = (np.arange(nblk) * hop).astype(np.int64)
    vs = np.full(nblk, valid_start, np.int32)
    vc = np.full(nblk, hop, np.int32)
    p64, p32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)

    def run():
        chk(LIB.bbt_osm_execute_channelized(plan, din.p, dout.p, C.c_int64(nblk), io.ctypes.data_as(p64),
                                            io.ctypes.data_as(p64), vs.ctypes.data_as(p32),
                                            vc.ctypes.data_as(p32), n_chan, C.c_int64(first_spec),
                                            C.c_int64(nspec), None))
    run()
    chk(LIB.bbt_device_sync())
    got = dout.to_host((nspec, n_chan, S), np.complex64)
    if N * nblk <= (1 << 22):
        y = osm_ref(x, H, N, hop, valid_start, nblk, resp_index)
        ref = np.fft.fft(y[:nspec_all * n_chan].reshape(nspec_all, n_chan, S), axis=1)
        ref = ref[first_spec:first_spec + nspec]
        e = relerr(got, ref)
        # the spectra straddling block seams, separately
        seams = [(b * hop) // n_chan - first_spec for b in range(1, nblk) if (b * hop) % n_chan]
        seams = [s_ for s_ in seams if 0 <= s_ < nspec]
        es = relerr(got[seams], ref[seams]) if seams else (0., 0.)
    else:
        y = osm_ref(x, H, N, hop, valid_start, 2, resp_index)     # first two blocks only
        k = (2 * hop) // n_chan
        ref = np.fft.fft(y[:k * n_chan].reshape(k, n_chan, S), axis=1)[first_spec:]
        e = relerr(got[:k - first_spec], ref)
        s_ = hop // n_chan - first_spec
        es = relerr(got[s_:s_ + 1], ref[s_:s_ + 1])
    print(f"fused N={N} S={S} nblk={nblk} n_chan={n_chan} C={C_resp} first={first_spec}: relL2={e[0]:.2e} "
          f"max/rms={e[1]:.2e}  seam spectra relL2={es[0]:.2e}")
    if timing:
        chk(LIB.bbt_osm_timing_enable(plan, 1))
        for _ in range(5):
            run()
        ms = (C.c_double * 3)()
        nl = C.c_int64()
        chk(LIB.bbt_osm_timing_read(plan, ms, C.byref(nl)))
        print(f"   passes ms (A,B,C) per 5 runs: {ms[0]:.3f} {ms[1]:.3f} {ms[2]:.3f} launches={nl.value}")
        chk(LIB.bbt_osm_timing_enable(plan, 0))
        chk(LIB.bbt_device_sync())
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        chk(LIB.bbt_device_sync())
        dt = (time.perf_counter() - t0) / 10
        print(f"   wall {dt * 1e3:.3f} ms per {nblk} blocks -> {nblk * hop / dt / 1e6:.1f} Msamples/s")
    chk(LIB.bbt_osm_plan_destroy(plan))
    din.free(), dout.free()
    assert e[0] < 1e-6 and e[1] < 1e-5 and es[0] < 1e-6, (e, es)


def check_pfb(n, S, ntap, nspec, rng):
    x = noise(((nspec + ntap - 1) * n, S), rng)
    taps = rng.normal(size=(ntap, n)).astype(np.float32)
    plan = C.c_void_p()
    chk(LIB.bbt_pfb_plan_create(C.byref(plan), ntap, n, S, taps.ctypes.data_as(C.POINTER(C.c_float))))
    din = Dev.from_host(x)
    dout = Dev(nspec * n * S * 8)
    chk(LIB.bbt_pfb_execute(plan, din.p, dout.p, C.c_int64(nspec), None))
    got = dout.to_host((nspec, n, S), np.complex64)
    xr = x.reshape(nspec + ntap - 1, n, S).astype(np.complex128)
    y = np.zeros((nspec, n, S), np.complex128)
    for t in range(ntap):
        y += xr[t:t + nspec] * taps[t].astype(np.float64)[None, :, None]
    ref = np.fft.fft(y, axis=1)
    e = relerr(got, ref)
    print(f"pfb n={n} S={S} ntap={ntap} nspec={nspec}: relL2={e[0]:.2e} max/rms={e[1]:.2e}")
    chk(LIB.bbt_pfb_plan_destroy(plan))
    din.free(), dout.free()
    assert e[0] < 1e-6 and e[1] < 1e-5, e


def main():
    big = '--big' in sys.argv
    if '--perf' in sys.argv:
        rng = np.random.default_rng(7)
        check_osm(1 << 20, 2, 32, rng, timing=True)
        check_fused(1 << 20, 2, 32, 1024, rng, timing=True)
        return
    if '--large' in sys.argv:
        rng = np.random.default_rng(9)
        check_osm(1 << 21, 2, 2, rng)
        check_osm(1 << 21, 4, 2, rng, C_resp=4)
        check_fused(1 << 21, 2, 2, 512, rng)
        check_osm(1 << 22, 2, 2, rng)
        check_fused(1 << 22, 2, 2, 1024, rng)
        check_osm(1 << 23, 2, 1, rng)
        check_osm(1 << 24, 2, 2, rng, timing=True)
        check_fused(1 << 24, 2, 2, 4096, rng, timing=True)
        print("LARGE OK")
        return
    if '--fused' in sys.argv:
        rng = np.random.default_rng(8)
        for N, ncs in ((1 << 13, (256, 512)), (1 << 14, (256, 1024)), (1 << 16, (256, 2048, 4096)),
                       (1 << 17, (256, 512)), (1 << 18, (1024,)), (1 << 19, (256, 2048))):
            for nc in ncs:
                check_fused(N, 2, 4, nc, rng)
        check_fused(1 << 15, 4, 3, 512, rng, C_resp=2)
        check_fused(1 << 17, 4, 3, 512, rng, C_resp=4, first_spec=3, drop_tail=2)
        check_fused(1 << 20, 2, 3, 1024, rng)
        check_fused(1 << 20, 2, 3, 4096, rng)
        check_fused(1 << 20, 2, 32, 1024, rng, timing=True)
        print("FUSED OK")
        return
    name = C.create_string_buffer(256)
    chk(LIB.bbt_device_name(name, 256))
    print("device:", name.value.decode())
    rng = np.random.default_rng(7)
    for n in (2, 4, 8, 16, 32, 64, 128):
        check_chan(n, 2, 1000 + n, -1, rng)
        check_chan(n, 4, 77, +1, rng)
    for n in (256, 512, 1024, 2048, 4096):
        check_chan(n, 2, 37, -1, rng)
    check_chan(1024, 2, 5, +1, rng)
    check_chan(512, 6, 9, -1, rng)
    for N in (256, 512, 1024, 2048, 4096):
        check_osm(N, 2, 3, rng)
    check_osm(1024, 4, 2, rng, C_resp=4)
    for N in (1 << 13, 1 << 14, 1 << 15, 1 << 16):
        check_osm(N, 2, 3, rng)
    check_osm(1 << 15, 4, 2, rng, C_resp=2)
    for N in (1 << 17, 1 << 18, 1 << 19):
        check_osm(N, 2, 2, rng)
    check_osm(1 << 17, 4, 3, rng, C_resp=4)
    check_pfb(1024, 2, 12, 21, rng)
    check_pfb(256, 4, 4, 50, rng)
    check_pfb(512, 2, 8, 13, rng)
    check_pfb(2048, 2, 16, 7, rng)
    check_pfb(2048, 2, 12, 5, rng)
    check_pfb(4096, 2, 12, 3, rng)
    check_pfb(1024, 6, 5, 9, rng)
    if big:
        check_osm(1 << 20, 2, 3, rng)
        check_osm(1 << 20, 2, 32, rng, timing=True)
    print("ALL OK")


if __name__ == '__main__':
    main()
