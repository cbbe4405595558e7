"""Numpy model of the 8192- / 16384-point workgroup FFT (csrc/fft_big.hpp), design tooling.

N = 16 x 16 x 16 x L (L = 2, 4), T = N / 16 threads with 16 points each, thread t register j
holding element t + T j on input AND output.  Four register stages, three exchanges through LDS:

  stage A  radix-16 over j -> c0, twiddle W_N^{t c0}
  E1       workgroup-wide: thread (c0, b1) = c0 M + b1 takes b = M a1 + b1 of sequence c0 (M = T / 16)
  stage B  radix-16 over a1 -> c1, twiddle W_T^{b1 c1}
  E2       within the M lanes of a sequence (a wave, or half of one): lane b2 16 + c1 takes
           b1 = L a2 + b2 of row c1
  stage C  radix-16 over a2 -> c2, twiddle W_M^{b2 c2}
  E3       workgroup-wide: thread c0 + 16 c1 + 256 g takes (c2 = g + L u, b2), register u L + b2
  stage D  radix-L over b2 -> c3;  k = c0 + 16 c1 + 256 c2 + 4096 c3 = thread + T (u + (16 / L) c3)

The model follows every value through the address maps, compares with numpy.fft and counts LDS
bank conflicts under the gfx950 rules (ds_write_b64: groups of 16 lanes, 32 banks of 4 bytes;
ds_read_b64: halves of 32 lanes, 64 banks).     python tools/fft_big_model.py
"""
import numpy as np


def geo(n):
    t = n // 16
    m = t // 16
    l = m // 16
    assert n == 4096 * l and l in (2, 4)
    p2 = m + 2
    p1 = 16 * p2
    p3 = 256 * l + 2
    return dict(N=n, T=t, M=m, L=l, P1=p1, P2=p2, P3=p3, LDS=max(16 * p1, 16 * p3))


def conflicts(addr, kind):
    """addr: (threads,) element addresses (8-byte elements) of one wave-wide instruction per wave."""
    worst = 1
    a = np.asarray(addr).reshape(-1, 64)
    for wave in a:
        groups = wave.reshape(4, 16) if kind == 'write' else wave.reshape(2, 32)
        mod = 16 if kind == 'write' else 32
        for g in groups:
            banks = {}
            for x in set(g.tolist()):
                banks.setdefault(x % mod, set()).add(x)
            worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def fft_big(x, sign=-1):
    g = geo(len(x))
    n, t, m, l = g['N'], g['T'], g['M'], g['L']
    p1, p2, p3 = g['P1'], g['P2'], g['P3']
    w = lambda nn, k: np.exp(sign * 2j * np.pi * (np.asarray(k) % nn) / nn)
    dft = (lambda a, ax: np.fft.fft(a, axis=ax)) if sign < 0 else (lambda a, ax: np.fft.ifft(a, axis=ax) * a.shape[ax])
    lds = np.zeros(g['LDS'], complex)
    worst = dict(write=1, read=1)

    def exchange(waddr, values, raddr):
        # waddr / raddr: (threads, 16) addresses; one instruction per register index
        lds[:] = np.nan
        for r in range(16):
            worst['write'] = max(worst['write'], conflicts(waddr[:, r], 'write'))
            assert len(set(waddr[:, r].tolist())) == t
            lds[waddr[:, r]] = values[:, r]
        out = np.empty((t, 16), complex)
        for r in range(16):
            worst['read'] = max(worst['read'], conflicts(raddr[:, r], 'read'))
            out[:, r] = lds[raddr[:, r]]
        assert not np.isnan(out).any()
        return out

    tid = np.arange(t)
    reg = np.arange(16)
    v = x.reshape(16, t).T.copy()                         # v[t, j] = x[t + T j]
    # stage A
    v = dft(v, 1) * w(n, tid[:, None] * reg[None, :])     # [t, c0]
    # E1
    c0, b1 = tid // m, tid % m
    v = exchange(reg[None, :] * p1 + tid[:, None], v, c0[:, None] * p1 + m * reg[None, :] + b1[:, None])   # [(c0, b1), a1]
    # stage B
    v = dft(v, 1) * w(t, b1[:, None] * reg[None, :])      # [(c0, b1), c1]
    # E2 (inside the M lanes of a sequence)
    lane = tid % m
    b2r, c1r = lane // 16, lane % 16
    v = exchange(c0[:, None] * p1 + reg[None, :] * p2 + b1[:, None], v,
                 c0[:, None] * p1 + c1r[:, None] * p2 + l * reg[None, :] + b2r[:, None])                   # [(c0, b2, c1), a2]
    # stage C
    v = dft(v, 1) * w(m, b2r[:, None] * reg[None, :])     # [(c0, b2, c1), c2]
    # E3
    oc0, oc1, og = tid % 16, (tid // 16) % 16, tid // 256
    u, b2 = reg // l, reg % l
    v = exchange(c0[:, None] * p3 + (reg[None, :] * l + b2r[:, None]) * 16 + c1r[:, None], v,
                 oc0[:, None] * p3 + ((og[:, None] + l * u[None, :]) * l + b2[None, :]) * 16 + oc1[:, None])
    # stage D: radix-L over b2 for every u; result register u + (16 / L) c3
    v = dft(v.reshape(t, 16 // l, l), 2)                  # [t, u, c3]
    v = v.transpose(0, 2, 1).reshape(t, 16)               # register c3 (16 / L) + u
    out = np.empty(n, complex)
    out[(tid[:, None] + t * reg[None, :]).ravel()] = v.ravel()
    return out, worst


if __name__ == '__main__':
    rng = np.random.default_rng(1)
    for n in (8192, 16384):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for sign in (-1, +1):
            y, worst = fft_big(x, sign)
            want = np.fft.fft(x) if sign < 0 else np.fft.ifft(x) * n
            err = np.abs(y - want).max() / np.abs(want).max()
            print(n, sign, 'max err', err, 'bank conflicts (ways)', worst, 'LDS elements', geo(n)['LDS'])
            assert err < 1e-12 and worst == dict(write=1, read=1)
