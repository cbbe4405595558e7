"""Numpy model of the workgroup FFT used by the HIP kernels (csrc/fft_core.hpp).

This is design tooling, not product code: it mirrors, thread by thread and
register by register, the index algebra of the 16 x 16 x R2 register FFT
(R2 in {1, 2, 4, 8, 16}; N = 256 * R2; T = N / 16 threads per transform,
16 points per thread) so the maps, twiddle tables and LDS paddings can be
checked against ``numpy.fft`` and for LDS bank conflicts before they are
transcribed to HIP.

Layout invariant: thread ``t`` register ``j`` holds element ``t + T * j`` of
the sequence, both on input and on output (an autosort transform).
"""
import numpy as np


def geometry(n):
    r2 = n // 256
    assert n == 256 * r2 and r2 in (1, 2, 4, 8, 16)
    t = n // 16
    return r2, t


# ---- LDS address maps (in units of one complex element) -------------------
def pad0(r2):
    """Row pitch of exchange 0 (c0-major rows of T elements)."""
    t = 16 * r2
    # rows are read with lanes (c0, b1): want pitch = R2 mod 32 so that
    # consecutive c0 rows land R2 elements apart in the bank row.
    return t + r2 if r2 < 16 else t + 16


def ex0_write(n, tau, c0):
    r2, t = geometry(n)
    return c0 * pad0(r2) + tau


def ex0_read(n, tau1, a1):
    """Stage-1 thread tau1 = c0 * R2 + b1 reads b = R2 * a1 + b1 of row c0."""
    r2, t = geometry(n)
    c0, b1 = tau1 // r2, tau1 % r2
    return c0 * pad0(r2) + r2 * a1 + b1


def pitch_b1(r2):
    return 16 + (4 if r2 == 4 else (2 if r2 == 8 else (1 if r2 == 16 else 8 if r2 == 2 else 0)))


def pitch_c1(r2):
    p = r2 * pitch_b1(r2)
    # want pitch_c1 = 16 mod 32 (readers: 16 lanes c0 then next g)
    while p % 32 != 16:
        p += 1
    return p


def ex1_addr(n, c0, c1, b1):
    r2, t = geometry(n)
    return c1 * pitch_c1(r2) + b1 * pitch_b1(r2) + c0


def lds_elems(n):
    r2, t = geometry(n)
    return max(16 * pad0(r2), 16 * pitch_c1(r2))


# ---- the transform ---------------------------------------------------------
def w(n, k):
    return np.exp(-2j * np.pi * (np.asarray(k) % n) / n)


def fft_model(x, sign=-1):
    """x: (N,) complex. Returns (N,) in natural order, via the thread model."""
    n = x.shape[0]
    r2, t = geometry(n)
    conj = (lambda z: z) if sign < 0 else np.conj
    # registers: reg[tau, j] = x[tau + t*j]
    reg = x.reshape(16, t).T.copy()          # [tau, a0]
    tau = np.arange(t)
    # stage 0: radix-16 over a0 -> c0 ; twiddle W_N^{tau*c0}
    y = np.fft.fft(reg, axis=1) if sign < 0 else np.fft.ifft(reg, axis=1) * 16
    c = np.arange(16)
    y = y * conj(w(n, tau[:, None] * c[None, :]))
    # exchange 0
    lds = np.full(lds_elems(n), np.nan, complex)
    for c0 in range(16):
        lds[ex0_write(n, tau, c0)] = y[:, c0]
    reg1 = np.empty((t, 16), complex)
    for a1 in range(16):
        reg1[:, a1] = lds[ex0_read(n, tau, a1)]
    # stage 1: thread tau1 = c0*R2 + b1 ; radix-16 over a1 -> c1
    z = np.fft.fft(reg1, axis=1) if sign < 0 else np.fft.ifft(reg1, axis=1) * 16
    b1 = tau % r2
    c0 = tau // r2
    z = z * conj(w(t, b1[:, None] * c[None, :]))   # W_T^{b1*c1}, T = 16*R2
    if r2 == 1:
        # thread c0 holds X[c0 + 16*c1] : already (tau + T*j)
        return z.T.reshape(-1)
    # exchange 1
    lds = np.full(lds_elems(n), np.nan, complex)
    for c1 in range(16):
        lds[ex1_addr(n, c0, c1, b1)] = z[:, c1]
    # stage 2 thread tau2 = c0 + 16*g handles c1 = g + R2*u, u < 16/R2
    c0r = tau % 16
    g = tau // 16
    nu = 16 // r2
    out = np.empty((t, 16), complex)
    for u in range(nu):
        c1 = g + r2 * u
        v = np.empty((t, r2), complex)
        for bb in range(r2):
            v[:, bb] = lds[ex1_addr(n, c0r, c1, bb)]
        vv = np.fft.fft(v, axis=1) if sign < 0 else np.fft.ifft(v, axis=1) * r2
        for c2 in range(r2):
            out[:, u + nu * c2] = vv[:, c2]
    return out.T.reshape(-1)


# ---- bank-conflict model ---------------------------------------------------
def conflicts(addrs, kind):
    """addrs: (64,) element (8-byte) addresses for one wave instruction.

    kind 'r64': ds_read_b64 -> 2 groups of 32 lanes, 64 dword banks.
    kind 'w64': ds_write_b64 -> 4 groups of 16 lanes, 32 dword banks.
    Returns the worst multiplicity (1 = conflict free).
    """
    worst = 1
    if kind == 'r64':
        groups, nb = [range(0, 32), range(32, 64)], 32
    else:
        groups, nb = [range(i, i + 16) for i in range(0, 64, 16)], 16
    for grp in groups:
        a = np.unique(np.asarray(addrs)[list(grp)])
        bank = a % nb
        worst = max(worst, np.bincount(bank, minlength=nb).max())
    return worst


def report(n):
    r2, t = geometry(n)
    print(f"N={n} R2={r2} T={t} lds_elems={lds_elems(n)} "
          f"({lds_elems(n) * 8} B) pad0={pad0(r2)} pb1={pitch_b1(r2)} pc1={pitch_c1(r2)}")
    nw = max(1, t // 64)
    lanes = np.arange(64)
    worst = {}
    for wv in range(nw):
        tau = (wv * 64 + lanes) % t
        for c0 in range(16):
            worst['ex0w'] = max(worst.get('ex0w', 1), conflicts(ex0_write(n, tau, c0), 'w64'))
        for a1 in range(16):
            worst['ex0r'] = max(worst.get('ex0r', 1), conflicts(ex0_read(n, tau, a1), 'r64'))
        if r2 > 1:
            for c1 in range(16):
                worst['ex1w'] = max(worst.get('ex1w', 1),
                                    conflicts(ex1_addr(n, tau // r2, c1, tau % r2), 'w64'))
            for u in range(16 // r2):
                for bb in range(r2):
                    worst['ex1r'] = max(worst.get('ex1r', 1),
                                        conflicts(ex1_addr(n, tau % 16, tau // 16 + r2 * u, bb), 'r64'))
    print("  worst bank multiplicity:", worst)


if __name__ == '__main__':
    rng = np.random.default_rng(1)
    for n in (256, 512, 1024, 2048, 4096):
        x = rng.normal(size=n) + 1j * rng.normal(size=n)
        for sign in (-1, +1):
            ref = np.fft.fft(x) if sign < 0 else np.fft.ifft(x) * n
            got = fft_model(x, sign)
            err = np.abs(got - ref).max()
            assert err < 1e-9, (n, sign, err)
        report(n)
    print("model OK")
