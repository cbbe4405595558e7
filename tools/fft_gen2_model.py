"""Model of the register-resident mixed-radix transform of csrc/fft_gen2.hpp (dev tool).

Every value is followed through the stages exactly as the kernel moves it: a thread owns
butterflies j = tj + b T of every stage in registers (slot b R + r), stages hand over through
one LDS buffer whose rows are the already-transformed digits:

    stage s (radix R, Ns = product of the earlier radices, m = n / R), butterfly j = q Ns + k:
        reads   buffer s-1 at  q P[s-1] + k + r (m / Ns) P[s-1]          (rows of Ns, pitch P[s-1])
        twiddle W_{Ns R}^{r k}, DFT_R
        writes  buffer s   at  q P[s] + k + r Ns                          (rows of Ns R, pitch P[s])

so every address is  base(q, k) + r * (a stride that is uniform over the workgroup).  The model
checks the result against numpy.fft, counts LDS bank conflicts of the 8-byte accesses under the
gfx950 rule (32 lanes per pass, 64 banks of 4 bytes) and picks the pitches the way the host
planner does.

    python tools/fft_gen2_model.py [n ...]
"""
import sys

import numpy as np

PMAX = 20           # points a thread holds at most (BBT_G2_PMAX)
MAXB = 4            # butterflies per thread and stage at most (BBT_G2_MAXB)
RADICES = (16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2)


def maxb(r):
    return min(MAXB, max(1, PMAX // r))


def factorise(n):
    """Fewest stages, then fewest threads per transform, large radices first."""
    best = {1: (0, ())}
    for d in sorted(x for x in range(1, n + 1) if n % x == 0):
        if d not in best:
            continue
        for r in RADICES:
            e = d * r
            if n % e:
                continue
            cand = (best[d][0] + 1, tuple(sorted(best[d][1] + (r,), reverse=True)))
            if e not in best or (cand[0], threads(n, cand[1])) < (best[e][0], threads(n, best[e][1])):
                best[e] = cand
    return best[n][1]


def threads(n, fac):
    return max(-(-(n // r) // maxb(r)) for r in fac) if fac else 1


def conflicts(addresses):
    """Passes the LDS needs for one 8-byte access of 32 lanes (1 = conflict-free)."""
    banks = {}
    for a in set(addresses):
        banks.setdefault(a % 32, set()).add(a)
    return max(len(v) for v in banks.values())


def choose_pitches(n, fac, tj, ct=1, pad_max=0):
    """Pitch of the buffer after stage s: rows of Ns R plus the pad (<= pad_max) with the fewest
    passes for the writes of the first 32 lanes (lanes: column fastest, then butterfly).  The
    library pads nothing by default (gen2_host.hpp: measured, padding buys no time and costs
    occupancy)."""
    ns, pitches = 1, []
    for s, r in enumerate(fac[:-1]):
        row = ns * r
        rows = n // row
        best = None
        for pad in range(0, pad_max + 1 if rows > 1 else 1):
            p = row + pad
            worst = 0
            for r_el in (0, r - 1):
                for half in (0, 32):
                    addr = []
                    for lane in range(half, half + 32):
                        col, j = lane % ct, lane // ct
                        q, k = divmod(j, ns)
                        addr.append(((q * p + k + r_el * ns) * ct + col))
                    worst = max(worst, conflicts(addr))
            if best is None or worst < best[0]:
                best = (worst, p)
        pitches.append(best[1])
        ns *= r
    return pitches


def run(n, fac=None, sign=-1, ct=1, verbose=True, tj=None, pitches=None):
    """Follow every value through the stages (`fac`, `tj`, `pitches`: a plan of the library's
    host planner, or this file's own choice) and compare with numpy.fft."""
    fac = tuple(fac or factorise(n))
    assert np.prod(fac) == n
    tj = tj or threads(n, fac)
    pitches = list(pitches) if pitches is not None else choose_pitches(n, fac, tj, ct)
    rng = np.random.default_rng(n)
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    # registers of thread tj_: list of slots
    regs = [[0j] * PMAX for _ in range(tj)]
    lds = None
    ns = 1
    worst_w = worst_r = 1
    for s, r in enumerate(fac):
        m = n // r
        nb = -(-m // tj)
        assert nb <= maxb(r)
        # read (first stage: from the source)
        for t in range(tj):
            for b in range(nb):
                j = t + b * tj
                if j >= m:
                    continue
                q, k = divmod(j, ns)
                for e in range(r):
                    if s == 0:
                        regs[t][b * r + e] = x[j + e * m]
                    else:
                        p = pitches[s - 1]
                        regs[t][b * r + e] = lds[q * p + k + e * (m // ns) * p]
        if s > 0:
            for e in (0, r - 1):
                for base in range(0, min(tj, 64), 32):
                    addr = []
                    for t in range(base, min(base + 32, tj)):
                        q, k = divmod(t, ns)
                        addr.append(q * pitches[s - 1] + k + e * (m // ns) * pitches[s - 1])
                    worst_r = max(worst_r, conflicts(addr))
        # twiddle + butterfly
        for t in range(tj):
            for b in range(nb):
                j = t + b * tj
                if j >= m:
                    continue
                k = j % ns
                v = np.array(regs[t][b * r:(b + 1) * r])
                v = v * np.exp(sign * 2j * np.pi * np.arange(r) * k / (ns * r))
                v = np.fft.fft(v) if sign < 0 else np.fft.ifft(v) * r
                regs[t][b * r:(b + 1) * r] = list(v)
        # write (last stage: to the sink, natural order j + e ns with ns == m)
        if s == len(fac) - 1:
            out = np.zeros(n, complex)
            for t in range(tj):
                for b in range(nb):
                    j = t + b * tj
                    if j < m:
                        for e in range(r):
                            out[j + e * m] = regs[t][b * r + e]
        else:
            p = pitches[s]
            lds = np.zeros((n // (ns * r)) * p, complex)
            for t in range(tj):
                for b in range(nb):
                    j = t + b * tj
                    if j >= m:
                        continue
                    q, k = divmod(j, ns)
                    for e in range(r):
                        lds[q * p + k + e * ns] = regs[t][b * r + e]
            for e in (0, r - 1):
                for base in range(0, min(tj, 64), 32):
                    addr = []
                    for t in range(base, min(base + 32, tj)):
                        q, k = divmod(t, ns)
                        addr.append(q * p + k + e * ns)
                    worst_w = max(worst_w, conflicts(addr))
        ns *= r
    want = np.fft.fft(x) if sign < 0 else np.fft.ifft(x) * n
    err = np.abs(out - want).max() / np.abs(want).max()
    size = max([(n // (np.prod(fac[:s + 1]))) * p for s, p in enumerate(pitches)] + [0])
    if verbose:
        print(f"n {n:5d} = {' x '.join(map(str, fac)):18s} threads {tj:4d} ({n / tj:5.1f} points each)  "
              f"pitches {pitches}  lds {size * 8 / 1024:5.1f} KiB  passes w {worst_w} r {worst_r}  err {err:.1e}")
    assert err < 1e-12
    return fac, tj, pitches


if __name__ == '__main__':
    lengths = [int(a) for a in sys.argv[1:]] or [3402, 2940, 8100, 1000, 3000, 1536, 6561, 490, 567, 486, 6174,
                                                 7938, 1890, 1764, 1470, 8192, 3125, 2401, 4374, 14, 30]
    for n in lengths:
        run(n)
        run(n, sign=+1, fac=tuple(reversed(factorise(n))), verbose=False)
