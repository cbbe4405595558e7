// Register-resident mixed-radix workgroup FFT for any length n = 2^a 3^b 5^c 7^d <= 8192
// (gfx950, wave64), specialised at compile time on its geometry: the engine behind the
// reference's DEFAULT block lengths.
//
// The reference sizes overlap-save blocks with the FFT engine's next_fast_len -- the smallest
// 2^a 3^b 5^c 7^d >= n for its NumPy engine (baseband_tasks/fourier/numpy.py:99-126; block rule
// base.py:750-758) -- so with default arguments blocks are 1 666 980 (800 MHz), 3 936 600
// (600 MHz), 1 049 760 (Resample) ... samples, not powers of two.  fft_generic.hpp (rounds 2-4)
// takes the stage list at run time: the data lives in LDS (16 bytes per point, 8 points per
// thread, radices <= 12, one barrier-separated load-butterfly-store chain per stage, every index
// computed at run time), and it runs latency-bound at 0.12 of the roofline.  A run-time radix
// switch with the data in REGISTERS was tried first (round 5): every case has to carry all of a
// thread's points through, and the compiler spilled hundreds of registers.  So this engine takes
// its geometry as a compile-time trait `G` -- the library specialises it per length when a plan
// is made (hipRTC, rtc.hpp), as rocFFT does -- and has the structure of the power-of-two core
// (fft_core.hpp):
//
//   * a thread keeps up to ~20 points in registers (slot b R + r: element r of its b-th
//     butterfly of the stage), radices up to 16: three stages up to 4096 points, four up to 8192;
//   * stages hand over through ONE LDS buffer of 8 bytes per point -- the real pairs, then the
//     imaginary pairs (c2 = {(re_A, re_B), (im_A, im_B)}: both streams of a pair packed across
//     each other) -- so a 3402-point transform takes 27 KiB and several 256-thread workgroups
//     share a CU;
//   * the buffer between two stages is laid out by transformed digits, rows of Ns R with a
//     padded pitch (stage s, radix R, Ns = product of the earlier radices, m = n / R,
//     butterfly j = q Ns + k):
//         reads   (q P[s-1] + k) + r (m / Ns) P[s-1]
//         writes  (q P[s]   + k) + r Ns
//     -- every address is a per-butterfly base plus a compile-time multiple of r (an immediate
//     offset of the ds instruction), and the pitches are chosen by the host planner
//     (gen2_host.hpp, modelled in tools/fft_gen2_model.py) so that the strided side of the
//     first exchanges stays (nearly) conflict-free;
//   * a stage's one table twiddle per butterfly W_{Ns R}^k is fetched before the exchange that
//     precedes it, its powers are formed by products (<= 4 roundings deep).
//
// As in fft_generic.hpp the transform has open ends -- the first stage takes its butterflies
// from a source functor (global memory), the last hands them to a sink -- and a convolution
// runs its inverse with the stages in reversed order so that the forward's last and the
// inverse's first stage (same radix, same elements j + r n/R in registers) are one `turn`.
//
// The trait:
//   struct G { static constexpr int n, nfac, tj (threads per transform), ct (interleaved
//              transforms per workgroup, a power of two);
//              static constexpr int fac[8], pitch[8]; };
#pragma once
#if !defined(__HIPCC_RTC__)          // (hipRTC provides the runtime's declarations itself)
#include <hip/hip_runtime.h>
#endif
#include "fft_generic.hpp"

namespace bbt {

#define BBT_G2_MAXS 8                  // stages (5^5 = 3125 needs five)

// ---- compile-time geometry of stage S of G ---------------------------------------------------
template <class G, int S>
struct G2Stage {
    static constexpr int R = G::fac[S];
    static constexpr int M = G::n / R;                       // butterflies
    static constexpr int ns_() {
        int p = 1;
        for (int s = 0; s < S; ++s) p *= G::fac[s];
        return p;
    }
    static constexpr int NS = ns_();
    static constexpr int B = (M + G::tj - 1) / G::tj;        // butterflies per thread
    static constexpr int woff_() {                           // table of stage s: W_{ns R}^{r k} at woff + (r - 1) ns + k
        int total = 0, ns = 1;
        for (int s = 0; s < S; ++s) {
            if (s > 0) total += (G::fac[s] - 1) * ns;
            ns *= G::fac[s];
        }
        return total;
    }
    static constexpr int WOFF = woff_();
    static constexpr int POUT = G::pitch[S];                 // pitch of the buffer this stage writes
    static constexpr int PIN = S > 0 ? G::pitch[S > 0 ? S - 1 : 0] : 1;
    static constexpr int RSTRIDE = (M / NS) * PIN * G::ct;   // between the elements a butterfly reads
    static constexpr int WSTRIDE = NS * G::ct;               // ... and writes
};
template <class G>
struct G2Info {
    static constexpr int vmax_() {
        int v = 1;
        for (int s = 0; s < G::nfac; ++s) {
            const int m = G::n / G::fac[s], b = (m + G::tj - 1) / G::tj;
            if (b * G::fac[s] > v) v = b * G::fac[s];
        }
        return v;
    }
    static constexpr int V = vmax_();                        // register slots a thread needs
    static constexpr int bmax_() {
        int v = 1;
        for (int s = 0; s < G::nfac; ++s) {
            const int m = G::n / G::fac[s], b = (m + G::tj - 1) / G::tj;
            if (b > v) v = b;
        }
        return v;
    }
    static constexpr int BMAX = bmax_();
    static constexpr int lds_() {
        int e = 0, ns = 1;
        for (int s = 0; s + 1 < G::nfac; ++s) {
            ns *= G::fac[s];
            const int rows = G::n / ns;
            if (rows * G::pitch[s] > e) e = rows * G::pitch[s];
        }
        return e * G::ct;
    }
    static constexpr int LDS = lds_() > 0 ? lds_() : 1;      // exchange area, 8-byte elements
    static constexpr int THREADS = (G::tj * G::ct + 63) / 64 * 64;
};

struct G2Ctx {
    v2* lds;
    int tj;                            // this thread's index within its transform
    int col;                           // column of the tile
    bool live;                         // tj < G::tj (the workgroup is whole waves)
};
template <class G>
__device__ __forceinline__ G2Ctx g2_ctx(v2* lds, int tid) {
    G2Ctx c;
    c.lds = lds;
    c.col = tid & (G::ct - 1);
    c.tj = tid / G::ct;
    c.live = c.tj < G::tj;
    return c;
}
// is butterfly b of stage S one this thread has?  (only the last round can run past the end)
template <class G, int S>
__device__ __forceinline__ bool g2_valid(const G2Ctx& c, int b) {
    typedef G2Stage<G, S> St;
    if ((b + 1) * G::tj <= St::M) return c.live;
    return c.live && c.tj + b * G::tj < St::M;
}

template <int R>
__device__ __forceinline__ void g2_twiddle_powers(cf (&w)[R > 1 ? R : 2]) {
#pragma unroll
    for (int r = 2; r < R; ++r) w[r] = cmul_v(w[(r + 1) / 2], w[r / 2]);
}

// ---- the exchange: stage SP's results -> stage SP + 1's butterflies -------------------------
template <class G, int SP, bool IM, int V>
__device__ __forceinline__ void g2_write(const G2Ctx& c, const c2 (&v)[V]) {
    typedef G2Stage<G, SP> St;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {
        const int j = c.tj + b * G::tj;
        const int q = j / St::NS, k = j - q * St::NS;
        const int base = (q * St::POUT + k) * G::ct + c.col;
        if (g2_valid<G, SP>(c, b)) {
#pragma unroll
            for (int r = 0; r < St::R; ++r)
                c.lds[base + r * St::WSTRIDE] = IM ? v[b * St::R + r].im : v[b * St::R + r].re;
        }
    }
}
template <class G, int S, bool IM, int V>
__device__ __forceinline__ void g2_read(const G2Ctx& c, c2 (&v)[V]) {
    typedef G2Stage<G, S> St;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {
        const int j = c.tj + b * G::tj;
        const int q = j / St::NS, k = j - q * St::NS;
        // (a butterfly past the end reads slot 0 onwards: defined, unused)
        const int base = g2_valid<G, S>(c, b) ? (q * St::PIN + k) * G::ct + c.col : 0;
#pragma unroll
        for (int r = 0; r < St::R; ++r) {
            const v2 x = c.lds[base + r * St::RSTRIDE];
            if (IM) v[b * St::R + r].im = x; else v[b * St::R + r].re = x;
        }
    }
}
// the table twiddles W^k of stage S's butterflies (issued before the exchange, used after it)
template <class G, int S, int BM>
__device__ __forceinline__ void g2_fetch_tw(const G2Ctx& c, const cf* __restrict__ wn, cf (&tw1)[BM]) {
    typedef G2Stage<G, S> St;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {
        const int j = c.tj + b * G::tj;
        const int k = j % St::NS;
        tw1[b] = wn[St::WOFF + (g2_valid<G, S>(c, b) ? k : 0)];
    }
}
template <class G, int S, int V, int BM>
__device__ __forceinline__ void g2_exchange(const G2Ctx& c, const cf* __restrict__ wn, c2 (&v)[V], cf (&tw1)[BM]) {
    g2_fetch_tw<G, S>(c, wn, tw1);
    __syncthreads();
    g2_write<G, S - 1, false>(c, v);
    __syncthreads();
    g2_read<G, S, false>(c, v);
    __syncthreads();
    g2_write<G, S - 1, true>(c, v);
    __syncthreads();
    g2_read<G, S, true>(c, v);
}

// twiddle and butterfly of stage S (S > 0) on the registers
template <class G, int SIGN, int S, int V, int BM>
__device__ __forceinline__ void g2_butterflies(c2 (&v)[V], const cf (&tw1)[BM]) {
    typedef G2Stage<G, S> St;
    constexpr int R = St::R;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {
        cf tw[R > 1 ? R : 2];
        tw[1] = tw1[b];
        g2_twiddle_powers<R>(tw);
        c2 u[R];
        u[0] = v[b * R];
#pragma unroll
        for (int r = 1; r < R; ++r) u[r] = twmul_v<SIGN>(v[b * R + r], tw[r]);
        gen_butterfly<SIGN, R>(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[b * R + r] = u[r];
    }
}

// first stage: source -> butterfly (no twiddles: Ns == 1), loads and arithmetic apart
template <class G, int V, class Src>
__device__ __forceinline__ void g2_load(const G2Ctx& c, c2 (&v)[V], Src& src) {
    typedef G2Stage<G, 0> St;
    constexpr int R = St::R;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {              // (all loads of the thread in flight together)
        c2 u[R];
        if (g2_valid<G, 0>(c, b)) {
            src.template load<R>(c.tj + b * G::tj, St::M, u);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = czero();
        }
#pragma unroll
        for (int r = 0; r < R; ++r) v[b * R + r] = u[r];
    }
}
template <class G, int SIGN, int V>
__device__ __forceinline__ void g2_first_butterflies(c2 (&v)[V]) {
    typedef G2Stage<G, 0> St;
    constexpr int R = St::R;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {
        c2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[b * R + r];
        gen_butterfly<SIGN, R>(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[b * R + r] = u[r];
    }
}
template <class G, int SIGN, int V, class Src>
__device__ __forceinline__ void g2_first(const G2Ctx& c, c2 (&v)[V], Src& src) {
    g2_load<G>(c, v, src);
    g2_first_butterflies<G, SIGN>(v);
}
// last stage's results -> sink (natural order: elements j + r m)
template <class G, int S, int V, class Dst>
__device__ __forceinline__ void g2_sink(const G2Ctx& c, c2 (&v)[V], Dst& dst) {
    typedef G2Stage<G, S> St;
    constexpr int R = St::R;
#pragma unroll
    for (int b = 0; b < St::B; ++b) {
        c2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[b * R + r];
        if (g2_valid<G, S>(c, b)) dst.template store<R>(c.tj + b * G::tj, St::M, u);
    }
}

// stages S .. nfac - 1 of a transform whose stage S - 1 results are in v
template <class G, int SIGN, int S, int V, int BM, class Dst>
__device__ __forceinline__ void g2_rest(const G2Ctx& c, const cf* __restrict__ wn, c2 (&v)[V], cf (&tw1)[BM], Dst& dst) {
    if constexpr (S < G::nfac) {
        g2_exchange<G, S>(c, wn, v, tw1);
        g2_butterflies<G, SIGN, S>(v, tw1);
        g2_rest<G, SIGN, S + 1>(c, wn, v, tw1, dst);
    } else {
        g2_sink<G, G::nfac - 1>(c, v, dst);
    }
}

// Transform with open ends: src -> ... -> dst.  `lds`: G2Info<G>::LDS elements of 8 bytes; all
// G2Info<G>::THREADS threads of the workgroup call it together.
template <class G, int SIGN, class Src, class Dst>
__device__ __forceinline__ void g2_fft_open(v2* __restrict__ lds, const cf* __restrict__ wn, int tid,
                                            Src& src, Dst& dst) {
    const G2Ctx c = g2_ctx<G>(lds, tid);
    c2 v[G2Info<G>::V];
    cf tw1[G2Info<G>::BMAX];
    g2_first<G, SIGN>(c, v, src);
    g2_rest<G, SIGN, 1>(c, wn, v, tw1, dst);
}

// forward stages S .. nfac - 1, then the multiply: the turn
template <class G, int S, int V, int BM, class Mul>
__device__ __forceinline__ void g2_forward_rest(const G2Ctx& c, const cf* __restrict__ wn, c2 (&v)[V], cf (&tw1)[BM],
                                                Mul& mul) {
    if constexpr (S < G::nfac) {
        g2_exchange<G, S>(c, wn, v, tw1);
        g2_butterflies<G, -1, S>(v, tw1);
        g2_forward_rest<G, S + 1>(c, wn, v, tw1, mul);
    } else {
        typedef G2Stage<G, G::nfac - 1> St;
        constexpr int R = St::R;
#pragma unroll
        for (int b = 0; b < St::B; ++b) {
            c2 u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = v[b * R + r];
            if (g2_valid<G, G::nfac - 1>(c, b)) mul.template apply<R>(c.tj + b * G::tj, St::M, u);
#pragma unroll
            for (int r = 0; r < R; ++r) v[b * R + r] = u[r];
        }
    }
}

// Convolution with open ends: src -> forward (stages of G) -> mul -> inverse (stages of GR: the
// SAME radices in reversed order, same tj and ct; tables wnr) -> dst.
template <class G, class GR, class Src, class Mul, class Dst>
__device__ __forceinline__ void g2_conv_open(v2* __restrict__ lds, const cf* __restrict__ wn,
                                             const cf* __restrict__ wnr, int tid, Src& src, Mul& mul, Dst& dst) {
    static_assert(G::fac[G::nfac - 1] == GR::fac[0] && G::tj == GR::tj && G::ct == GR::ct, "GR must be G reversed");
    static_assert(G2Info<G>::V == G2Info<GR>::V, "reversed stages need the same registers");
    const G2Ctx c = g2_ctx<G>(lds, tid);
    c2 v[G2Info<G>::V];
    cf tw1[G2Info<G>::BMAX];
    g2_first<G, -1>(c, v, src);
    g2_forward_rest<G, 1>(c, wn, v, tw1, mul);
    // the inverse's first stage: the same radix on the same registers
    {
        typedef G2Stage<GR, 0> St;
        constexpr int R = St::R;
#pragma unroll
        for (int b = 0; b < St::B; ++b) {
            c2 u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = v[b * R + r];
            gen_butterfly<+1, R>(u);
#pragma unroll
            for (int r = 0; r < R; ++r) v[b * R + r] = u[r];
        }
    }
    g2_rest<GR, +1, 1>(c, wnr, v, tw1, dst);
}

}  // namespace bbt
