// Workgroup FFT for any length n = 2^a 3^b 5^c 7^d <= 8192 (gfx950, wave64).
//
// The reference sizes its overlap-save blocks with the FFT engine's
// next_fast_len -- for its NumPy engine the smallest 2^a 3^b 5^c 7^d >= n
// (baseband_tasks/fourier/numpy.py:99-126; block rule base.py:750-758) -- so
// the reference's DEFAULT block lengths are not powers of two (6174 in its
// tests, 1 049 760 for Resample at 2^20, 11 059 200 for config 4).  The
// power-of-two core (fft_core.hpp) is the fast path; this file provides the
// same transforms for every length the reference's engine would choose.
//
// Stockham autosort, in place in LDS.  The tile is `ct` interleaved
// transforms (element i of transform c at lds[i * ct + c], 16 bytes each: both
// streams of a pair, re_A re_B im_A im_B).  A stage of radix R with
// Ns = product of the earlier radices:
//
//   for j in [0, n / R):  k = j mod Ns
//       v[r]  = x[j + r n/R] * W_{Ns R}^{r k}          r < R
//       v     = DFT_R(v)
//       x'[(j - k) R + k + r Ns] = v[r]
//
// after the last stage x' is the transform in natural order.  Every thread
// first reads all its butterflies into registers, the workgroup synchronises,
// then it writes: one buffer suffices, so n * ct <= 8192 elements (128 KiB)
// fit one CU.  A thread owns at most MAXB(R) = ceil(BBT_GEN_EPT / R) butterflies per
// stage, so the caller must launch with  n * ct <= BBT_GEN_EPT * blockDim.x
// (BBT_GEN_EPT = 8 elements per thread: 128 registers, four waves per SIMD; measured on
// MI355X against 16 elements per thread -- 202 registers, two waves per SIMD, no spilled
// dword instead of 3-6 --: Channelize(1000) 114 against 61, default-geometry Dedisperse at
// 800 MHz 18.9 against 15.2 Gsamples/s).
//
// Twiddles come from per-stage tables W_{Ns R}^{r k} at [(r - 1) Ns + k] (contiguous in k,
// so neighbouring lanes read neighbouring entries), evaluated in double on the host.  Butterflies for 2, 4, 8 are those of fft_core.hpp; 3, 5
// and 7 use the symmetric form (pairs v[k] +- v[p-k], real coefficient sums).
#pragma once
#if !defined(__HIPCC_RTC__)          // (hipRTC provides the runtime's declarations itself)
#include <hip/hip_runtime.h>
#endif
#include "fft_core.hpp"

namespace bbt {

#define BBT_GEN_MAX_FACTORS 24
#define BBT_GEN_MAX_LEN 8192          // elements of one LDS tile (n * ct)
#ifndef BBT_GEN_EPT
#define BBT_GEN_EPT 8                 // tile elements per thread
#endif
#define BBT_GEN_MAX_THREADS (BBT_GEN_MAX_LEN / BBT_GEN_EPT)
// Stage twiddles W^{r k}, r < R: 1 = one table value W^k per butterfly and its powers by products
// (w[r] = w[ceil(r/2)] w[floor(r/2)], at most four roundings deep); 0 = R - 1 table loads.  The
// loads were what the stages waited for (eight 8-byte loads per radix-9 butterfly against nine
// 16-byte LDS reads): measured on MI355X (round 3) Dedisperse with default arguments at 800 / 600
// MHz 19.3 -> 21.9 / 19.1 -> 20.4 G, Channelize 1000 / 6561 / 8192: 125 -> 139 / 70 -> 82 / 71 -> 76 G,
// 1536 and 3000 unchanged; 4-6 spilled dwords in three of the kernels.
#ifndef BBT_GEN_TW_POWERS
#define BBT_GEN_TW_POWERS 1
#endif
#ifndef BBT_GEN_MAXR
#define BBT_GEN_MAXR 12               // largest radix of a stage (14 .. 16: spilled registers; measured 5 % slower)
#endif
struct GenGeo {
    int n;                            // transform length
    int nfac;                         // number of stages
    int fac[BBT_GEN_MAX_FACTORS];     // radices in {2..10, 12, 14, 15, 16}, product n
    int woff[BBT_GEN_MAX_FACTORS];    // where stage s finds its twiddles in the table (host: get_gen_table)
};

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c2 f4_to_c2(f4 x) { return c2{v2{x.x, x.y}, v2{x.z, x.w}}; }
__device__ __forceinline__ f4 c2_to_f4(c2 a) { return f4{a.re.x, a.re.y, a.im.x, a.im.y}; }

// cos / sin (2 pi m / p) for the odd radices
template <int P> struct OddRoots;
template <> struct OddRoots<3> {
    static constexpr float c[3] = {1.f, -0.5f, -0.5f};
    static constexpr float s[3] = {0.f, 0.86602540378443864676f, -0.86602540378443864676f};
};
template <> struct OddRoots<5> {
    static constexpr float c[5] = {1.f, 0.30901699437494742410f, -0.80901699437494742410f,
                                   -0.80901699437494742410f, 0.30901699437494742410f};
    static constexpr float s[5] = {0.f, 0.95105651629515357212f, 0.58778525229247312917f,
                                   -0.58778525229247312917f, -0.95105651629515357212f};
};
template <> struct OddRoots<7> {
    static constexpr float c[7] = {1.f, 0.62348980185873353053f, -0.22252093395631440429f,
                                   -0.90096886790241912624f, -0.90096886790241912624f,
                                   -0.22252093395631440429f, 0.62348980185873353053f};
    static constexpr float s[7] = {0.f, 0.78183148246802980871f, 0.97492791218182360702f,
                                   0.43388373911755812048f, -0.43388373911755812048f,
                                   -0.97492791218182360702f, -0.78183148246802980871f};
};

// DFT of odd prime length P, natural order in and out.
//   a_k = v_k + v_{P-k}, b_k = v_k - v_{P-k}
//   X_j, X_{P-j} = (v_0 + sum_k cos(2 pi j k / P) a_k)  -+ i (sum_k sin(2 pi j k / P) b_k)   (forward)
template <int SIGN, int P>
__device__ __forceinline__ void radix_odd(c2 (&v)[P]) {
    constexpr int H = (P - 1) / 2;
    c2 a[H], b[H];
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        a[k - 1] = cadd(v[k], v[P - k]);
        b[k - 1] = csub(v[k], v[P - k]);
    }
    const c2 v0 = v[0];
    c2 sum = v0;
#pragma unroll
    for (int k = 0; k < H; ++k) sum = cadd(sum, a[k]);
    v[0] = sum;
#pragma unroll
    for (int j = 1; j <= H; ++j) {
        c2 cc = v0, ss = czero();
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            const float co = OddRoots<P>::c[(j * k) % P], si = OddRoots<P>::s[(j * k) % P];
            cc.re += a[k - 1].re * co;
            cc.im += a[k - 1].im * co;
            ss.re += b[k - 1].re * si;
            ss.im += b[k - 1].im * si;
        }
        // -i ss = (ss.im, -ss.re)
        const c2 lo = c2{cc.re + ss.im, cc.im - ss.re};      // cc - i ss
        const c2 hi = c2{cc.re - ss.im, cc.im + ss.re};      // cc + i ss
        v[j] = SIGN < 0 ? lo : hi;
        v[P - j] = SIGN < 0 ? hi : lo;
    }
}

// cos / sin (2 pi m / n) at compile time (Taylor series in double on [-pi, pi]: 1e-15).
constexpr double cx_angle(int m, int n) {
    m %= n;
    if (m < 0) m += n;
    double x = 6.283185307179586476925286766559 * (double)m / (double)n;
    if (x > 3.14159265358979323846264338327950288) x -= 6.283185307179586476925286766559;
    return x;
}
constexpr double cx_cos(int m, int n) {
    const double x = cx_angle(m, n), x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 26; ++i) {
        term *= -x2 / (double)((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
constexpr double cx_sin(int m, int n) {
    const double x = cx_angle(m, n), x2 = x * x;
    double term = x, sum = x;
    for (int i = 1; i < 26; ++i) {
        term *= -x2 / (double)((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}

template <int SIGN, int R>
__device__ __forceinline__ void gen_butterfly(c2 (&v)[R]);

// DFT of composite length R = A * B on registers (Cooley-Tukey, natural order in and out):
//   X[k1 + A k2] = sum_n2 W_B^{n2 k2} W_R^{n2 k1} sum_n1 x[B n1 + n2] W_A^{n1 k1}
// so that one LDS round trip of the Stockham transform takes a factor of up to 16 instead of
// a single small prime (1260 = 12 x 15 x 7: three stages instead of five).
template <int SIGN, int A, int B>
__device__ __forceinline__ void radix_composite(c2 (&v)[A * B]) {
    constexpr int R = A * B;
    c2 y[B][A];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
        c2 u[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) u[n1] = v[B * n1 + n2];
        gen_butterfly<SIGN, A>(u);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            if (n2 == 0 || k1 == 0) {
                y[n2][k1] = u[k1];
            } else {
                // forward twiddle exp(-2 pi i n2 k1 / R); twmul conjugates it for SIGN > 0
                y[n2][k1] = twmul<SIGN>(u[k1], make_float2((float)cx_cos(n2 * k1, R), (float)-cx_sin(n2 * k1, R)));
            }
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        c2 w[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) w[n2] = y[n2][k1];
        gen_butterfly<SIGN, B>(w);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = w[k2];
    }
}

template <int SIGN, int R>
__device__ __forceinline__ void gen_butterfly(c2 (&v)[R]) {
    if constexpr (R == 2 || R == 4 || R == 8 || R == 16) radixR<SIGN, R>(v);
    else if constexpr (R == 6) radix_composite<SIGN, 2, 3>(v);
    else if constexpr (R == 9) radix_composite<SIGN, 3, 3>(v);
    else if constexpr (R == 10) radix_composite<SIGN, 2, 5>(v);
    else if constexpr (R == 12) radix_composite<SIGN, 4, 3>(v);
    else if constexpr (R == 14) radix_composite<SIGN, 2, 7>(v);
    else if constexpr (R == 15) radix_composite<SIGN, 3, 5>(v);
    else radix_odd<SIGN, R>(v);
}

// One in-place Stockham stage over a tile of `ct` interleaved transforms; ct is a power of
// two and divides the number of threads, so a thread keeps its column and steps through the
// butterflies j = tid / ct + b * (nthr / ct) with additions only (dividing by a run-time ct
// and ns for every butterfly cost the column kernels 250 registers and 100 spilled dwords).
template <int SIGN, int R>
__device__ __forceinline__ void gen_stage(f4* __restrict__ lds, int n, int ns, int ct,
                                          const cf* __restrict__ wn, int tid, int nthr) {
    constexpr int MAXB = (BBT_GEN_EPT + R - 1) / R;
    const int m = n / R;               // butterflies per transform
    const int lg = __ffs(ct) - 1;
    const int col = tid & (ct - 1), j_first = tid >> lg, j_step = nthr >> lg;
    const int k_step = j_step % ns;
    int k = j_first % ns;
    c2 v[MAXB][R];
    int j0s[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = j_first + b * j_step;
        j0s[b] = (j - k) * R + k;
        if (j < m) {
            const f4* src = lds + (j << lg) + col;
            const int mstride = m << lg;
#if BBT_GEN_TW_POWERS
            // one table value W^k per butterfly, its powers by products (depth <= log2 R)
            cf w[R];
            w[1] = wn[k];
#pragma unroll
            for (int r = 2; r < R; ++r) w[r] = cmul(w[(r + 1) / 2], w[r / 2]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c2 t = f4_to_c2(src[r * mstride]);
                if (r > 0 && ns > 1) t = twmul<SIGN>(t, w[r]);
                v[b][r] = t;
            }
#else
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c2 t = f4_to_c2(src[r * mstride]);
                if (r > 0 && ns > 1) t = twmul<SIGN>(t, wn[(r - 1) * ns + k]);
                v[b][r] = t;
            }
#endif
            gen_butterfly<SIGN, R>(v[b]);
        }
        k += k_step;
        k -= k >= ns ? ns : 0;
        // (one butterfly at a time: hoisting every load of the stage to the front costs
        // hundreds of registers)
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = j_first + b * j_step;
        if (j < m) {
            f4* dst = lds + (j0s[b] << lg) + col;
            const int sstride = ns << lg;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r * sstride] = c2_to_f4(v[b][r]);
        }
    }
    __syncthreads();
}

// The whole transform; all threads of the workgroup call it together.  The
// tile must be complete (a __syncthreads() after filling it) on entry; on
// return it holds the transforms in natural order and is synchronised.
template <int SIGN>
__device__ __forceinline__ void gen_fft(f4* __restrict__ lds, const GenGeo& g, int ct,
                                        const cf* __restrict__ wn, int tid, int nthr) {
    int ns = 1;
    const cf* wall = wn;
    for (int s = 0; s < g.nfac; ++s) {
        const int r = g.fac[s];
        wn = wall + g.woff[s];
        switch (r) {
            case 2: gen_stage<SIGN, 2>(lds, g.n, ns, ct, wn, tid, nthr); break;
            case 3: gen_stage<SIGN, 3>(lds, g.n, ns, ct, wn, tid, nthr); break;
            case 4: gen_stage<SIGN, 4>(lds, g.n, ns, ct, wn, tid, nthr); break;
            case 5: gen_stage<SIGN, 5>(lds, g.n, ns, ct, wn, tid, nthr); break;
            case 6: gen_stage<SIGN, 6>(lds, g.n, ns, ct, wn, tid, nthr); break;
            case 7: gen_stage<SIGN, 7>(lds, g.n, ns, ct, wn, tid, nthr); break;
            case 8: gen_stage<SIGN, 8>(lds, g.n, ns, ct, wn, tid, nthr); break;
#if BBT_GEN_MAXR >= 9
            case 9: gen_stage<SIGN, 9>(lds, g.n, ns, ct, wn, tid, nthr); break;
#endif
#if BBT_GEN_MAXR >= 10
            case 10: gen_stage<SIGN, 10>(lds, g.n, ns, ct, wn, tid, nthr); break;
#endif
#if BBT_GEN_MAXR >= 12
            case 12: gen_stage<SIGN, 12>(lds, g.n, ns, ct, wn, tid, nthr); break;
#endif
#if BBT_GEN_MAXR >= 14
            case 14: gen_stage<SIGN, 14>(lds, g.n, ns, ct, wn, tid, nthr); break;
#endif
#if BBT_GEN_MAXR >= 15
            case 15: gen_stage<SIGN, 15>(lds, g.n, ns, ct, wn, tid, nthr); break;
#endif
#if BBT_GEN_MAXR >= 16
            case 16: gen_stage<SIGN, 16>(lds, g.n, ns, ct, wn, tid, nthr); break;
#endif
            default: break;
        }
        ns *= r;
    }
}

// ---------------------------------------------------------------------------
// The same transform with its two ends open: the FIRST stage takes its butterflies from a
// source functor (global memory, with whatever factor the caller folds in) instead of the LDS
// tile, the LAST stage hands its results to a sink functor instead of writing them back --
// the fill pass before and the drain pass after the transform, one LDS round trip with a
// barrier each, disappear.  For a convolution (forward, multiply, inverse) the inverse runs its
// stages in REVERSED order: the forward's last stage leaves butterfly j with elements
// j + r n/R, r < R, in registers, which is exactly what the inverse's first stage of the same
// radix wants -- so the two share one stage (gen_stage_turn): multiply in registers, second
// butterfly, one write.  S stages each way: 2 S + 3 round trips become 2 S - 1.
//
//   Src:  template <int R> void load(int j, int m, c2 (&v)[R])    v[r] = x[j + r m]
//   Mul:  template <int R> void apply(int j, int m, c2 (&v)[R])   v[r] *= h[j + r m]
//   Dst:  template <int R> void store(int j, int m, c2 (&v)[R])   y[j + r m] = v[r]
// (m = n / R; a thread's butterflies are j = tid / ct + b nthr / ct, its column tid % ct.)
template <int R>
struct GenButterflies {                  // which butterflies of a stage of radix R a thread owns
    static constexpr int MAXB = (BBT_GEN_EPT + R - 1) / R;
};

// first stage (ns == 1): source -> butterfly -> LDS at j R + r
template <int SIGN, int R, class Src>
__device__ __forceinline__ void gen_stage_first(f4* __restrict__ lds, int n, int ct, int tid, int nthr,
                                                Src& src) {
    constexpr int MAXB = GenButterflies<R>::MAXB;
    const int m = n / R, lg = __ffs(ct) - 1;
    const int col = tid & (ct - 1), j_first = tid >> lg, j_step = nthr >> lg;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = j_first + b * j_step;
        if (j < m) {
            c2 v[R];
            src.template load<R>(j, m, v);
            gen_butterfly<SIGN, R>(v);
            f4* dst = lds + ((j * R) << lg) + col;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r << lg] = c2_to_f4(v[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
}

// last stage (ns == n / R): LDS -> twiddle -> butterfly -> sink (natural order: j + r ns)
template <int SIGN, int R, class Dst>
__device__ __forceinline__ void gen_stage_last(const f4* __restrict__ lds, int n, int ct,
                                               const cf* __restrict__ wn, int tid, int nthr, Dst& dst) {
    constexpr int MAXB = GenButterflies<R>::MAXB;
    const int m = n / R, lg = __ffs(ct) - 1;
    const int col = tid & (ct - 1), j_first = tid >> lg, j_step = nthr >> lg;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = j_first + b * j_step;
        if (j < m) {
            c2 v[R];
            const f4* src = lds + (j << lg) + col;
            const int mstride = m << lg;
            cf w[R];
#if BBT_GEN_TW_POWERS
            w[1] = wn[j];                                   // (k == j: ns == m)
#pragma unroll
            for (int r = 2; r < R; ++r) w[r] = cmul(w[(r + 1) / 2], w[r / 2]);
#else
#pragma unroll
            for (int r = 1; r < R; ++r) w[r] = wn[(r - 1) * m + j];
#endif
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c2 t = f4_to_c2(src[r * mstride]);
                if (r > 0) t = twmul<SIGN>(t, w[r]);
                v[r] = t;
            }
            gen_butterfly<SIGN, R>(v);
            dst.template store<R>(j, m, v);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the turn of a convolution: last forward stage, multiply, first inverse stage (same radix)
template <int R, class Mul>
__device__ __forceinline__ void gen_stage_turn(f4* __restrict__ lds, int n, int ct,
                                               const cf* __restrict__ wn, int tid, int nthr, Mul& mul) {
    constexpr int MAXB = GenButterflies<R>::MAXB;
    const int m = n / R, lg = __ffs(ct) - 1;
    const int col = tid & (ct - 1), j_first = tid >> lg, j_step = nthr >> lg;
    c2 v[MAXB][R];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = j_first + b * j_step;
        if (j < m) {
            const f4* src = lds + (j << lg) + col;
            const int mstride = m << lg;
            cf w[R];
#if BBT_GEN_TW_POWERS
            w[1] = wn[j];
#pragma unroll
            for (int r = 2; r < R; ++r) w[r] = cmul(w[(r + 1) / 2], w[r / 2]);
#else
#pragma unroll
            for (int r = 1; r < R; ++r) w[r] = wn[(r - 1) * m + j];
#endif
#pragma unroll
            for (int r = 0; r < R; ++r) {
                c2 t = f4_to_c2(src[r * mstride]);
                if (r > 0) t = twmul<-1>(t, w[r]);
                v[b][r] = t;
            }
            gen_butterfly<-1, R>(v[b]);
            mul.template apply<R>(j, m, v[b]);
            gen_butterfly<+1, R>(v[b]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = j_first + b * j_step;
        if (j < m) {
            f4* dst = lds + ((j * R) << lg) + col;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r << lg] = c2_to_f4(v[b][r]);
        }
    }
    __syncthreads();
}

// a single stage (n == R): everything on registers
template <int R, class Src, class Mul, class Dst>
__device__ __forceinline__ void gen_conv_single(int ct, int tid, int nthr, Src& src, Mul& mul, Dst& dst) {
    if ((tid >> (__ffs(ct) - 1)) == 0) {
        c2 v[R];
        src.template load<R>(0, 1, v);
        gen_butterfly<-1, R>(v);
        mul.template apply<R>(0, 1, v);
        gen_butterfly<+1, R>(v);
        dst.template store<R>(0, 1, v);
    }
}

#if BBT_GEN_MAXR >= 12
#define BBT_GEN_RADIX_CASES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(12)
#elif BBT_GEN_MAXR >= 10
#define BBT_GEN_RADIX_CASES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#else
#define BBT_GEN_RADIX_CASES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#endif

// Transform with open ends: src -> ... -> dst (SIGN as gen_fft; g: the stage list, wn its
// tables).  The tile needs no filling; on return nothing of the result is in LDS.
template <int SIGN, class Src, class Dst>
__device__ __forceinline__ void gen_fft_open(f4* __restrict__ lds, const GenGeo& g, int ct,
                                             const cf* __restrict__ wn, int tid, int nthr, Src& src, Dst& dst) {
    const int n = g.n, last = g.nfac - 1;
    if (last == 0) {                     // one stage: source -> butterfly -> sink
        if ((tid >> (__ffs(ct) - 1)) == 0) {
            switch (g.fac[0]) {
#define BBT_X(R_) case R_: { c2 v[R_]; src.template load<R_>(0, 1, v); gen_butterfly<SIGN, R_>(v); \
                             dst.template store<R_>(0, 1, v); } break;
                BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
                default: break;
            }
        }
        return;
    }
    switch (g.fac[0]) {
#define BBT_X(R_) case R_: gen_stage_first<SIGN, R_>(lds, n, ct, tid, nthr, src); break;
        BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
        default: break;
    }
    int ns = g.fac[0];
    for (int s = 1; s < last; ++s) {
        const cf* w = wn + g.woff[s];
        switch (g.fac[s]) {
#define BBT_X(R_) case R_: gen_stage<SIGN, R_>(lds, n, ns, ct, w, tid, nthr); break;
            BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
            default: break;
        }
        ns *= g.fac[s];
    }
    switch (g.fac[last]) {
#define BBT_X(R_) case R_: gen_stage_last<SIGN, R_>(lds, n, ct, wn + g.woff[last], tid, nthr, dst); break;
        BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
        default: break;
    }
}

// Convolution with open ends: src -> forward (stages of g) -> mul -> inverse (stages of gr, the
// SAME radices in reversed order, tables wnr) -> dst.
template <class Src, class Mul, class Dst>
__device__ __forceinline__ void gen_conv_open(f4* __restrict__ lds, const GenGeo& g, const GenGeo& gr, int ct,
                                              const cf* __restrict__ wn, const cf* __restrict__ wnr, int tid,
                                              int nthr, Src& src, Mul& mul, Dst& dst) {
    const int n = g.n, last = g.nfac - 1;
    if (last == 0) {
        switch (g.fac[0]) {
#define BBT_X(R_) case R_: gen_conv_single<R_>(ct, tid, nthr, src, mul, dst); break;
            BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
            default: break;
        }
        return;
    }
    switch (g.fac[0]) {
#define BBT_X(R_) case R_: gen_stage_first<-1, R_>(lds, n, ct, tid, nthr, src); break;
        BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
        default: break;
    }
    int ns = g.fac[0];
    for (int s = 1; s < last; ++s) {
        const cf* w = wn + g.woff[s];
        switch (g.fac[s]) {
#define BBT_X(R_) case R_: gen_stage<-1, R_>(lds, n, ns, ct, w, tid, nthr); break;
            BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
            default: break;
        }
        ns *= g.fac[s];
    }
    switch (g.fac[last]) {               // (== gr.fac[0])
#define BBT_X(R_) case R_: gen_stage_turn<R_>(lds, n, ct, wn + g.woff[last], tid, nthr, mul); break;
        BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
        default: break;
    }
    ns = gr.fac[0];
    for (int s = 1; s < last; ++s) {
        const cf* w = wnr + gr.woff[s];
        switch (gr.fac[s]) {
#define BBT_X(R_) case R_: gen_stage<+1, R_>(lds, n, ns, ct, w, tid, nthr); break;
            BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
            default: break;
        }
        ns *= gr.fac[s];
    }
    switch (gr.fac[last]) {
#define BBT_X(R_) case R_: gen_stage_last<+1, R_>(lds, n, ct, wnr + gr.woff[last], tid, nthr, dst); break;
        BBT_GEN_RADIX_CASES(BBT_X)
#undef BBT_X
        default: break;
    }
}

// W_N^m for m < N up to 2^26 from two tables evaluated in double on the host:
//   lo[i] = W_N^i (i < 4096),  hi[j] = W_N^{4096 j}
__device__ __forceinline__ cf big_twiddle(const cf* __restrict__ lo, const cf* __restrict__ hi, int m) {
    return cmul(hi[m >> 12], lo[m & 4095]);
}

}  // namespace bbt
