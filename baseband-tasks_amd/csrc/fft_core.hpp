// Workgroup-level complex64 FFT for gfx950 (wave64), registers + LDS exchange.
//
// Replaces the numpy.fft.fft / ifft calls of the reference's FFT engine
// (reference: baseband_tasks/fourier/numpy.py:33-39) for power-of-two lengths
// N = 256 * R2, R2 in {1, 2, 4, 8, 16}.
//
// Decomposition N = 16 x 16 x R2, 16 points per thread, T = N / 16 threads
// per transform.  Layout invariant (input AND output, an autosort
// transform): thread `tau`, register `j` holds element  tau + T * j.
//
//   stage 0  radix-16 over a0 (elements T apart), twiddle W_N^{tau*c0}
//   exchange (c0, b)       -> thread (c0, b1), b = R2*a1 + b1
//   stage 1  radix-16 over a1, twiddle W_T^{b1*c1}
//   exchange (c0, c1, b1)  -> thread (c0 + 16 g), c1 = g + R2*u
//   stage 2  radix-R2 over b1 -> c2;  k = c0 + 16 c1 + 256 c2
//
// The LDS address maps (pitches chosen so every ds_read_b64 / ds_write_b64 is
// bank-conflict free) are modelled and checked in tools/fft_model.py.
//
// Register format: the two streams of a pair (the two polarisations) are
// packed ACROSS each other, c2 = {(re_A, re_B), (im_A, im_B)}, so every
// butterfly add / twiddle multiply is one v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 on an aligned register pair and the +-i rotations are pure
// register renaming.  (Packing (re, im) of one stream instead costs a v_mov
// shuffle for every rotation and cross term: 29 % of the VALU stream.)  The
// kernels are VALU-issue-bound per wave before they are HBM-bound, so
// instruction count is what matters.
//
// No MFMA: this is butterfly arithmetic at ~6 flop/byte of on-chip data
// (DESIGN.md).
#pragma once
#if !defined(__HIPCC_RTC__)          // (hipRTC provides the runtime's declarations itself)
#include <hip/hip_runtime.h>
#endif

namespace bbt {

typedef float2 cf;                                        // one complex number (tables)
typedef float v2 __attribute__((ext_vector_type(2)));     // the same component of streams A, B
typedef float f4v __attribute__((ext_vector_type(4)));
struct c2 {                                               // one complex number per stream
    v2 re, im;
};

__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return c2{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return c2{a.re - b.re, a.im - b.im}; }
// both streams times the same complex scalar w (tables hold forward twiddles
// exp(-2 pi i ...); SIGN > 0 multiplies by the conjugate)
template <int SIGN>
__device__ __forceinline__ c2 twmul(c2 a, cf w) {
    if (SIGN < 0) return c2{a.re * w.x - a.im * w.y, a.re * w.y + a.im * w.x};
    return c2{a.re * w.x + a.im * w.y, a.im * w.x - a.re * w.y};
}
// per-stream complex multiply
__device__ __forceinline__ c2 cmul2(c2 a, c2 h) {
    return c2{a.re * h.re - a.im * h.im, a.re * h.im + a.im * h.re};
}
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// The same two products for operands that live in VECTOR registers (table twiddles), written with
// the operand selectors of the packed instructions: a v_pk_* whose scalar factor is the same in
// both lanes takes it from one half of a register pair (op_sel / op_sel_hi), so the twiddle is
// used as loaded.  The compiler does not form these: it copies w.x and w.y into pairs first (two
// v_mov per product, 8 % of the instructions of a generic-length row pass).  Constants (SGPRs,
// literals) stay with twmul / cmul.  Measured on MI355X (round 5): rows of 3402 points 16.77 ->
// 16.39 us per block; in the power-of-two kernels (fft_butterfly_twiddle and the passes of
// bbt_kernels.hpp) nothing -- headline 52.2 / 52.2 / 51.8 without, 51.1 / 51.7 / 51.7 with, 6
// registers more in the row pass -- so those keep the compiler's form.
#ifndef BBT_OPSEL
#define BBT_OPSEL 1
#endif
template <int SIGN>
__device__ __forceinline__ c2 twmul_v(c2 a, cf w) {
#if BBT_OPSEL
    const v2 wv = {w.x, w.y};
    v2 t1, t2, re, im;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t1) : "v"(a.im), "v"(wv));          // im * w.y
    if (SIGN < 0) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
            : "=v"(re) : "v"(a.re), "v"(wv), "v"(t1));                                                    // re * w.x - t1
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t2) : "v"(a.im), "v"(wv));                   // im * w.x
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
            : "=v"(im) : "v"(a.re), "v"(wv), "v"(t2));                                                    // re * w.y + t2
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(re) : "v"(a.re), "v"(wv), "v"(t1));    // re * w.x + t1
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t2) : "v"(a.re), "v"(wv));      // re * w.y
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
            : "=v"(im) : "v"(a.im), "v"(wv), "v"(t2));                                                    // im * w.x - t2
    }
    return c2{re, im};
#else
    return twmul<SIGN>(a, w);
#endif
}
// a * b of two table values as two packed instructions: (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
__device__ __forceinline__ cf cmul_v(cf a, cf b) {
#if BBT_OPSEL
    const v2 av = {a.x, a.y}, bv = {b.x, b.y};
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(av), "v"(bv));   // (-a.y b.y, a.y b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));               // a.x (b.x, b.y) + t
    return make_float2(r.x, r.y);
#else
    return cmul(a, b);
#endif
}
__device__ __forceinline__ c2 splat(cf w) { return c2{v2{w.x, w.x}, v2{w.y, w.y}}; }
__device__ __forceinline__ c2 czero() { return c2{v2{0.f, 0.f}, v2{0.f, 0.f}}; }

template <int SIGN>
__device__ __forceinline__ void radix2(c2& a, c2& b) {
    c2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// (a,b,c,d) -> (X0,X1,X2,X3); the -i / +i rotation of (b - d) is folded into
// the signs of the last four adds.
template <int SIGN>
__device__ __forceinline__ void radix4(c2& a, c2& b, c2& c, c2& d) {
    const c2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), u = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    if (SIGN < 0) {
        b = c2{t1.re + u.im, t1.im - u.re};
        d = c2{t1.re - u.im, t1.im + u.re};
    } else {
        b = c2{t1.re - u.im, t1.im + u.re};
        d = c2{t1.re + u.im, t1.im - u.re};
    }
}

#define BBT_C8 0.70710678118654752440f
#define BBT_C16 0.92387953251128675613f
#define BBT_S16 0.38268343236508977173f

// multiply by W_16^K (forward) or its conjugate
template <int SIGN, int K>
__device__ __forceinline__ c2 mulw16(c2 a) {
    constexpr int k = K & 15;
    if constexpr (k == 0) return a;
    else if constexpr (k == 4) return SIGN < 0 ? c2{a.im, -a.re} : c2{-a.im, a.re};
    else if constexpr (k == 8) return c2{-a.re, -a.im};
    else if constexpr (k == 12) return SIGN < 0 ? c2{-a.im, a.re} : c2{a.im, -a.re};
    else {
        constexpr float cr[16] = {1.f, BBT_C16, BBT_C8, BBT_S16, 0.f, -BBT_S16, -BBT_C8, -BBT_C16,
                                  -1.f, -BBT_C16, -BBT_C8, -BBT_S16, 0.f, BBT_S16, BBT_C8, BBT_C16};
        // forward W = cos - i sin ; sin(2 pi k/16) = cr[(k+12)&15]
        return twmul<SIGN>(a, make_float2(cr[k], -cr[(k + 12) & 15]));
    }
}

// In-place radix-8, natural order out.
template <int SIGN>
__device__ __forceinline__ void radix8(c2 (&v)[8]) {
    radix4<SIGN>(v[0], v[2], v[4], v[6]);  // r=0: t[0][q] at v[2q]
    radix4<SIGN>(v[1], v[3], v[5], v[7]);  // r=1: t[1][q] at v[1+2q]
    v[3] = mulw16<SIGN, 2>(v[3]);          // W8^1
    v[5] = mulw16<SIGN, 4>(v[5]);          // W8^2
    v[7] = mulw16<SIGN, 6>(v[7]);          // W8^3
    radix2<SIGN>(v[0], v[1]);              // X[0], X[4]
    radix2<SIGN>(v[2], v[3]);              // X[1], X[5]
    radix2<SIGN>(v[4], v[5]);              // X[2], X[6]
    radix2<SIGN>(v[6], v[7]);              // X[3], X[7]
    const c2 x1 = v[2], x2 = v[4], x3 = v[6], x4 = v[1], x5 = v[3], x6 = v[5];
    v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
}

// In-place radix-16, natural order out.
template <int SIGN>
__device__ __forceinline__ void radix16(c2 (&v)[16]) {
    // a = r + 4 s ; c = q + 4 p
    radix4<SIGN>(v[0], v[4], v[8], v[12]);   // t[0][q] at v[0+4q]
    radix4<SIGN>(v[1], v[5], v[9], v[13]);
    radix4<SIGN>(v[2], v[6], v[10], v[14]);
    radix4<SIGN>(v[3], v[7], v[11], v[15]);
    // twiddle W16^{r q} on v[r + 4 q]
    v[5] = mulw16<SIGN, 1>(v[5]);
    v[9] = mulw16<SIGN, 2>(v[9]);
    v[13] = mulw16<SIGN, 3>(v[13]);
    v[6] = mulw16<SIGN, 2>(v[6]);
    v[10] = mulw16<SIGN, 4>(v[10]);
    v[14] = mulw16<SIGN, 6>(v[14]);
    v[7] = mulw16<SIGN, 3>(v[7]);
    v[11] = mulw16<SIGN, 6>(v[11]);
    v[15] = mulw16<SIGN, 9>(v[15]);
    // radix-4 over r for each q: v[p + 4q] = X[q + 4p]
    radix4<SIGN>(v[0], v[1], v[2], v[3]);
    radix4<SIGN>(v[4], v[5], v[6], v[7]);
    radix4<SIGN>(v[8], v[9], v[10], v[11]);
    radix4<SIGN>(v[12], v[13], v[14], v[15]);
    // transpose 4x4 to natural order (register renaming)
    c2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int SIGN, int R>
__device__ __forceinline__ void radixR(c2 (&v)[R]) {
    if constexpr (R == 2) radix2<SIGN>(v[0], v[1]);
    else if constexpr (R == 4) radix4<SIGN>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) radix8<SIGN>(v);
    else if constexpr (R == 16) radix16<SIGN>(v);
}

// ---------------------------------------------------------------------------
// Geometry of the 16 x 16 x R2 transform (mirrors tools/fft_model.py).
template <int N>
struct FftGeo {
    static constexpr int R2 = N / 256;
    static constexpr int T = N / 16;
    static_assert(N == 256 * R2 && (R2 == 1 || R2 == 2 || R2 == 4 || R2 == 8 || R2 == 16),
                  "N must be 256..4096, power of two");
    static constexpr int PAD0 = (R2 < 16) ? (T + R2) : (T + 16);
    static constexpr int PB1 = 16 + (R2 == 2 ? 8 : R2 == 4 ? 4 : R2 == 8 ? 2 : R2 == 16 ? 1 : 0);
    static constexpr int pc1() {
        int p = R2 * PB1;
        while (p % 32 != 16) ++p;
        return p;
    }
    static constexpr int PC1 = pc1();
    // exchange area in 8-byte elements (one v2 per point and component)
    static constexpr int LDS_ELEMS = (16 * PAD0 > 16 * PC1) ? 16 * PAD0 : 16 * PC1;
};

// LDS placement of one transform's exchange area.
//   COLMODE == 0 (row mode):  idx(inner) = inner           (lds already offset to this FFT's slot)
//   COLMODE == F (16 or 32):  idx(inner) = inner * F + f   (F transforms interleaved, f fastest)
template <int COLMODE>
__device__ __forceinline__ int lds_idx(int inner, int f) {
    return COLMODE ? inner * COLMODE + f : inner;
}

// One workgroup-cooperative FFT of both streams of a pair, as building blocks
// (so a kernel can issue the table loads of a later stage before the LDS
// exchange of an earlier one) and composed as wg_fft / wg_fft_tail.
//   v[j]  : element tau + T*j of this thread's transform (both streams)
//   lds   : exchange area, LDS_ELEMS v2 per transform (x16 in COLMODE).
//           IMOFF == 0: the real and the imaginary pairs go through it one
//           after the other (4 barriers per exchange); IMOFF > 0: the
//           imaginary pairs use lds + IMOFF (2 barriers, twice the LDS).
//   tau   : thread index within the transform (0..T-1)
//   f     : transform lane within a COLMODE group (0..15), ignored otherwise
//   tw0   : [16][T] forward twiddles W_N^{tau c0};  tw1: [16][R2] W_T^{b1 c1}
// All threads of the workgroup must call the exchanges together (__syncthreads).

// twiddles of stage 0 (index c-1 holds W_N^{tau c}) and stage 1 (W_T^{b1 c})
template <int N>
__device__ __forceinline__ void fft_load_tw0(cf (&w)[15], const cf* __restrict__ tw0, int tau) {
#pragma unroll
    for (int c = 1; c < 16; ++c) w[c - 1] = tw0[c * FftGeo<N>::T + tau];
}
template <int N>
__device__ __forceinline__ void fft_load_tw1(cf (&w)[15], const cf* __restrict__ tw1, int tau) {
    constexpr int R2 = FftGeo<N>::R2;
#pragma unroll
    for (int c = 1; c < 16; ++c) w[c - 1] = (R2 > 1) ? tw1[c * R2 + tau % R2] : make_float2(1.f, 0.f);
}
// The same tables read ONCE per thread: index c - 1 holds the c-th power of the table's c = 1
// entry (W^{tau c} = (W^tau)^c), formed by products at most four roundings deep.  Fourteen
// loads fewer per stage for kernels that wait for their loads rather than for the VALU.
__device__ __forceinline__ void fft_twiddle_powers(cf (&w)[15]) {
#pragma unroll
    for (int c = 2; c < 16; ++c) w[c - 1] = cmul(w[(c + 1) / 2 - 1], w[c / 2 - 1]);
}
template <int N>
__device__ __forceinline__ void fft_load_tw0_pow(cf (&w)[15], const cf* __restrict__ tw0, int tau) {
    w[0] = tw0[FftGeo<N>::T + tau];
    fft_twiddle_powers(w);
}
template <int N>
__device__ __forceinline__ void fft_load_tw1_pow(cf (&w)[15], const cf* __restrict__ tw1, int tau) {
    constexpr int R2 = FftGeo<N>::R2;
    w[0] = (R2 > 1) ? tw1[R2 + tau % R2] : make_float2(1.f, 0.f);
    fft_twiddle_powers(w);
}
// Four loads (c = 1, 2, 4, 8) and eleven products at most three roundings deep: most of the
// saving at about half the rounding of the one-load form (POW == 2).
__device__ __forceinline__ void fft_twiddle_powers4(cf (&w)[15]) {
    w[2] = cmul(w[1], w[0]);                       // 3 = 2 + 1
    w[4] = cmul(w[3], w[0]);                       // 5 = 4 + 1
    w[5] = cmul(w[3], w[1]);                       // 6 = 4 + 2
    w[6] = cmul(w[3], w[2]);                       // 7 = 4 + 3
#pragma unroll
    for (int c = 9; c < 16; ++c) w[c - 1] = cmul(w[7], w[c - 9]);      // 8 + (1 .. 7)
}
template <int N>
__device__ __forceinline__ void fft_load_tw0_pow4(cf (&w)[15], const cf* __restrict__ tw0, int tau) {
    constexpr int T = FftGeo<N>::T;
    w[0] = tw0[T + tau];
    w[1] = tw0[2 * T + tau];
    w[3] = tw0[4 * T + tau];
    w[7] = tw0[8 * T + tau];
    fft_twiddle_powers4(w);
}
template <int N>
__device__ __forceinline__ void fft_load_tw1_pow4(cf (&w)[15], const cf* __restrict__ tw1, int tau) {
    constexpr int R2 = FftGeo<N>::R2;
    const int b1 = tau % R2;
    w[0] = tw1[R2 + b1];
    w[1] = tw1[2 * R2 + b1];
    w[3] = tw1[4 * R2 + b1];
    w[7] = tw1[8 * R2 + b1];
    fft_twiddle_powers4(w);
}
template <int SIGN>
__device__ __forceinline__ void fft_butterfly_twiddle(c2 (&v)[16], const cf (&w)[15]) {
    radix16<SIGN>(v);
#pragma unroll
    for (int c = 1; c < 16; ++c) v[c] = twmul<SIGN>(v[c], w[c - 1]);
}

// exchange 0: (c0, b) -> thread (c0, b1), b = R2 a1 + b1

template <int N, int COLMODE, int IMOFF>
__device__ __forceinline__ void fft_exchange0(c2 (&v)[16], v2* __restrict__ lds, int tau, int f) {
    typedef FftGeo<N> G;
    constexpr int R2 = G::R2;
    const int c0s = tau / R2, b1 = tau % R2;
    v2* __restrict__ lds_im = lds + IMOFF;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) lds[lds_idx<COLMODE>(c * G::PAD0 + tau, f)] = v[c].re;
    if (IMOFF) {
#pragma unroll
        for (int c = 0; c < 16; ++c) lds_im[lds_idx<COLMODE>(c * G::PAD0 + tau, f)] = v[c].im;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a].re = lds[lds_idx<COLMODE>(c0s * G::PAD0 + R2 * a + b1, f)];
    if (!IMOFF) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds[lds_idx<COLMODE>(c * G::PAD0 + tau, f)] = v[c].im;
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a].im = lds_im[lds_idx<COLMODE>(c0s * G::PAD0 + R2 * a + b1, f)];
}

// exchange 1: (c0, c1, b1) -> thread (c0 + 16 g), c1 = g + R2 u ; then the
// radix-R2 stage.  No-op for N == 256.
template <int N, int SIGN, int COLMODE, int IMOFF>
__device__ __forceinline__ void fft_exchange1_stage2(c2 (&v)[16], v2* __restrict__ lds, int tau, int f) {
    typedef FftGeo<N> G;
    constexpr int R2 = G::R2;
    if constexpr (R2 > 1) {
        constexpr int NU = 16 / R2;
        // (COLMODE 4 / 8: a few transforms interleaved, lanes over them first -- the many-stream filter
        // bank -- so 16 lanes of a write are 4 or 2 consecutive b1: their rows must differ by an odd pitch)
        constexpr int PB1 = ((COLMODE == 4 || COLMODE == 8) && G::PB1 % 2 == 0) ? G::PB1 - 1 : G::PB1;
        const int c0s = tau / R2, b1 = tau % R2;
        const int c0r = tau & 15, g = tau >> 4;
        v2* __restrict__ lds_im = lds + IMOFF;
        c2 t[NU][R2];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds[lds_idx<COLMODE>(c * G::PC1 + b1 * PB1 + c0s, f)] = v[c].re;
        if (IMOFF) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                lds_im[lds_idx<COLMODE>(c * G::PC1 + b1 * PB1 + c0s, f)] = v[c].im;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int bb = 0; bb < R2; ++bb)
                t[u][bb].re = lds[lds_idx<COLMODE>((g + R2 * u) * G::PC1 + bb * PB1 + c0r, f)];
        if (!IMOFF) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 16; ++c)
                lds[lds_idx<COLMODE>(c * G::PC1 + b1 * PB1 + c0s, f)] = v[c].im;
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int bb = 0; bb < R2; ++bb)
                t[u][bb].im = lds_im[lds_idx<COLMODE>((g + R2 * u) * G::PC1 + bb * PB1 + c0r, f)];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            radixR<SIGN, R2>(t[u]);
#pragma unroll
            for (int c2i = 0; c2i < R2; ++c2i) v[u + NU * c2i] = t[u][c2i];
        }
    }
}

// wg_fft_tail is everything after stage 0: given y[c0][b] (thread b = tau,
// register c0, already twiddled) it computes the T-point transform over b of
// each of the 16 sequences c0.  On return thread tau2 holds, in register
// u + (16/R2) * c2, output k' = c1 + 16 * c2 (c1 = (tau2 >> 4) + R2 * u) of
// sequence c0 = tau2 & 15 -- for the full transform that is element
// tau2 + T * register (k = c0 + 16 k').  The fused channelizer and the PFB
// enter here with their own stage 0.
template <int N, int SIGN, int COLMODE, int IMOFF = 0, int POW = 0>
__device__ __forceinline__ void wg_fft_tail(c2 (&v)[16], v2* __restrict__ lds, int tau, int f,
                                            const cf* __restrict__ tw1) {
    fft_exchange0<N, COLMODE, IMOFF>(v, lds, tau, f);
    if constexpr (FftGeo<N>::R2 > 1) {
        cf w1[15];
        if constexpr (POW == 1) fft_load_tw1_pow<N>(w1, tw1, tau);
        else if constexpr (POW == 2) fft_load_tw1_pow4<N>(w1, tw1, tau);
        else fft_load_tw1<N>(w1, tw1, tau);
        fft_butterfly_twiddle<SIGN>(v, w1);
    } else {
        radix16<SIGN>(v);
    }
    fft_exchange1_stage2<N, SIGN, COLMODE, IMOFF>(v, lds, tau, f);
}

template <int N, int SIGN, int COLMODE, int IMOFF = 0, int POW = 0>
__device__ __forceinline__ void wg_fft(c2 (&v)[16], v2* __restrict__ lds, int tau, int f,
                                       const cf* __restrict__ tw0, const cf* __restrict__ tw1) {
    cf w0[15];
    if constexpr (POW == 1) fft_load_tw0_pow<N>(w0, tw0, tau);
    else if constexpr (POW == 2) fft_load_tw0_pow4<N>(w0, tw0, tau);
    else fft_load_tw0<N>(w0, tw0, tau);
    fft_butterfly_twiddle<SIGN>(v, w0);
    wg_fft_tail<N, SIGN, COLMODE, IMOFF, POW>(v, lds, tau, f, tw1);
}

// ---------------------------------------------------------------------------
// Memory formats of one complete 2-stream sample (16 bytes):
//   external (numpy complex64 (n, 2)):   re_A im_A re_B im_B
//   internal (work buffers):             re_A re_B im_A im_B   (= c2 as stored)
__device__ __forceinline__ c2 ld_ext(const float2* p) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    return c2{v2{x.x, x.z}, v2{x.y, x.w}};
}
__device__ __forceinline__ void st_ext(float2* p, c2 a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.re.x, a.im.x, a.re.y, a.im.y);
}
// Non-temporal forms for streams that are touched once in whole cache lines
// (S == 2: one complete sample per 16 bytes).  Measured on MI355X: the output
// stores of the last column pass +10 %, Channelize alone 165 -> 185 Gsamples/s,
// because the stream stops displacing the work buffers from L2 / Infinity
// Cache.  Not for S > 2 (pairs share lines: -40 %) nor for re-read inputs (PFB).
__device__ __forceinline__ c2 ld_ext_nt(const float2* p) {
    const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return c2{v2{x.x, x.z}, v2{x.y, x.w}};
}
// (BBT_OUT_ST, experiment switch for the whole-line output stream: 0 non-temporal (default),
// 1 plain, 2 write-through sc1, 3 sc1 nt)
#ifndef BBT_OUT_ST
#define BBT_OUT_ST 0
#endif
__device__ __forceinline__ void st_ext_nt(float2* p, c2 a) {
    const f4v x = {a.re.x, a.im.x, a.re.y, a.im.y};
#if BBT_OUT_ST == 1
    *reinterpret_cast<f4v*>(p) = x;
#elif BBT_OUT_ST == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
#elif BBT_OUT_ST == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
#else
    __builtin_nontemporal_store(x, reinterpret_cast<f4v*>(p));
#endif
}
__device__ __forceinline__ void st_ext(float2* p, c2 a, bool nt) {
    if (nt) st_ext_nt(p, a); else st_ext(p, a);
}
__device__ __forceinline__ c2 ld_int(const float2* p) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    return c2{v2{x.x, x.y}, v2{x.z, x.w}};
}
// Work-buffer stores.  BBT_WORK_ST selects the cache policy: 0 plain (the lines stay dirty in
// the XCD's L2 until evicted or written back at the end of the kernel), 1 non-temporal,
// 2 write-through (sc1: the next pass reads them from other XCDs anyway, and a kernel that
// ends with nothing dirty hands over to the next one sooner), 3 both.  Measured on MI355X
// (headline, round 3, two runs each): plain 46.2 / 46.3, sc1 47.4 / 47.3, nt 42.6 / 42.4,
// sc1 nt 42.6 / 42.6 Gsamples/s -- non-temporal stores lose the Infinity Cache residency of
// the work buffers; write-through is the default.
#ifndef BBT_WORK_ST
#define BBT_WORK_ST 2
#endif
__device__ __forceinline__ void st_int(float2* p, c2 a) {
    const f4v x = {a.re.x, a.re.y, a.im.x, a.im.y};
#if BBT_WORK_ST == 1
    __builtin_nontemporal_store(x, reinterpret_cast<f4v*>(p));
#elif BBT_WORK_ST == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
#elif BBT_WORK_ST == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
#else
    *reinterpret_cast<f4v*>(p) = x;
#endif
}

}  // namespace bbt
