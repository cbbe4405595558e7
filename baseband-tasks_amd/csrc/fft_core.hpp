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
// No MFMA: this is butterfly arithmetic at ~6 flop/byte of on-chip data, the
// kernels built on it are HBM-bound.
#pragma once
#include <hip/hip_runtime.h>

namespace bbt {

typedef float2 cf;

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// Tables hold forward twiddles exp(-2 pi i ...); SIGN>0 uses the conjugate.
template <int SIGN>
__device__ __forceinline__ cf twmul(cf a, cf w) {
    return SIGN < 0 ? cmul(a, w) : cmulc(a, w);
}
// multiply by -i (SIGN<0) or +i (SIGN>0)
template <int SIGN>
__device__ __forceinline__ cf mul_mi(cf a) {
    return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

template <int SIGN>
__device__ __forceinline__ void radix2(cf& a, cf& b) {
    cf t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// (a,b,c,d) -> (X0,X1,X2,X3)
template <int SIGN>
__device__ __forceinline__ void radix4(cf& a, cf& b, cf& c, cf& d) {
    cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_mi<SIGN>(csub(b, d));
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = cadd(t1, t3);
    d = csub(t1, t3);
}

#define BBT_C8 0.70710678118654752440f
#define BBT_C16 0.92387953251128675613f
#define BBT_S16 0.38268343236508977173f

// multiply by W_16^K (forward) or its conjugate
template <int SIGN, int K>
__device__ __forceinline__ cf mulw16(cf a) {
    constexpr int k = K & 15;
    if constexpr (k == 0) return a;
    else if constexpr (k == 4) return mul_mi<SIGN>(a);
    else if constexpr (k == 8) return make_float2(-a.x, -a.y);
    else if constexpr (k == 12) return mul_mi<-SIGN>(a);
    else {
        // generic constant twiddle
        constexpr float cr[16] = {1.f, BBT_C16, BBT_C8, BBT_S16, 0.f, -BBT_S16, -BBT_C8, -BBT_C16,
                                  -1.f, -BBT_C16, -BBT_C8, -BBT_S16, 0.f, BBT_S16, BBT_C8, BBT_C16};
        // forward W = cos - i sin ; sin(2 pi k/16) = cr[(k+12)&15]
        constexpr float wr = cr[k];
        constexpr float wi = -cr[(k + 12) & 15];
        return twmul<SIGN>(a, make_float2(wr, wi));
    }
}

// In-place radix-8, natural order out.  v[q + 4p] = sum_a v[a] W8^{a (q+4p)}
template <int SIGN>
__device__ __forceinline__ void radix8(cf (&v)[8]) {
    radix4<SIGN>(v[0], v[2], v[4], v[6]);  // r=0: t[0][q] at v[2q]
    radix4<SIGN>(v[1], v[3], v[5], v[7]);  // r=1: t[1][q] at v[1+2q]
    v[3] = mulw16<SIGN, 2>(v[3]);          // W8^1
    v[5] = mulw16<SIGN, 4>(v[5]);          // W8^2
    v[7] = mulw16<SIGN, 6>(v[7]);          // W8^3
    radix2<SIGN>(v[0], v[1]);              // X[0], X[4]
    radix2<SIGN>(v[2], v[3]);              // X[1], X[5]
    radix2<SIGN>(v[4], v[5]);              // X[2], X[6]
    radix2<SIGN>(v[6], v[7]);              // X[3], X[7]
    // v[2q+p] = X[q+4p]  -> natural
    cf x1 = v[2], x2 = v[4], x3 = v[6], x4 = v[1], x5 = v[3], x6 = v[5];
    v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
}

// In-place radix-16, natural order out.
template <int SIGN>
__device__ __forceinline__ void radix16(cf (&v)[16]) {
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
    // transpose 4x4 to natural order
    cf t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int SIGN, int R>
__device__ __forceinline__ void radixR(cf (&v)[R]) {
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
    static constexpr int LDS_ELEMS = (16 * PAD0 > 16 * PC1) ? 16 * PAD0 : 16 * PC1;
    // twiddle tables (device, cf): tw0[c0 * T + tau] = W_N^{tau c0};
    //                              tw1[c1 * R2 + b1] = W_T^{b1 c1}
    static constexpr int TW0_ELEMS = 16 * T;
    static constexpr int TW1_ELEMS = 16 * R2;
};

// LDS placement of one transform's exchange area.
//   ROWMODE:  idx(inner) = inner            (lds already offset to this FFT's slot)
//   COLMODE:  idx(inner) = inner * 16 + f   (16 transforms interleaved, f fastest)
template <bool COLMODE>
__device__ __forceinline__ int lds_idx(int inner, int f) {
    return COLMODE ? inner * 16 + f : inner;
}

// One workgroup-cooperative FFT over NPL register planes.
//   v[p][j] : plane p, element tau + T*j of this thread's transform
//   lds     : exchange area (LDS_ELEMS cf per transform; x16 in COLMODE)
//   tau     : thread index within the transform (0..T-1)
//   f       : transform lane within a COLMODE group (0..15), ignored otherwise
// All threads of the workgroup must call this together (it uses __syncthreads).
template <int N, int SIGN, int NPL, bool COLMODE>
__device__ __forceinline__ void wg_fft(cf (&v)[NPL][16], cf* __restrict__ lds, int tau, int f,
                                       const cf* __restrict__ tw0, const cf* __restrict__ tw1) {
    typedef FftGeo<N> G;
    constexpr int R2 = G::R2, T = G::T;
    const int c0s = tau / R2;  // stage-1 role
    const int b1 = tau % R2;

    // ---- stage 0
#pragma unroll
    for (int p = 0; p < NPL; ++p) radix16<SIGN>(v[p]);
#pragma unroll
    for (int c = 1; c < 16; ++c) {
        cf w = tw0[c * T + tau];
#pragma unroll
        for (int p = 0; p < NPL; ++p) v[p][c] = twmul<SIGN>(v[p][c], w);
    }
    // ---- exchange 0
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) lds[lds_idx<COLMODE>(c * G::PAD0 + tau, f)] = v[p][c];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) v[p][a] = lds[lds_idx<COLMODE>(c0s * G::PAD0 + R2 * a + b1, f)];
    }
    // ---- stage 1
#pragma unroll
    for (int p = 0; p < NPL; ++p) radix16<SIGN>(v[p]);
    if constexpr (R2 > 1) {
#pragma unroll
        for (int c = 1; c < 16; ++c) {
            cf w = tw1[c * R2 + b1];
#pragma unroll
            for (int p = 0; p < NPL; ++p) v[p][c] = twmul<SIGN>(v[p][c], w);
        }
        // ---- exchange 1 + stage 2
        constexpr int NU = 16 / R2;
        const int c0r = tau & 15;
        const int g = tau >> 4;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 16; ++c)
                lds[lds_idx<COLMODE>(c * G::PC1 + b1 * G::PB1 + c0s, f)] = v[p][c];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                cf t[R2];
#pragma unroll
                for (int bb = 0; bb < R2; ++bb)
                    t[bb] = lds[lds_idx<COLMODE>((g + R2 * u) * G::PC1 + bb * G::PB1 + c0r, f)];
                radixR<SIGN, R2>(t);
#pragma unroll
                for (int c2 = 0; c2 < R2; ++c2) v[p][u + NU * c2] = t[c2];
            }
        }
    }
}

}  // namespace bbt
