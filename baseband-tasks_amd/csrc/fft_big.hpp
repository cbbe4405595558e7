// Workgroup FFTs of 8192 and 16384 points for gfx950: N = 16 x 16 x 16 x L (L = 2, 4), T = N / 16
// threads of 16 points each (512 or 1024: one workgroup per CU at 16384), the register format
// and the autosort contract of fft_core.hpp -- thread t, register j holds element t + T j of both
// streams of a pair, on input and on output.
//
// Replaces numpy.fft.fft / ifft of the reference's FFT engine (baseband_tasks/fourier/numpy.py:33-39)
// for Channelize with 8192 / 16384 channels (reference channelize.py:73-74).
//
//   stage A  radix-16 over j -> c0, twiddle W_N^{t c0}
//   E1       workgroup-wide: thread (c0, b1) = c0 M + b1 takes b = M a1 + b1 of sequence c0,  M = T / 16
//   stage B  radix-16 over a1 -> c1, twiddle W_T^{b1 c1}
//   E2       inside the M lanes of a sequence -- one wave (M = 64) or half of one (M = 32): no
//            workgroup barrier -- lane b2 16 + c1 takes b1 = L a2 + b2 of row c1
//   stage C  radix-16 over a2 -> c2, twiddle W_M^{b2 c2}
//   E3       workgroup-wide: thread c0 + 16 c1 + 256 g takes (c2 = g + L u, b2) into register u L + b2
//   stage D  radix-L over b2 -> c3:  k = c0 + 16 c1 + 256 c2 + 4096 c3 = thread + T (u + (16 / L) c3)
//
// One exchange more than the 16 x 16 x R2 transform of fft_core.hpp, and it costs no barrier: after
// E1 a whole sequence sits in one wave.  Real and imaginary parts go through the exchange area one
// after the other (8 bytes per point: 68 / 132 KiB).  The address maps are modelled, compared with
// numpy.fft and checked to be free of bank conflicts (ds_write_b64: 16-lane groups on 32 banks,
// ds_read_b64: 32-lane halves on 64) in tools/fft_big_model.py.
#pragma once
#include "fft_core.hpp"

namespace bbt {

template <int N>
struct BigGeo {
    static constexpr int T = N / 16, M = T / 16, L = M / 16;
    static_assert(N == 4096 * L && (L == 2 || L == 4), "N must be 8192 or 16384");
    static constexpr int P2 = M + 2;          // row pitch of E2 (rows c1 of a sequence)
    static constexpr int P1 = 16 * P2;        // pitch of a sequence: its E1 row and its E2 area
    static constexpr int P3 = 256 * L + 2;    // row pitch of E3 (rows c0)
    static constexpr int LDS_ELEMS = (P1 > P3 ? P1 : P3) * 16;
    // twiddle table (host: get_big_table): the powers c = 1, 2, 4, 8 of each stage's factors
    //   [0, 4 T)              W_N^{t c}
    //   [4 T, 4 T + 4 M)      W_T^{b1 c}
    //   [4 T + 4 M, + 4 L)    W_M^{b2 c}
    static constexpr int TW_B = 4 * T, TW_C = 4 * T + 4 * M, TW_ELEMS = 4 * (T + M + L);
};

__device__ __forceinline__ void big_load_tw(cf (&w)[15], const cf* __restrict__ tab, int n, int i) {
    w[0] = tab[i];
    w[1] = tab[n + i];
    w[3] = tab[2 * n + i];
    w[7] = tab[3 * n + i];
    fft_twiddle_powers4(w);
}

// One exchange: register r goes to wb + ws r and comes from rb + rs r.  WG: every wave reads what
// other waves wrote (barriers); otherwise a wave reads only what it wrote itself -- its LDS
// instructions execute in order, so the compiler only has to keep them in program order.
template <bool WG>
__device__ __forceinline__ void big_sync() {
    if constexpr (WG) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
template <bool WG>
__device__ __forceinline__ void big_exchange(c2 (&v)[16], v2* __restrict__ lds, int wb, int ws, int rb, int rs) {
    if constexpr (WG) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[wb + ws * r] = v[r].re;
    big_sync<WG>();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r].re = lds[rb + rs * r];
    big_sync<WG>();
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[wb + ws * r] = v[r].im;
    big_sync<WG>();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r].im = lds[rb + rs * r];
}

// Everything after stage A (wg_fft_tail of fft_core.hpp for these lengths): given y[c0][b] --
// thread b = tid, register c0, already twiddled -- the T-point transform over b of each of the 16
// sequences c0.  On return thread t holds in register j output k' = (t >> 4) + M j of sequence
// c0 = t & 15: for the full transform element t + T j (k = c0 + 16 k').
template <int N, int SIGN>
__device__ __forceinline__ void wg_fft_big_tail(c2 (&v)[16], v2* __restrict__ lds, int tid,
                                                const cf* __restrict__ tw) {
    typedef BigGeo<N> G;
    constexpr int M = G::M, L = G::L, P1 = G::P1, P2 = G::P2, P3 = G::P3;
    const int c0 = tid / M, b1 = tid % M;
    big_exchange<true>(v, lds, tid, P1, c0 * P1 + b1, M);        // E1
    {
        cf w[15];
        big_load_tw(w, tw + G::TW_B, M, b1);
        fft_butterfly_twiddle<SIGN>(v, w);                       // stage B
    }
    const int b2 = b1 >> 4, c1 = b1 & 15;                        // (the lane's new place in its sequence)
    v2* seq = lds + c0 * P1;
    big_exchange<false>(v, seq, b1, P2, c1 * P2 + b2, L);        // E2
    {
        cf w[15];
        big_load_tw(w, tw + G::TW_C, L, b2);
        fft_butterfly_twiddle<SIGN>(v, w);                       // stage C
    }
    // E3: register c2 to row c0 at (c2 L + b2) 16 + c1; thread c0' + 16 c1' + 256 g takes
    // (g + L u, b2') into register u L + b2'
    const int oc0 = tid & 15, oc1 = (tid >> 4) & 15, og = tid >> 8;
    const int wb = c0 * P3 + b2 * 16 + c1, rb = oc0 * P3 + og * (16 * L) + oc1;
    c2 x[16];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[wb + (16 * L) * r] = v[r].re;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16 / L; ++u)
#pragma unroll
        for (int b = 0; b < L; ++b) x[u * L + b].re = lds[rb + u * (16 * L * L) + b * 16];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[wb + (16 * L) * r] = v[r].im;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16 / L; ++u)
#pragma unroll
        for (int b = 0; b < L; ++b) x[u * L + b].im = lds[rb + u * (16 * L * L) + b * 16];
    // stage D
#pragma unroll
    for (int u = 0; u < 16 / L; ++u) {
        c2 t[L];
#pragma unroll
        for (int b = 0; b < L; ++b) t[b] = x[u * L + b];
        radixR<SIGN, L>(t);
#pragma unroll
        for (int c3 = 0; c3 < L; ++c3) v[u + (16 / L) * c3] = t[c3];
    }
}

template <int N, int SIGN>
__device__ __forceinline__ void wg_fft_big(c2 (&v)[16], v2* __restrict__ lds, int tid,
                                           const cf* __restrict__ tw) {
    {
        cf w[15];
        big_load_tw(w, tw, BigGeo<N>::T, tid);
        fft_butterfly_twiddle<SIGN>(v, w);                       // stage A
    }
    wg_fft_big_tail<N, SIGN>(v, lds, tid, tw);
}

// Batched transforms over contiguous groups of N complete samples of stream pairs (k_fft_rows for
// these lengths).  grid: n_fft * npair workgroups of T threads, BigGeo<N>::LDS_ELEMS * 8 bytes of
// dynamic LDS.
template <int N, int SIGN>
__global__ __launch_bounds__(N / 16, 4) void k_fft_rows_big(const float2* __restrict__ in,
                                                          float2* __restrict__ out, long long n_fft, int S,
                                                          float scale, const cf* __restrict__ tw) {
    extern __shared__ v2 big_lds[];
    constexpr int T = BigGeo<N>::T;
    const int tid = threadIdx.x, npair = S >> 1;
    const long long i = blockIdx.x / npair;
    const int sp = blockIdx.x % npair;
    const float2* src = in + ((i * N + tid) * S + 2 * sp);
    c2 v[16];
    if (S == 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_ext_nt(src + (long long)T * j * S);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_ext(src + (long long)T * j * S);
    }
    wg_fft_big<N, SIGN>(v, big_lds, tid, tw);
    float2* dst = out + ((i * N + tid) * S + 2 * sp);
#pragma unroll
    for (int j = 0; j < 16; ++j)
        st_ext(dst + (long long)T * j * S, c2{v[j].re * scale, v[j].im * scale}, S == 2);
}

}  // namespace bbt
