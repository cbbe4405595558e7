// HIP kernels for the dedispersion -> channelizer path on gfx950.
//
// Data contract everywhere: C-contiguous, time-major, streams innermost,
// interleaved (re, im) float32 -- numpy complex64 of shape (n, S).  S (number
// of streams = prod(sample_shape)) is even; kernels work on stream PAIRS so a
// complete 2-pol sample is one aligned 16-byte float4.
//
//   k_fft_rows      Channelize.task / Dechannelize.task
//                   (reference baseband_tasks/channelize.py:73-74, 164-165)
//   k_osm_*         Disperse.task / Convolve.task: ifft(fft(x) * H)[valid]
//                   (reference dispersion.py:135-139, convolution.py:116-120)
//                   as a four-step FFT  N = N1 x N2:
//                     col pass (fwd, over n1) -> row pass (twiddle, fwd over
//                     n2, * H, inv over k2, conj twiddle) -> col pass (inv
//                     over k1, writes only the valid samples)
//   k_pfb           PolyphaseFilterBankSamples.ppf + Channelize.task
//                   (reference pfb.py:91-100, channelize.py:73-74)
//
// Registers hold both streams of a pair packed across each other (c2, see
// fft_core.hpp); the work buffer between the overlap-save passes is stored
// in that order too (re_A re_B im_A im_B), external arrays are numpy order.
#pragma once
#include <hip/hip_runtime.h>
#include "fft_core.hpp"
#include "osm_chunk.hpp"

// Transform twiddles W^{tau c}, c < 16, of a stage: 0 = fifteen table loads per thread, 1 = one
// load and its powers (products at most four roundings deep), 2 = four loads (c = 1, 2, 4, 8)
// and eleven products at most three deep (fft_core.hpp).  The kernels that wait for loads rather
// than for the VALU gain; measured on MI355X (round 3, same-box pairs):
//   row pass      headline 49.63-49.96 -> 50.51-50.78 (1) / 50.58-50.89 (2) G; rel-L2 against the
//                 oracle 3.81e-7 -> 5.77e-7 (1) / 4.23e-7 (2): 2
//   one-kernel    inverse filter bank 51.0 -> 51.9 G, config 5 8.44 -> 8.62 G (1 and 2 alike): 2
//   filter bank   config 3 136.5 -> 138.4-140.0 (1) / 139.0 (2) G: 2
//   column passes 48.8 -> 48.6 G and Channelize (k_fft_rows) 193.8 -> 194.0 G: no gain, 0
#ifndef BBT_SHORT_TW_POW
#define BBT_SHORT_TW_POW 1            // k_fft_short (Channelize with 32 .. 128 channels): the same
#endif
#ifndef BBT_SMALL_TW_POW
#define BBT_SMALL_TW_POW 2
#endif
#ifndef BBT_ROWS_TW_POW
#define BBT_ROWS_TW_POW 0
#endif
#ifndef BBT_COL_TW_POW
#define BBT_COL_TW_POW 0
#endif
#ifndef BBT_ROWPASS_TW_POW
#define BBT_ROWPASS_TW_POW 2
#endif
#ifndef BBT_PFB_TW_POW
#define BBT_PFB_TW_POW 2
#endif

namespace bbt {

// (xcd_remap: osm_chunk.hpp)

// ---------------------------------------------------------------------------
// Batched FFT over contiguous groups of N complete samples.
//   in/out : (n_fft * N, S) complex64 ; pair index = blockIdx.y
// SINGLE (S == 1): the two transforms a thread carries side by side are two
// consecutive groups of the one stream instead of the two streams of a pair
// (8-byte loads and stores, fully coalesced; nothing is padded or wasted).
// SPLIT (with SINGLE, forward): the one stream is z = a + i b of two real streams;
// instead of the spectrum Z of z the kernel writes the half spectra of a and b,
//   A[k] = (Z[k] + conj Z[n-k]) / 2,  B[k] = (Z[k] - conj Z[n-k]) / 2i,  k <= n/2,
// as (transform, k, 2 streams): Z goes through the transform's exchange area once
// more (real parts, then imaginary parts) so that a thread can pair k with n - k.
// This is Channelize of float32 streams without the separate pass over the spectra.
template <int N, int SIGN, int FPW, bool SINGLE = false, bool SPLIT = false>
__global__ __launch_bounds__(FPW* N / 16) void k_fft_rows(const float2* __restrict__ in,
                                                           float2* __restrict__ out, long long n_fft,
                                                           int S, float scale,
                                                           const cf* __restrict__ tw0,
                                                           const cf* __restrict__ tw1) {
    typedef FftGeo<N> G;
    constexpr int T = G::T;
    __shared__ v2 lds[FPW * G::LDS_ELEMS];
    const int slot = threadIdx.x / T, tau = threadIdx.x % T;
    // pairs of the same samples share cache lines: keep them adjacent in the
    // XCD-contiguous virtual block order
    const int npair = SINGLE ? 1 : S >> 1;
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    const long long i = (long long)(vb / npair) * FPW + slot;
    const int sp = vb % npair;
    c2 v[16];
    if constexpr (SINGLE) {
        const bool act_a = 2 * i < n_fft, act_b = 2 * i + 1 < n_fft;
        const float2* src = in + (2 * i * N + tau);
        const float2 zero = make_float2(0.f, 0.f);
        if constexpr (SPLIT && SIGN > 0) {
            // the inverse of the split below: the input holds the half spectra (transform, k, 2
            // streams) of two real streams a, b; Z[k] = A[k] + i B[k] for k <= n/2, and
            // conj(A[n-k]) + i conj(B[n-k]) above (imaginary parts of DC and Nyquist ignored, as
            // irfft does) -- Dechannelize to two float32 streams without a separate merge pass
            constexpr int HALF = N / 2 + 1;
            const float4* half_a = reinterpret_cast<const float4*>(in) + 2 * i * HALF;
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = tau + T * j, kk = k <= N / 2 ? k : N - k;
                const float4 pa = act_a ? half_a[kk] : zero4, pb = act_b ? half_a[HALF + kk] : zero4;   // (A.re A.im B.re B.im)
                const float sg = (kk == 0 || 2 * kk == N) ? 0.f : (k <= N / 2 ? 1.f : -1.f);     // (0: DC, Nyquist)
                // upper: (A.re - B.im, A.im + B.re); lower (mirror): (A.re + B.im, B.re - A.im)
                v[j] = c2{v2{pa.x - sg * pa.w, pb.x - sg * pb.w}, v2{pa.z + sg * pa.y, pb.z + sg * pb.y}};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 a = act_a ? src[T * j] : zero;
                const float2 b = act_b ? src[N + T * j] : zero;
                v[j] = c2{v2{a.x, b.x}, v2{a.y, b.y}};
            }
        }
        wg_fft<N, SIGN, false, 0, BBT_ROWS_TW_POW>(v, lds + slot * G::LDS_ELEMS, tau, 0, tw0, tw1);
        if constexpr (SPLIT && SIGN < 0) {
            v2* area = lds + slot * G::LDS_ELEMS;                 // N elements fit (LDS_ELEMS >= N)
            // k = tau + T j for j < 8 covers 0 .. N/2 - 1; thread 0 also takes k = N/2
            v2 zk_re[9], zm_re[9], zk_im[9], zm_im[9];
            __syncthreads();                                      // (the transform's last reads are done)
#pragma unroll
            for (int j = 0; j < 16; ++j) area[tau + T * j] = v[j].re;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = tau + T * j;
                zk_re[j] = area[k];
                zm_re[j] = area[(N - k) & (N - 1)];
            }
            zk_re[8] = zm_re[8] = area[N / 2];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) area[tau + T * j] = v[j].im;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = tau + T * j;
                zk_im[j] = area[k];
                zm_im[j] = area[(N - k) & (N - 1)];
            }
            zk_im[8] = zm_im[8] = area[N / 2];
            constexpr int HALF = N / 2 + 1;
            float4* dst_a = reinterpret_cast<float4*>(out) + 2 * i * HALF;        // (a, b) per k: 16 bytes
            float4* dst_b = dst_a + HALF;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (j == 8 && tau != 0) break;
                const int k = j < 8 ? tau + T * j : N / 2;
                const v2 ar = 0.5f * (zk_re[j] + zm_re[j]), ai = 0.5f * (zk_im[j] - zm_im[j]);
                const v2 br = 0.5f * (zk_im[j] + zm_im[j]), bi = -0.5f * (zk_re[j] - zm_re[j]);
                if (act_a) dst_a[k] = make_float4(ar.x, ai.x, br.x, bi.x);
                if (act_b) dst_b[k] = make_float4(ar.y, ai.y, br.y, bi.y);
            }
            return;
        }
        float2* dst = out + (2 * i * N + tau);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (act_a) dst[T * j] = make_float2(v[j].re.x * scale, v[j].im.x * scale);
            if (act_b) dst[N + T * j] = make_float2(v[j].re.y * scale, v[j].im.y * scale);
        }
        return;
    }
    const bool active = i < n_fft;
    if constexpr (SPLIT && SIGN > 0) {
        // inverse of the pair-mode split: half spectra (transform, k, 2 S reals) in, S streams z = a + i b out
        constexpr int HALF = N / 2 + 1;
        const float4* half = reinterpret_cast<const float4*>(in) + i * HALF * S + 2 * sp;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = tau + T * j, kk = k <= N / 2 ? k : N - k;
            const float4 p0 = active ? half[(long long)kk * S] : zero4, p1 = active ? half[(long long)kk * S + 1] : zero4;
            // +1 / -1 above and below n/2; 0 at DC and Nyquist, whose imaginary parts do not count
            const float sg = (kk == 0 || 2 * kk == N) ? 0.f : (k <= N / 2 ? 1.f : -1.f);
            v[j] = c2{v2{p0.x - sg * p0.w, p1.x - sg * p1.w}, v2{p0.z + sg * p0.y, p1.z + sg * p1.y}};
        }
    } else if (active) {
        const float2* src = in + ((i * N + tau) * S + 2 * sp);
        if (S == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_ext_nt(src + (long long)T * j * S);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_ext(src + (long long)T * j * S);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = czero();
    }
    wg_fft<N, SIGN, false, 0, BBT_ROWS_TW_POW>(v, lds + slot * G::LDS_ELEMS, tau, 0, tw0, tw1);
    if constexpr (SPLIT && SIGN < 0) {
        // Stream pairs, each stream z = a + i b of two real streams (S complex = 2 S real streams):
        // the half spectra of all of them, (transform, k, 2 S reals), as in the one-stream case above.
        v2* area = lds + slot * G::LDS_ELEMS;
        v2 zk_re[9], zm_re[9], zk_im[9], zm_im[9];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) area[tau + T * j] = v[j].re;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = tau + T * j;
            zk_re[j] = area[k];
            zm_re[j] = area[(N - k) & (N - 1)];
        }
        zk_re[8] = zm_re[8] = area[N / 2];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) area[tau + T * j] = v[j].im;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = tau + T * j;
            zk_im[j] = area[k];
            zm_im[j] = area[(N - k) & (N - 1)];
        }
        zk_im[8] = zm_im[8] = area[N / 2];
        if (active) {
            constexpr int HALF = N / 2 + 1;
            // per (transform, k): S float4, one per complex stream; this pair's two are adjacent
            float4* dst = reinterpret_cast<float4*>(out) + i * HALF * S + 2 * sp;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (j == 8 && tau != 0) break;
                const int k = j < 8 ? tau + T * j : N / 2;
                const v2 ar = 0.5f * (zk_re[j] + zm_re[j]), ai = 0.5f * (zk_im[j] - zm_im[j]);
                const v2 br = 0.5f * (zk_im[j] + zm_im[j]), bi = -0.5f * (zk_re[j] - zm_re[j]);
                dst[(long long)k * S] = make_float4(ar.x, ai.x, br.x, bi.x);
                dst[(long long)k * S + 1] = make_float4(ar.y, ai.y, br.y, bi.y);
            }
        }
        return;
    }
    if (active) {
        float2* dst = out + ((i * N + tau) * S + 2 * sp);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            st_ext(dst + (long long)T * j * S, c2{v[j].re * scale, v[j].im * scale}, S == 2);
    }
}

// k_fft_rows for MANY streams: the lanes run over PP stream pairs first (PP * 16 contiguous bytes
// of every complete sample, where one pair per workgroup takes 16 bytes out of every S * 8-byte
// row: Channelize(1024) of 16 / 128 / 2048 streams ran at 156 / 141 / 116 G stream-samples/s
// against 374 for two), the PP transforms interleaved in the exchange area (COLMODE = PP, as
// k_osm_small).  One transform per pair and workgroup; PP * LDS_ELEMS * 8 bytes of dynamic LDS.
template <int N, int SIGN, int PP>
__global__ __launch_bounds__(PP* N / 16) void k_fft_rows_pp(const float2* __restrict__ in,
                                                            float2* __restrict__ out, long long n_fft, int S,
                                                            float scale, const cf* __restrict__ tw0,
                                                            const cf* __restrict__ tw1) {
    typedef FftGeo<N> G;
    constexpr int T = G::T;
    extern __shared__ v2 rows_pp_lds[];
    const int npg = (S >> 1) / PP;                      // groups of PP pairs (npair % PP == 0)
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    const int pl = threadIdx.x % PP, tau = threadIdx.x / PP;
    const long long i = vb / npg;
    const int sp = (vb % npg) * PP + pl;
    const float2* src = in + ((i * N + tau) * S + 2 * sp);
    c2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = ld_ext(src + (long long)T * j * S);
    wg_fft<N, SIGN, PP, 0, BBT_ROWS_TW_POW>(v, rows_pp_lds, tau, pl, tw0, tw1);
    float2* dst = out + ((i * N + tau) * S + 2 * sp);
#pragma unroll
    for (int j = 0; j < 16; ++j) st_ext(dst + (long long)T * j * S, c2{v[j].re * scale, v[j].im * scale});
}

// Batched FFT for short groups, N = 2 .. 128 (Channelize with few channels).
//   N <= 16 : one thread per transform, radix-N in registers.
//   N = 16 R, R in {2, 4, 8}: R threads per transform; radix-16 over the
//             elements tau + R j, twiddle W_N^{tau c}, exchange through LDS,
//             radix-R; output k = c + 16 k'.
// Lanes run over PP stream pairs first (PP * 16 bytes of one complete sample
// are contiguous: with 8 pairs a whole 128-byte line per 8 lanes; config 4's
// share on one GPU went from 2.06 to 2.25 G complete samples/s), then over the R threads of
// a transform (consecutive samples), then over transforms.
template <int N, int SIGN, int PP>
__global__ __launch_bounds__(256) void k_fft_short(const float2* __restrict__ in,
                                                   float2* __restrict__ out, long long n_fft, int S,
                                                   float scale, const cf* __restrict__ wroot) {
    constexpr int R = (N <= 16) ? 1 : N / 16;        // threads per transform
    constexpr int P = (N <= 16) ? N : 16;            // points per thread
    constexpr int FPW = 256 / R;                     // (transform, pair) slots per workgroup
    constexpr int FPB = FPW / PP;                    // transforms per workgroup
    constexpr int PITCH = R + 1;
    __shared__ v2 lds[(R > 1) ? FPW * 16 * PITCH : 1];
    const int npair = S >> 1, npg = npair / PP;      // pair groups (npair % PP == 0)
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    const int pl = threadIdx.x % PP, rest = threadIdx.x / PP;
    const int tau = rest % R, tslot = rest / R;
    const int slot = tslot * PP + pl;                // LDS region
    const long long i = (long long)(vb / npg) * FPB + tslot;
    const int sp = (vb % npg) * PP + pl;
    const bool active = i < n_fft;
    c2 v[P];
    if (active) {
#pragma unroll
        for (int j = 0; j < P; ++j) v[j] = ld_ext(in + ((i * N + tau + R * j) * S + 2 * sp));
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) v[j] = czero();
    }
    radixR<SIGN, P>(v);
    if constexpr (R == 1) {
        if (active) {
#pragma unroll
            for (int j = 0; j < P; ++j)
                st_ext(out + ((i * N + j) * S + 2 * sp), c2{v[j].re * scale, v[j].im * scale});
        }
    } else {
        constexpr int NU = 16 / R;
#if BBT_SHORT_TW_POW
        {   // W_N^{tau c}, c < 16, from four loads and products (fft_twiddle_powers4)
            cf w[15];
            const int step = tau * (4096 / N);
            w[0] = wroot[step];
            w[1] = wroot[2 * step];
            w[3] = wroot[4 * step];
            w[7] = wroot[8 * step];
            fft_twiddle_powers4(w);
#pragma unroll
            for (int c = 1; c < 16; ++c) v[c] = twmul<SIGN>(v[c], w[c - 1]);
        }
#else
#pragma unroll
        for (int c = 1; c < 16; ++c) v[c] = twmul<SIGN>(v[c], wroot[(tau * c) * (4096 / N)]);
#endif
        v2* my = lds + slot * 16 * PITCH;
        c2 t[NU][R];
#pragma unroll
        for (int c = 0; c < 16; ++c) my[c * PITCH + tau] = v[c].re;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int b = 0; b < R; ++b) t[u][b].re = my[(tau + R * u) * PITCH + b];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) my[c * PITCH + tau] = v[c].im;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int b = 0; b < R; ++b) t[u][b].im = my[(tau + R * u) * PITCH + b];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            radixR<SIGN, R>(t[u]);
            if (active) {
#pragma unroll
                for (int kp = 0; kp < R; ++kp)
                    st_ext(out + ((i * N + (tau + R * u) + 16 * kp) * S + 2 * sp),
                           c2{t[u][kp].re * scale, t[u][kp].im * scale});
            }
        }
    }
}

// The same for N <= 16 on ONE stream pair (S == 2), where a thread's N samples are 16 N contiguous
// bytes and the lanes of a load sit 16 N bytes apart -- 64 cache lines per wave instruction for
// 16 bytes each.  Here the workgroup's 256 transforms (4096 N contiguous bytes) come in and go out
// in whole lines and change hands in LDS: element e at e + e / N (16-byte slots, one pad per
// transform: a thread's N slots are then (N + 1) 16 bytes apart from its neighbour's, free of
// bank conflicts for N = 2 .. 16); a thread reads and rewrites only its own slots, so one barrier
// each way.  T transforms per workgroup: T (N + 1) 16 bytes of dynamic LDS.
template <int N, int SIGN, int T>
__global__ __launch_bounds__(T) void k_fft_tiny(const float2* __restrict__ in, float2* __restrict__ out,
                                                  long long n_fft, float scale) {
    extern __shared__ f4v tiny_lds[];
    const int t = threadIdx.x;
    const long long first = (long long)blockIdx.x * T * N;         // first complete sample of this workgroup
    const long long total = n_fft * N;
    const f4v* src = reinterpret_cast<const f4v*>(in) + first;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int e = t + T * k;
        if (first + e < total) tiny_lds[e + e / N] = __builtin_nontemporal_load(src + e);
    }
    __syncthreads();
    const bool active = (long long)blockIdx.x * T + t < n_fft;
    if (active) {
        c2 v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const f4v x = tiny_lds[t * (N + 1) + j];
            v[j] = c2{v2{x.x, x.z}, v2{x.y, x.w}};
        }
        radixR<SIGN, N>(v);
#pragma unroll
        for (int j = 0; j < N; ++j)
            tiny_lds[t * (N + 1) + j] = f4v{v[j].re.x * scale, v[j].im.x * scale, v[j].re.y * scale, v[j].im.y * scale};
    }
    __syncthreads();
    f4v* dst = reinterpret_cast<f4v*>(out) + first;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int e = t + T * k;
        if (first + e < total) __builtin_nontemporal_store(tiny_lds[e + e / N], dst + e);
    }
}

// (overlap-save block descriptors OsmBlock / OsmChunk: osm_chunk.hpp)

// Where the fused channelizer's column pass puts its spectra.
struct SpecOut {
    float2* seam;        // [block][head, tail][pair][n_chan] spectra straddling block seams
    long long s_base;    // spectrum index stored at out[0]
    long long n_out;     // number of spectra in out
    int n_chan;
    int lg_chan;         // log2(n_chan)
    int n_fft;           // N = N1 * N2
    // fused detection + integration (k_osm_col256<..., DET>): spectra are not
    // stored; their power is summed into det[bin][channel][...], bin =
    // (spectrum - s_base) / det_step, n_out = number of bins * det_step.
    float* det;          // zeroed by the caller; nullptr: store spectra
    int det_step;
    int det_mode;        // 0 |z|^2 per stream, 1 (|X|^2, |Y|^2, Re XY*, Im XY*) per pair
    float det_scale;
    // n_chan < 256: the row pass leaves channel a + 16 bitrev_L(c) of group q at
    // row position (L q + c) + T a, L = n_chan / 16, T = N2 / 16 (k_osm_rowpass)
    int tiny_lg;         // n_chan = 2, 4, 8 (log2 here, else 0): the row pass transforms the n_chan neighbouring
                         // LANES that hold a group and leaves channel bitrev(c) at position q n_chan + c
    int small_l;         // 0: natural order (position == q * n_chan + channel)
    int small_row;       // row length N2 of the row pass (a column pass of a three-level
                         // transform sees 16 such rows side by side)
    // four-step twiddles applied by the 256-point column passes instead of the row pass
    // (col_twiddles below): twa [16][N2] = W_N^{tau n2}, twg [4][N2] = W_N^{16 n2 2^i}; nullptr:
    // the row pass applies them
    const cf* twa;
    const cf* twg;
};

// The four-step twiddles W_N^{k1 n2} of a two-level transform (N = 256 N2) on the registers of a
// 256-point column pass: thread (tau, column n2) holds k1 = tau + 16 j, so the factor is
// a g^j with a = W_N^{tau n2} and g, g^2, g^4, g^8 (g = W_N^{16 n2}) from tables
// (float64-derived), the other powers by products at most four roundings deep.  SIGN < 0: multiply (forward, after the
// column transform); SIGN > 0: by the conjugate (inverse, before it).  The column passes issue
// 8-22 % of their cycles (profiles/r03_headline_sq_counters.json), the row pass that did this
// before is bound by its VALU work.
template <int SIGN>
__device__ __forceinline__ void col_twiddles(c2 (&v)[16], const cf* __restrict__ twa,
                                             const cf* __restrict__ twg, int tau, int n2, int N2) {
    const cf a = twa[tau * N2 + n2], g = twg[n2], g2 = twg[N2 + n2], g4 = twg[2 * N2 + n2],
             g8 = twg[3 * N2 + n2];
    cf t[4];                                  // a g^{4 m}, m < 4
    t[0] = a;
    t[1] = cmul(a, g4);
    t[2] = cmul(a, g8);
    t[3] = cmul(t[2], g4);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const cf t1 = cmul(t[m], g), t2 = cmul(t[m], g2);
        v[4 * m] = twmul<SIGN>(v[4 * m], t[m]);
        v[4 * m + 1] = twmul<SIGN>(v[4 * m + 1], t1);
        v[4 * m + 2] = twmul<SIGN>(v[4 * m + 2], t2);
        v[4 * m + 3] = twmul<SIGN>(v[4 * m + 3], cmul(t2, g));
    }
}

// Column position -> q * n_chan + channel for the small-channel-count layout.
__device__ __forceinline__ int small_channel_slot(int pos, const SpecOut& so) {
    if (so.tiny_lg) {
        const int c = pos & ((1 << so.tiny_lg) - 1);
        return pos - c + (int)(__brev((unsigned)c) >> (32 - so.tiny_lg));
    }
    if (!so.small_l) return pos;
    const int L = so.small_l, N2 = so.small_row, T = N2 >> 4;
    const int row = pos / N2, p = pos - row * N2;
    const int a = p / T, t = p - a * T;
    const int q = t / L, c = t - q * L;
    const int r = L == 1 ? 0 : (int)(__brev((unsigned)c) >> (32 - (31 - __clz(L))));
    return row * N2 + q * so.n_chan + a + 16 * r;
}

// floats a pair contributes per (bin, channel): 2 (mode 0) or 4 (mode 1)
__device__ __forceinline__ float4 detect_pair(c2 z, int mode) {
    float4 p;
    p.x = z.re.x * z.re.x + z.im.x * z.im.x;
    p.y = z.re.y * z.re.y + z.im.y * z.im.y;
    p.z = mode ? z.re.x * z.re.y + z.im.x * z.im.y : 0.f;
    p.w = mode ? z.im.x * z.re.y - z.re.x * z.im.y : 0.f;
    return p;
}
// index of component 0 of (bin, channel, pair) in the detected output
__device__ __forceinline__ long long detect_index(long long bin, int ch, int sp, int lg_chan,
                                                  int npair, int mode) {
    return (((bin << lg_chan) + ch) * npair + sp) * (mode ? 4 : 2);
}
#define BBT_DET_MAX_BINS 64      // integration bins one column-pass workgroup may touch

// Fused channelizer output.  A column-pass thread holds, for one column
// `slot` = q * n_chan + ch of the work rows, the rows n1 = row0 + 16 j (j < 16,
// N1 == 256) or n1 = j (N1 == 16): element j is channel ch of the spectrum
// whose samples are the (circularly shifted) block samples
// [n1 * N2 + q * n_chan, + n_chan).  Everything is linear in j, so the
// per-thread part is computed once: position of the group relative to the
// first kept sample (rel), its stride, and the output pointer and stride.
struct SpecCursor {
    float2* dst;        // where a fully valid element j == 0 would go
    long long dstride;  // float2 units per j
    float2* seam0;      // this thread's slot in the block's head / tail seam spectrum
    float2* seam1;
    int rel, drel;      // group start minus valid_start, and its step per j
    int wrap_at;        // rel value from which the group wraps around the block end
    int n_fft, nch, vc;
    long long s0, ds, n_out;   // spectrum index (relative to out[0]) and its step per j
    bool nt;                   // whole-line output stream (S == 2): non-temporal stores
};
__device__ __forceinline__ SpecCursor spec_cursor(float2* __restrict__ out, const SpecOut& so,
                                                  const OsmBlock& blk, int row0, int row_step,
                                                  int N2, int slot, int S, int sp, int npair) {
    SpecCursor c;
    const int lg = so.lg_chan;
    c.nch = 1 << lg;
    const int ch = slot & (c.nch - 1);
    c.n_fft = so.n_fft;
    c.vc = blk.valid_count;
    c.rel = row0 * N2 + (slot - ch) + blk.shift - blk.valid_start;
    c.drel = row_step * N2;
    c.wrap_at = so.n_fft - c.nch - blk.valid_start + 1;      // rel >= wrap_at  <=>  ystart + nch > N
    c.s0 = ((blk.out_off + c.rel) >> lg) - so.s_base;        // (arithmetic shift: exact multiples)
    c.ds = (long long)c.drel >> lg;
    c.n_out = so.n_out;
    c.nt = S == 2;
    c.dst = out + (((c.s0 << lg) + ch) * S + 2 * sp);
    c.dstride = ((c.ds << lg)) * S;
    c.seam0 = so.seam + (((((long long)blk.index * 2 + 0) * npair + sp) << lg) + ch) * 2;
    c.seam1 = so.seam + (((((long long)blk.index * 2 + 1) * npair + sp) << lg) + ch) * 2;
    return c;
}
__device__ __forceinline__ void emit_spectrum(c2 val, const SpecCursor& c, int j) {
    int rel = c.rel + j * c.drel;
    if (rel >= c.wrap_at) {
        // The one group that wraps around the end of the block.  Its leading samples
        // are the block's last ones, so it can hold the end of the kept range (when
        // n_chan exceeds the padding); its trailing samples sit before sample 0, so
        // it can hold the start of the kept range too.  Both are seam spectra.
        if (rel < c.vc) st_ext(c.seam1, val);
        rel -= c.n_fft;
        if (rel < 0 && rel + c.nch > 0) st_ext(c.seam0, val);
        return;
    }
    if (rel >= 0 && rel + c.nch <= c.vc) {
        const long long s = c.s0 + j * c.ds;
        if (s >= 0 && s < c.n_out) st_ext(c.dst + j * c.dstride, val, c.nt);
    } else if (rel < 0 && rel + c.nch > 0) {                  // straddles the block start
        st_ext(c.seam0, val);
    } else if (rel < c.vc && rel + c.nch > c.vc) {            // straddles the block end
        st_ext(c.seam1, val);
    }
}

// One stream: the value of one block (one half of the register pair) goes out as 8 bytes;
// the seam slots keep the pair format (second half zero) so k_seam_fix serves both.
__device__ __forceinline__ void emit_spectrum_single(float2 val, const SpecCursor& c, int j) {
    const c2 padded = c2{v2{val.x, 0.f}, v2{val.y, 0.f}};
    int rel = c.rel + j * c.drel;
    if (rel >= c.wrap_at) {
        if (rel < c.vc) st_ext(c.seam1, padded);
        rel -= c.n_fft;
        if (rel < 0 && rel + c.nch > 0) st_ext(c.seam0, padded);
        return;
    }
    if (rel >= 0 && rel + c.nch <= c.vc) {
        const long long s = c.s0 + j * c.ds;
        if (s >= 0 && s < c.n_out) c.dst[j * c.dstride] = val;
    } else if (rel < 0 && rel + c.nch > 0) {
        st_ext(c.seam0, padded);
    } else if (rel < c.vc && rel + c.nch > c.vc) {
        st_ext(c.seam1, padded);
    }
}

// One-stream plans (S == 1, SINGLE kernels): the two transforms a thread carries
// side by side are two consecutive BLOCKS of the one stream -- blocks 2q and
// 2q + 1 of the chunk, the second absent when the chunk holds an odd number --
// instead of the two streams of a pair.  Only the stream side differs (8-byte
// loads and stores with each block's own offsets); the work buffers, the row
// pass and the response (the same column for both) are those of a pair.
struct SinglePair {
    OsmBlock a, b;
    bool has_b;
};
__device__ __forceinline__ SinglePair single_pair(const OsmChunk& ch, int q) {
    SinglePair sp;
    sp.a = osm_block(ch, 2 * q);
    sp.has_b = 2 * q + 1 < (ch.reg_count ? ch.reg_count : ch.nblk);
    sp.b = osm_block(ch, sp.has_b ? 2 * q + 1 : 2 * q);
    return sp;
}
__device__ __forceinline__ c2 ld_single(const float2* __restrict__ in, const SinglePair& sp, long long e) {
    const float2 a = in[sp.a.in_off + e];
    const float2 b = sp.has_b ? in[sp.b.in_off + e] : make_float2(0.f, 0.f);
    return c2{v2{a.x, b.x}, v2{a.y, b.y}};
}
// (fused channelizer: each block read circularly shifted by its own blk.shift, n = block length)
__device__ __forceinline__ c2 ld_single_shifted(const float2* __restrict__ in, const SinglePair& sp, long long e,
                                                long long n) {
    long long ea = e + sp.a.shift, eb = e + sp.b.shift;
    ea -= ea >= n ? n : 0;
    eb -= eb >= n ? n : 0;
    const float2 a = in[sp.a.in_off + ea];
    const float2 b = sp.has_b ? in[sp.b.in_off + eb] : make_float2(0.f, 0.f);
    return c2{v2{a.x, b.x}, v2{a.y, b.y}};
}
__device__ __forceinline__ void st_single(float2* __restrict__ out, const SinglePair& sp, long long e, c2 v) {
    const long long ra = e - sp.a.valid_start, rb = e - sp.b.valid_start;
    if (ra >= 0 && ra < sp.a.valid_count) out[sp.a.out_off + ra] = make_float2(v.re.x, v.im.x);
    if (sp.has_b && rb >= 0 && rb < sp.b.valid_count) out[sp.b.out_off + rb] = make_float2(v.re.y, v.im.y);
}

// Multiply by the response (chirp), already scaled by 1/N: element j of this
// thread is at h[STRIDE * j].  The "same column for both streams" test is
// hoisted out of the loop (a per-element runtime select makes hipcc branch
// around every load).
template <int STRIDE>
__device__ __forceinline__ void apply_resp(c2 (&v)[16], const cf* __restrict__ h0,
                                           const cf* __restrict__ h1, bool same) {
    if (same) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = twmul<-1>(v[j], h0[STRIDE * j]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const cf x = h0[STRIDE * j], y = h1[STRIDE * j];
            v[j] = cmul2(v[j], c2{v2{x.x, y.x}, v2{x.y, y.y}});
        }
    }
}

// Single-kernel path, N <= 4096: one workgroup per (block, group of PP pairs).
// With many streams (InversePolyphaseFilterBank runs this along the block axis
// with n_chan * S streams) the lanes run over PP pairs first, so a wave touches
// PP * 16 contiguous bytes of each complete sample instead of 16; the PP
// transforms are interleaved in LDS (COLMODE = PP).
// (its transform twiddles come as powers, BBT_SMALL_TW_POW below)
template <int N, int PP, bool SINGLE = false>
__global__ __launch_bounds__(PP* N / 16) void k_osm_small(const float2* __restrict__ in,
                                                          float2* __restrict__ out, OsmChunk ch, int S,
                                                          const cf* __restrict__ resp,
                                                          const int* __restrict__ resp_index,
                                                          const cf* __restrict__ tw0,
                                                          const cf* __restrict__ tw1) {
    typedef FftGeo<N> G;
    constexpr int T = G::T;
    constexpr int CM = PP > 1 ? PP : 0;
    extern __shared__ v2 osm_small_lds[];          // G::LDS_ELEMS * PP elements (up to 72 KiB)
    v2* lds = osm_small_lds;
    // the pairs of one block share cache lines: consecutive virtual ids, one XCD
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    c2 v[16];
    if constexpr (SINGLE) {                       // PP == 1; workgroup vb = pair of blocks
        const int tau = threadIdx.x;
        const SinglePair pr = single_pair(ch, vb);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_single(in, pr, tau + T * j);
        wg_fft<N, -1, CM, 0, BBT_SMALL_TW_POW>(v, lds, tau, 0, tw0, tw1);
        const cf* h = resp + (long long)resp_index[0] * N + tau;
        apply_resp<T>(v, h, h, true);
        wg_fft<N, +1, CM, 0, BBT_SMALL_TW_POW>(v, lds, tau, 0, tw0, tw1);
#pragma unroll
        for (int j = 0; j < 16; ++j) st_single(out, pr, tau + T * j, v[j]);
        return;
    }
    const int npg = (S >> 1) / PP;
    const int pl = threadIdx.x % PP, tau = threadIdx.x / PP;
    // Which (block, pair group) a workgroup takes.  Default: consecutive virtual ids = the pair
    // groups of one block (their 64-byte runs share lines), so an XCD holds a few whole blocks.
    // With many pair groups (InversePolyphaseFilterBank: one response column per polyphase phase,
    // 8 MiB in all) an XCD instead takes an eighth of the GROUPS of every block of the launch:
    // its share of the response (1 MiB) then stays in its L2 from block to block instead of
    // being fetched again for each -- the response was 11 % of that task's traffic.
    unsigned grp = vb % npg, bix = vb / npg;
    if (npg >= 64 && npg % 8 == 0 && gridDim.x % 8 == 0) {
        const unsigned per = gridDim.x / 8, x = vb / per, l = vb - x * per, gpx = npg / 8;
        grp = x * gpx + l % gpx;
        bix = l / gpx;
    }
    const int sp = grp * PP + pl;
    const OsmBlock blk = osm_block(ch, bix);
    const float2* src = in + ((blk.in_off + tau) * S + 2 * sp);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = ld_ext(src + (long long)T * j * S);
    wg_fft<N, -1, CM, 0, BBT_SMALL_TW_POW>(v, lds, tau, pl, tw0, tw1);
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    const cf* h0 = resp + (long long)c0 * N + tau;
    const cf* h1 = resp + (long long)c1 * N + tau;
    apply_resp<T>(v, h0, h1, c0 == c1);
    wg_fft<N, +1, CM, 0, BBT_SMALL_TW_POW>(v, lds, tau, pl, tw0, tw1);
    if (blk.flat) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long e = (long long)(tau + T * j - blk.valid_start) * S + 2 * sp - blk.flat_sub;
            if (e >= 0 && e < blk.valid_count) st_ext(out + (blk.out_off + e), v[j]);
        }
        return;
    }
    if (ch.out_plane) {
        float2* dst = out + ((long long)sp * ch.out_plane + blk.out_off) * 2;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int r = tau + T * j - blk.valid_start;
            if (r >= 0 && r < blk.valid_count) st_ext(dst + (long long)r * 2, v[j]);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = tau + T * j - blk.valid_start;
        if (r >= 0 && r < blk.valid_count) st_ext(out + ((blk.out_off + r) * S + 2 * sp), v[j]);
    }
}

// Column pass, N1 == 16: one thread per 2-stream column, radix-16 in registers.
//   FIRST: stream -> work (forward).  !FIRST: work -> valid output (inverse).
// PP > 1 (many streams): the lanes run over PP stream pairs first, then over 256 / PP columns --
// PP * 16 contiguous bytes of every complete sample on the stream side (a whole line for 8
// pairs, where one pair per workgroup takes 16 bytes out of every S * 8-byte row) and
// 256 / PP * 16-byte runs of the work buffer (1024 sub-bands x 2 pol on 2^16-sample blocks,
// the CHIME-native form of config 4: Dedisperse 49.9 G stream-samples/s with PP = 1).
template <bool FIRST, bool SPEC = false, bool SINGLE = false, int PP = 1>
__global__ __launch_bounds__(256) void k_osm_col16(const float2* __restrict__ in,
                                                   float2* __restrict__ out,
                                                   float2* __restrict__ work, OsmChunk ch, int S,
                                                   int N2, SpecOut so) {
    const int npair = SINGLE ? 1 : S >> 1;
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int COLS = 256 / PP;
    const int npg = npair / PP;                   // groups of PP pairs (npair % PP == 0)
    const int n2 = (vb / npg) * COLS + threadIdx.x / PP;
    const int b = blockIdx.y, sp = (vb % npg) * PP + threadIdx.x % PP;
    float2* w = work + ((long long)(b * npair + sp) * 16) * N2 * 2 + (long long)n2 * 2;
    c2 v[16];
    if constexpr (SINGLE) {                       // b = pair of blocks of the one stream
        const SinglePair pr = single_pair(ch, b);
        if (FIRST) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = ld_single_shifted(in, pr, (long long)j * N2 + n2, 16ll * N2);
            radix16<-1>(v);
#pragma unroll
            for (int j = 0; j < 16; ++j) st_int(w + (long long)j * N2 * 2, v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_int(w + (long long)j * N2 * 2);
            radix16<+1>(v);
            if constexpr (SPEC) {
                const SpecCursor ca = spec_cursor(out, so, pr.a, 0, 1, N2, n2, 1, 0, 1);
                const SpecCursor cb = spec_cursor(out, so, pr.b, 0, 1, N2, n2, 1, 0, 1);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    emit_spectrum_single(make_float2(v[j].re.x, v[j].im.x), ca, j);
                    if (pr.has_b) emit_spectrum_single(make_float2(v[j].re.y, v[j].im.y), cb, j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) st_single(out, pr, (long long)j * N2 + n2, v[j]);
            }
        }
        return;
    }
    const OsmBlock blk = osm_block(ch, b);
    if (FIRST) {
        // blk.shift (fused channelizer): the block is read circularly shifted, element
        // e from sample (e + shift) mod N -- a circular convolution commutes with it.
        // shift < n_chan <= N2, so only the last row can wrap.
        const float2* src = in + ((blk.in_off + n2 + blk.shift) * S + 2 * sp);
        const long long wrap = (n2 + blk.shift >= N2) ? (long long)16 * N2 * S : 0;
        if (S == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_ext_nt(src + (long long)j * N2 * S - (j == 15 ? wrap : 0));
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_ext(src + (long long)j * N2 * S - (j == 15 ? wrap : 0));
        }
        radix16<-1>(v);
#pragma unroll
        for (int j = 0; j < 16; ++j) st_int(w + (long long)j * N2 * 2, v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_int(w + (long long)j * N2 * 2);
        radix16<+1>(v);
        SpecCursor cur;
        if (SPEC) cur = spec_cursor(out, so, blk, 0, 1, N2, small_channel_slot(n2, so), S, sp, npair);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (SPEC) {
                emit_spectrum(v[j], cur, j);
            } else {
                const int r = j * N2 + n2 - blk.valid_start;
                if (r >= 0 && r < blk.valid_count)
                    st_ext(out + ((blk.out_off + r) * S + 2 * sp), v[j], S == 2);
            }
        }
    }
}

// Column pass, N1 == 256, 512 or 1024: FCOL two-stream columns (f, fastest lane index ->
// FCOL * 16-byte runs per row: 256 B for 16) x T1 = N1 / 16 threads per N1-point
// transform; FCOL * T1 threads per workgroup.  (512 / 1024 points: blocks of 2^21 / 2^22
// samples as N1 x 4096 -- three passes where the three-level scheme takes five; 8 columns per
// workgroup, 48 / 80 KiB of exchange area, so that three / two workgroups share a CU.)
// With many streams the FCOL lanes of a row can instead cover PP pairs x
// FCOL / PP columns (PP > 1): the stream side then moves PP * 16 contiguous
// bytes per complete sample (a whole 128-byte line for 8 pairs) at the price
// of FCOL / PP * 16-byte runs on the work-buffer side.
template <bool FIRST, bool SPEC, int FCOL, bool DET = false, int PP = 1, bool SINGLE = false, int N1 = 256>
__global__ __launch_bounds__(FCOL * (N1 / 16)) void k_osm_col256(const float2* __restrict__ in,
                                                          float2* __restrict__ out,
                                                          float2* __restrict__ work, OsmChunk ch,
                                                          int S, int N2, const cf* __restrict__ tw0,
                                                          SpecOut so, const cf* __restrict__ tw1 = nullptr) {
    typedef FftGeo<N1> G;
    constexpr int T1 = N1 / 16;                          // threads per column: row k1 = tau + T1 j
    static_assert(!DET || N1 == 256, "fused detection: 256-point columns");
    extern __shared__ v2 col256_lds[];                   // G::LDS_ELEMS * FCOL elements
    v2* lds = col256_lds;
    static_assert(!DET || PP == 1, "fused detection needs the lanes of a row in one stream pair");
    const int f = threadIdx.x % FCOL, tau = threadIdx.x / FCOL;
    const int npair = SINGLE ? 1 : S >> 1;
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int CPT = FCOL / PP;                       // columns per tile
    const int npg = npair / PP;                          // pair groups
    const int n2 = (vb / npg) * CPT + f / PP;
    const int b = blockIdx.y, sp = (vb % npg) * PP + f % PP;
    // work element (k1, n2) of this (block, pair): 16 bytes at ((b*npair+sp)*N1 + k1)*N2 + n2
    float2* w = work + (((long long)(b * npair + sp) * N1 + tau) * N2 + n2) * 2;
    c2 v[16];
    if constexpr (SINGLE) {                              // b = pair of blocks of the one stream
        static_assert(!DET && PP == 1, "one-stream plans: no fused detection");
        const SinglePair pr = single_pair(ch, b);
        if (FIRST) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = ld_single_shifted(in, pr, (long long)(tau + T1 * j) * N2 + n2, (long long)N1 * N2);
            wg_fft<N1, -1, FCOL, 0, BBT_COL_TW_POW>(v, lds, tau, f, tw0, tw1);
            if (so.twa) col_twiddles<-1>(v, so.twa, so.twg, tau, n2, N2);
#pragma unroll
            for (int j = 0; j < 16; ++j) st_int(w + (long long)T1 * j * N2 * 2, v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_int(w + (long long)T1 * j * N2 * 2);
            if (so.twa) col_twiddles<+1>(v, so.twa, so.twg, tau, n2, N2);
            wg_fft<N1, +1, FCOL, 0, BBT_COL_TW_POW>(v, lds, tau, f, tw0, tw1);
            if constexpr (SPEC) {
                const SpecCursor ca = spec_cursor(out, so, pr.a, tau, T1, N2, n2, 1, 0, 1);
                const SpecCursor cb = spec_cursor(out, so, pr.b, tau, T1, N2, n2, 1, 0, 1);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    emit_spectrum_single(make_float2(v[j].re.x, v[j].im.x), ca, j);
                    if (pr.has_b) emit_spectrum_single(make_float2(v[j].re.y, v[j].im.y), cb, j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) st_single(out, pr, (long long)(tau + T1 * j) * N2 + n2, v[j]);
            }
        }
        return;
    }
    const OsmBlock blk = osm_block(ch, b);
    if (FIRST) {
        // read circularly shifted by blk.shift (see k_osm_col16): the last row can wrap
        // (pair-planar input: this pair's samples are an array of their own, see OsmChunk)
        const int Si = ch.in_plane ? 2 : S;
        const float2* src = in + (ch.in_plane ? (long long)sp * ch.in_plane * 2 : 2 * sp) +
                            (blk.in_off + (long long)tau * N2 + n2 + blk.shift) * Si;
        const long long wrap = (tau == T1 - 1 && n2 + blk.shift >= N2) ? (long long)N1 * N2 * Si : 0;
        if (Si == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = ld_ext_nt(src + (long long)T1 * j * N2 * 2 - (j == 15 ? wrap : 0));
        } else if (S == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = ld_ext_nt(src + (long long)T1 * j * N2 * S - (j == 15 ? wrap : 0));
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = ld_ext(src + (long long)T1 * j * N2 * S - (j == 15 ? wrap : 0));
        }
        wg_fft<N1, -1, FCOL, 0, BBT_COL_TW_POW>(v, lds, tau, f, tw0, tw1);
        if (so.twa) col_twiddles<-1>(v, so.twa, so.twg, tau, n2, N2);
#pragma unroll
        for (int j = 0; j < 16; ++j) st_int(w + (long long)T1 * j * N2 * 2, v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_int(w + (long long)T1 * j * N2 * 2);
        if (so.twa) col_twiddles<+1>(v, so.twa, so.twg, tau, n2, N2);
        wg_fft<N1, +1, FCOL, 0, BBT_COL_TW_POW>(v, lds, tau, f, tw0, tw1);
        SpecCursor cur;
        if (SPEC) cur = spec_cursor(out, so, blk, tau, T1, N2, small_channel_slot(n2, so), S, sp, npair);
        if constexpr (SPEC && DET) {
            // Detection + integration instead of storing spectra.  The workgroup
            // holds, for each of its FCOL channels, every (N2 / n_chan)-th of
            // 256 * N2 / n_chan consecutive spectra: row n1 = tau + 16 j is
            // spectrum s_wg + n1 * spr.  The powers go to LDS as P[n1][f] (two
            // components at a time: 32 KiB), then one thread per (bin, channel,
            // component) sums the rows of its bin -- they are contiguous -- and
            // adds the result to the output once.
            constexpr int NT = FCOL * 16;
            float2* pw = reinterpret_cast<float2*>(lds);            // [256][FCOL]
            static_assert(256 * FCOL * 8 <= G::LDS_ELEMS * FCOL * (int)sizeof(v2),
                          "power tile does not fit the exchange buffer");
            const int step = so.det_step, mode = so.det_mode;
            const int spr = N2 >> so.lg_chan;                        // spectra per row step
            const int lg_spr = __ffs(spr) - 1;
            const long long s_wg = cur.s0 - (long long)tau * spr;   // spectrum of row 0
            long long b_wg = s_wg / step;
            if (b_wg * step > s_wg) --b_wg;                          // floor: bin of row 0
            const long long n_bins = cur.n_out / step;
            const int ch0 = (n2 - f) & (cur.nch - 1);
            const int nb = (int)((s_wg + 255ll * spr) / step - b_wg) + 1;   // local bins (<= BBT_DET_MAX_BINS)
            float4 p[16];
            unsigned whole = 0;                                      // (bit j: p[j] is a spectrum of this call)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int rel = cur.rel + j * cur.drel;
                bool full = false;
                if (rel >= cur.wrap_at) {                        // see emit_spectrum
                    if (rel < cur.vc) st_ext(cur.seam1, v[j]);
                    rel -= cur.n_fft;
                    if (rel < 0 && rel + cur.nch > 0) st_ext(cur.seam0, v[j]);
                } else if (rel >= 0 && rel + cur.nch <= cur.vc) {
                    full = true;
                } else if (rel < 0 && rel + cur.nch > 0) {
                    st_ext(cur.seam0, v[j]);
                } else if (rel < cur.vc && rel + cur.nch > cur.vc) {
                    st_ext(cur.seam1, v[j]);
                }
                const long long sj = cur.s0 + j * cur.ds;
                const bool mine = full && sj >= 0 && sj < cur.n_out;
                whole |= (unsigned)mine << j;
                p[j] = mine ? detect_pair(v[j], mode) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (step == 1) {
                // Square / Power without integration: every power is final, so it is stored where the
                // spectrum would have gone -- no sums, no atomics, no zeroed output
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (!((whole >> j) & 1)) continue;
                    float* dst = so.det + detect_index(cur.s0 + j * cur.ds, ch0 + f, sp, so.lg_chan, npair, mode);
                    if (mode)
                        *reinterpret_cast<float4*>(dst) = make_float4(p[j].x * so.det_scale, p[j].y * so.det_scale,
                                                                      p[j].z * so.det_scale, p[j].w * so.det_scale);
                    else
                        *reinterpret_cast<float2*>(dst) = make_float2(p[j].x * so.det_scale, p[j].y * so.det_scale);
                }
                return;
            }
            const int npass = mode ? 2 : 1;
            for (int pass = 0; pass < npass; ++pass) {
                __syncthreads();                                     // exchange buffer / previous pass done
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    pw[(tau + 16 * j) * FCOL + f] = pass ? make_float2(p[j].z, p[j].w)
                                                         : make_float2(p[j].x, p[j].y);
                __syncthreads();
                const float* pf = reinterpret_cast<const float*>(pw);
                for (int i = threadIdx.x; i < nb * FCOL * 2; i += NT) {
                    const int bl = i / (FCOL * 2), fc = i - bl * (FCOL * 2);          // fc = 2 f + comp
                    const long long bin = b_wg + bl;
                    if (bin < 0 || bin >= n_bins) continue;
                    // rows n1 with bin * step <= s_wg + n1 * spr < (bin + 1) * step  (spr = 2^k)
                    const int d0 = (int)(bin * step - s_wg);        // > -step, <= 255 * spr
                    int lo = d0 > 0 ? (d0 + spr - 1) >> lg_spr : 0;
                    int hi = (d0 + step + spr - 1) >> lg_spr;
                    if (hi > 256) hi = 256;
                    float sum = 0.f;
                    for (int n1 = lo; n1 < hi; ++n1) sum += pf[n1 * (FCOL * 2) + fc];
                    if (sum != 0.f)
                        unsafeAtomicAdd(so.det + detect_index(bin, ch0 + (fc >> 1), sp, so.lg_chan, npair, mode)
                                            + 2 * pass + (fc & 1),
                                        sum * so.det_scale);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (SPEC) {
                emit_spectrum(v[j], cur, j);
            } else {
                const long long r = (long long)(tau + T1 * j) * N2 + n2 - blk.valid_start;
                if (r >= 0 && r < blk.valid_count)
                    st_ext(out + ((blk.out_off + r) * S + 2 * sp), v[j], S == 2);
            }
        }
    }
}

// Middle level of a three-level transform N = 256 * 16 * N2 (N > 2^20), in
// place on the work buffer.  After the outer 256-point column pass the work
// buffer holds, per (block, pair), rows k1o of M = 16 * N2 elements indexed
// m = N2 * a + n2.  FWD: multiply by the outer four-step twiddle W_N^{m k1o}
// and transform over a (radix-16 in registers); !FWD: the inverse, twiddle
// conjugated and applied after the butterfly.  One thread per (row, n2).
template <bool FWD>
__global__ __launch_bounds__(256) void k_osm_mid16(float2* __restrict__ work, int N2, int n_fft,
                                                   const cf* __restrict__ wroot, int skip_base, int y0) {
    // (y0: first outer row of this launch -- the middle passes run over pieces of the work
    // buffer that stay in the Infinity Cache from one pass to the next)
    const int n2 = blockIdx.x * 256 + threadIdx.x;
    const int yrow = blockIdx.y + y0;
    const int k1o = yrow & 255;
    float2* w = work + ((long long)yrow * 16 * N2 + n2) * 2;
    // W_N^{(N2 a + n2) k1o} = W_N^{n2 k1o} * W_4096^{a k1o}      (N / N2 = 4096)
    cf base;
    {
        float s, c;
        sincospif(-2.0f * (float)(n2 * k1o) / (float)n_fft, &s, &c);
        base = skip_base ? make_float2(1.f, 0.f) : make_float2(c, s);   // (fused: done in the row pass)
    }
    c2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = ld_int(w + (long long)a * N2 * 2);
    if (FWD) {
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = twmul<-1>(v[a], cmul(base, wroot[(a * k1o) & 4095]));
        radix16<-1>(v);
    } else {
        radix16<+1>(v);
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = twmul<+1>(v[a], cmul(base, wroot[(a * k1o) & 4095]));
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) st_int(w + (long long)a * N2 * 2, v[a]);
}

// value of lane (id ^ H) for H = 1, 2 (DPP quad_perm) or 4 (ds_swizzle bit mode)
template <int H>
__device__ __forceinline__ float lane_xor(float x) {
    int i = __float_as_int(x);
    if constexpr (H == 1) i = __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (H == 2) i = __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else i = __builtin_amdgcn_ds_swizzle(i, (H << 10) | 0x1F);                                 // xor H within 32 lanes
    return __int_as_float(i);
}
// one radix-2 DIF stage over the lane digit c (mask H) of an L-point transform spread over L lanes
template <int H, int L>
__device__ __forceinline__ void lane_radix2_stage(c2 (&v)[16], int c, const cf* __restrict__ wroot) {
    const bool upper = (c & H) != 0;
    const cf w = wroot[((c & (H - 1)) * (L / (2 * H))) * (4096 / L)];     // W_{2H}^{c mod H}
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        c2 o;
        o.re.x = lane_xor<H>(v[a].re.x);
        o.re.y = lane_xor<H>(v[a].re.y);
        o.im.x = lane_xor<H>(v[a].im.x);
        o.im.y = lane_xor<H>(v[a].im.y);
        if (upper) v[a] = twmul<-1>(csub(o, v[a]), w);
        else v[a] = cadd(v[a], o);
    }
}

// Row pass: for row k1 of a (block, pair): four-step twiddle, forward FFT
// over n2, multiply by the response, inverse FFT over k2, conjugate twiddle.
//   wroot : W_4096^m, m in [0, 4096)
//
// NCH > 0 fuses Channelize.task (reference channelize.py:73-74) into this
// pass.  The channelizer transforms groups of NCH consecutive dedispersed
// samples, i.e. acts along n2 inside a row, while the remaining inverse step
// (column pass) acts along k1: the two commute, so the NCH-point FFT is done
// here, on registers, and the column pass then emits spectra directly.  To
// make the groups start at multiples of NCH in n2, the block is circularly
// shifted by `shift` samples: the first column pass reads it that way (a
// circular convolution commutes with a circular shift), which costs address
// arithmetic there instead of a phase ramp exp(+2 pi i k shift/N) here.
// Spectra that straddle the first / last kept sample of a block are spliced
// afterwards (k_seam_fix).
#ifndef BBT_ROWPASS_TWO_REGIONS
#define BBT_ROWPASS_TWO_REGIONS 0
#endif
#ifndef BBT_ROWPASS_MINWAVES
#define BBT_ROWPASS_MINWAVES 1
#endif
// (Capping the plain row pass at 168 VGPRs with launch bounds spilled 34 dwords
// and was 12 % slower, 35.7 against 40.7 Gsamples/s for config 2; the register
// count was brought down at the source instead, see the end of the kernel.)
#ifndef BBT_ROWPASS_PLAIN_MINWAVES
#define BBT_ROWPASS_PLAIN_MINWAVES 1
#endif
// The few-channel variants (32..128 channels) sit at 170-176 VGPRs; asking for
// 3 waves per SIMD costs 2-11 spilled dwords.
#ifndef BBT_ROWPASS_SMALL_MINWAVES
#define BBT_ROWPASS_SMALL_MINWAVES 3
#endif
template <int N2, int NCH>
__global__ __launch_bounds__(N2 / 16, (BBT_ROWPASS_MINWAVES > 1 ? BBT_ROWPASS_MINWAVES : ((NCH == 0 && N2 >= 1024) ? BBT_ROWPASS_PLAIN_MINWAVES : ((NCH > 16 && NCH < 256) ? BBT_ROWPASS_SMALL_MINWAVES : 1)))) void k_osm_rowpass(
    float2* __restrict__ work, int N1, const cf* __restrict__ resp,
    const int* __restrict__ resp_index, int npair, const cf* __restrict__ tw0,
    const cf* __restrict__ tw1, const cf* __restrict__ wroot,
    OsmChunk ch, int outer, int y0, const cf* __restrict__ tw4row, const cf* __restrict__ tw4base,
    int tw_in_col, const cf* __restrict__ tw4o, const cf* __restrict__ tw4u) {
    // Three-level transforms (N > 2^20) run this pass once per row k1o of the
    // outer 256-point level: blockIdx.y = (block * npair + pair) * outer + k1o;
    // the full frequency index is k = k1o + outer * (k1 + N1 * k2).  outer == 1
    // is the plain two-level case.
    typedef FftGeo<N2> G;
    constexpr int T = G::T;
    constexpr int IMOFF = BBT_ROWPASS_TWO_REGIONS ? G::LDS_ELEMS : 0;
    __shared__ v2 lds[G::LDS_ELEMS + IMOFF];
    const int tau = threadIdx.x;
    // Two launch shapes.  grid (N1, blocks * pairs * outer): row k1 = blockIdx.x.
    // grid (N1 * blocks * pairs, 1) (outer == 1 only): the workgroups that share
    // response row k1 -- the same row of every block of the chunk -- get
    // neighbouring ids on one XCD, so the response is read from HBM once per
    // launch and then from that XCD's L2 (it was re-read 2.6 times per launch
    // through the Infinity Cache).
    int k1, by;
    if (gridDim.y == 1 && outer == 1) {
        const int nbp = ch.nblk * npair;
        const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
        k1 = vb / nbp;
        by = vb - k1 * nbp;
    } else {
        k1 = blockIdx.x;
        by = blockIdx.y + y0;          // (y0: first outer row of this launch, three-level plans)
    }
    const int k1o = by % outer;
    const int bp = by / outer;                    // block * npair + pair
    const int sp = bp % npair;
    float2* row = work + (((long long)by * N1 + k1) * N2) * 2;
    c2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = ld_int(row + (long long)(tau + T * j) * 2);
    // W_N^{k1 (tau + T j)} = W_N^{k1 tau} * W_{16 N1}^{k1 j},  N = N1 * N2.  Two-level plans bring
    // both factors as tables (tw4base [N1][T] per thread, tw4row [N1][16] per row: its 16 values
    // are one or two wide scalar loads, where evaluating them from the W_4096 table was sixteen
    // scalar loads one waiting for the other, and the thread's factor a sincospif each way).
    // tw_in_col (two-level plans with tables): bit 0, the first column pass has applied the
    // forward twiddles; bit 1, the last column pass will apply the inverse ones (col_twiddles;
    // not with the fused channelizer, whose transform over n2 comes after them)
    cf base = make_float2(1.f, 0.f);
    if (!(tw_in_col & 1)) {
        base = tw4base[k1 * T + tau];
        const cf* tr = tw4row + k1 * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = twmul<-1>(v[j], cmul(base, tr[j]));
    }
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    const cf* h0 = resp + (((long long)c0 * outer + k1o) * N1 + k1) * N2 + tau;
    const cf* h1 = resp + (((long long)c1 * outer + k1o) * N1 + k1) * N2 + tau;
    wg_fft<N2, -1, 0, IMOFF, BBT_ROWPASS_TW_POW>(v, lds, tau, 0, tw0, tw1);
    // keep the response and second-transform table loads from being hoisted above
    // the first transform (they were for NCH == 0: 199 VGPRs, 2 waves per SIMD)
    __builtin_amdgcn_sched_barrier(0);
    apply_resp<T>(v, h0, h1, c0 == c1);
    __builtin_amdgcn_sched_barrier(0);
    {
        // The second transform reads the same twiddle tables as the first: passed
        // through an opaque move, or the compiler keeps all 30 table values of the
        // first alive in registers across the whole kernel instead of reloading
        // them from L1.
        const cf* tw0b = tw0;
        const cf* tw1b = tw1;
        asm volatile("" : "+s"(tw0b), "+s"(tw1b));
        wg_fft<N2, +1, 0, IMOFF, BBT_ROWPASS_TW_POW>(v, lds, tau, 0, tw0b, tw1b);
    }
    if (NCH > 0 && outer > 1) {
        // Three-level + fused channelizer: the outer four-step twiddle
        // W_N^{(N2 a + n2) k1o} has a factor that depends on n2; it must act
        // before the channel FFT, so it is applied here (its a-dependent factor
        // stays in k_osm_mid16).  W_N^{(tau + T j) k1o} = W_N^{tau k1o} W_65536^{j k1o}.
        // from tables (three-level plans, bbt_osm_plan_create): tw4o [outer][T] = W_N^{tau k1o};
        // tw4u [outer][N1][16] = W_{16 N1}^{k1 j} W_65536^{k1o j}, the same for the whole
        // workgroup -- where a sincospif, forty-eight scalar-valued table loads and thirty-two
        // complex products per thread were
        const cf bb = cmul(base, tw4o[k1o * T + tau]);
        const cf* uu = tw4u + ((long long)k1o * N1 + k1) * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = twmul<+1>(v[j], cmul(bb, uu[j]));
    } else if (tw_in_col & 2) {
    } else {
        // (the tables again, through opaque moves: see below)
        const cf* tr = tw4row + k1 * 16;
        const cf* tb = tw4base;
        asm volatile("" : "+s"(tr), "+s"(tb));
        const cf base2 = tb[k1 * T + tau];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = twmul<+1>(v[j], cmul(base2, tr[j]));
    }
    // (Stores that go back to the addresses the row was loaded from take the row
    // pointer through an opaque move: otherwise the 16 load addresses, 32 VGPRs,
    // stay alive from the first instruction to the last -- that, with the tables
    // above, is what made the plain row pass 199 VGPRs / 2 waves per SIMD and
    // plain Dedisperse slower than the fused pipeline; now 148 / 3.)
    if constexpr (NCH == 0) {
        float2* row2 = row;
        asm volatile("" : "+s"(row2));
#pragma unroll
        for (int j = 0; j < 16; ++j) st_int(row2 + (long long)(tau + T * j) * 2, v[j]);
    } else if constexpr (NCH < 16) {
        // Very few channels, NCH = 2, 4, 8 (Channelize(4) behind a many-stream Dedisperse: the
        // CHIME-native form of config 4, reference channelize.py:73-74): register j of thread tau is
        // n2 = tau + T j, so the NCH consecutive samples of a group sit in the same register of NCH
        // neighbouring LANES (groups start at multiples of NCH, T is a multiple of 16).  The
        // transform is the lane butterflies alone -- no exchange, no register stage --: lane c of a
        // group ends up with channel bitrev(c), stored where the sample was; the column pass maps
        // the position back (small_channel_slot, SpecOut::tiny_lg).
        static_assert(NCH == 2 || NCH == 4 || NCH == 8, "NCH below 16: 2, 4 or 8");
        const int c = tau & (NCH - 1);
        if constexpr (NCH >= 8) lane_radix2_stage<4, NCH>(v, c, wroot);
        if constexpr (NCH >= 4) lane_radix2_stage<2, NCH>(v, c, wroot);
        lane_radix2_stage<1, NCH>(v, c, wroot);
        float2* row2 = row;
        asm volatile("" : "+s"(row2));
#pragma unroll
        for (int j = 0; j < 16; ++j) st_int(row2 + (long long)(tau + T * j) * 2, v[j]);
    } else if constexpr (NCH < 256) {
        // Few channels, NCH = 16 L (L = 1, 2, 4, 8): a group of NCH consecutive
        // samples n2 is spread over the lanes (register j of thread tau is
        // n2 = tau + T j), so one more exchange through LDS hands thread
        // t' = L q + c the 16 samples c + L i (i < 16) of group q; radix-16 over
        // i on registers, twiddle W_NCH^{c a}, then the radix-L step over c
        // across L neighbouring lanes with wavefront butterflies (DPP quad_perm /
        // ds_swizzle moves, lane_radix2_stage).  Lane c ends up
        // with channels a + 16 bitrev_L(c); they are stored at row position
        // t' + T a (coalesced), which the column pass maps back
        // (small_channel_slot below).  LDS pitch: L extra slots per group, so the
        // strided reads are conflict free.
        constexpr int L = NCH / 16;
        static_assert(NCH == 16 * L && (L == 1 || L == 2 || L == 4 || L == 8), "NCH must be 16..128");
        static_assert(N2 + N2 / 16 <= G::LDS_ELEMS, "exchange area too small");
        const int q = tau / L, c = tau % L;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            // element n2 sits at n2 + (n2 >> 4); written out so that both address sets are one
            // per-thread base plus compile-time offsets (T and 16 L are multiples of 16, c < L)
            static_assert(T % 16 == 0, "row threads must be a multiple of 16");
            const int wbase = tau + (tau >> 4), rbase = 17 * L * q + c;
#pragma unroll
            for (int j = 0; j < 16; ++j) lds[wbase + T * j + (T * j) / 16] = half ? v[j].im : v[j].re;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const v2 x = lds[rbase + L * i + (L * i) / 16];
                if (half) v[i].im = x; else v[i].re = x;
            }
        }
        radix16<-1>(v);
        if constexpr (L > 1) {
            // (unsigned index: scalar base + 32-bit lane offset, no 64-bit address per load;
            // four loads in flight at a time instead of fifteen)
#if BBT_ROWPASS_TW_POW
            {   // W_NCH^{c a}, a < 16, from four loads (a = 1, 2, 4, 8) and products (fft_twiddle_powers4)
                cf w[15];
                const unsigned step = (unsigned)c * (unsigned)(4096 / NCH);
                w[0] = wroot[step];
                w[1] = wroot[2 * step];
                w[3] = wroot[4 * step];
                w[7] = wroot[8 * step];
                fft_twiddle_powers4(w);
#pragma unroll
                for (int a = 1; a < 16; ++a) v[a] = twmul<-1>(v[a], w[a - 1]);
            }
#else
#pragma unroll
            for (int a = 1; a < 16; ++a) {
                v[a] = twmul<-1>(v[a], wroot[(unsigned)(c * a) * (unsigned)(4096 / NCH)]);
                if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#endif
            // DIF over the lane digit c: masks L/2, ..., 2, 1 -- within a quad by DPP
            // quad_perm moves, across quads (L = 8) by ds_swizzle; no address registers
            if constexpr (L >= 8) lane_radix2_stage<4, L>(v, c, wroot);
            if constexpr (L >= 4) lane_radix2_stage<2, L>(v, c, wroot);
            lane_radix2_stage<1, L>(v, c, wroot);
        }
        float2* row2 = row;
        asm volatile("" : "+s"(row2));
#pragma unroll
        for (int a = 0; a < 16; ++a) st_int(row2 + (long long)(tau + T * a) * 2, v[a]);
    } else {
        // channelizer: thread tau holds n2 = tau + T j: group q = j / P, element
        // m = tau + T (j % P) of the NCH = P * T point transform.
        constexpr int NG = (NCH > 0) ? N2 / NCH : 1;     // groups per row
        constexpr int P = 16 / NG;                       // points per thread per group
        static_assert(NCH < 256 || (NG * NCH == N2 && P * NG == 16), "NCH must divide N2, N2/NCH <= 16");
        // stage 0: radix-P over the P points of each group, twiddle W_NCH^{tau c} = tw0[(c NG) T + tau]
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if constexpr (P > 1) {
                c2 t[P];
#pragma unroll
                for (int i = 0; i < P; ++i) t[i] = v[q * P + i];
                radixR<-1, P>(t);
#pragma unroll
                for (int i = 0; i < P; ++i) v[q * P + i] = t[i];
            }
        }
#pragma unroll
        for (int c = 1; c < P; ++c) {
            const cf w = tw0[(c * NG) * T + tau];
#pragma unroll
            for (int q = 0; q < NG; ++q) v[q * P + c] = twmul<-1>(v[q * P + c], w);
        }
        // the 16 sequences f = q P + c, one element b = tau each: T-point transforms over b
        wg_fft_tail<N2, -1, 0, IMOFF, BBT_ROWPASS_TW_POW>(v, lds, tau, 0, tw1);
        // thread tau2: register u + NU c2 holds k' = (g + R2 u) + 16 c2 of sequence f = tau2 & 15
        constexpr int R2 = G::R2, NU = 16 / R2;
        const int f = tau & 15, g = tau >> 4;
        const int q = f / P, c = f - q * P;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int c2i = 0; c2i < R2; ++c2i) {
                const int kp = g + R2 * u + 16 * c2i;
                st_int(row + (long long)(q * NCH + c + P * kp) * 2, v[u + NU * c2i]);
            }
    }
}

// Splice the spectrum that straddles the seam between two consecutive blocks:
// both blocks computed it from their own (circular) data; inverse-transform
// both, take samples [0, L) from the earlier block and [L, NCH) from the later
// one, transform again.  One workgroup per (seam, pair).
struct SeamJob {
    long long spectrum;   // output spectrum index (relative to out[0])
    int first_block;      // call-wide index of the earlier block
    int split;            // L
};
#define BBT_SEAM_JOBS_PER_LAUNCH 64
struct SeamJobs {         // passed by value: no upload, no synchronisation
    SeamJob j[BBT_SEAM_JOBS_PER_LAUNCH];
};
template <int NCH, bool SINGLE = false>
__global__ __launch_bounds__(NCH / 16) void k_seam_fix(const float2* __restrict__ seam,
                                                       float2* __restrict__ out, SeamJobs jobs,
                                                       int S, int npair,
                                                       const cf* __restrict__ tw0,
                                                       const cf* __restrict__ tw1, SpecOut so) {
    typedef FftGeo<NCH> G;
    constexpr int T = G::T;
    __shared__ v2 lds[G::LDS_ELEMS];
    const int tau = threadIdx.x, sp = blockIdx.y;
    const SeamJob job = jobs.j[blockIdx.x];
    const float2* za = seam + ((((long long)job.first_block * 2 + 1) * npair + sp) * NCH + tau) * 2;
    const float2* zb = seam + ((((long long)(job.first_block + 1) * 2 + 0) * npair + sp) * NCH + tau) * 2;
    c2 va[16], vb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) va[j] = ld_ext(za + (long long)T * j * 2);
    wg_fft<NCH, +1, 0>(va, lds, tau, 0, tw0, tw1);
#pragma unroll
    for (int j = 0; j < 16; ++j) vb[j] = ld_ext(zb + (long long)T * j * 2);
    wg_fft<NCH, +1, 0>(vb, lds, tau, 0, tw0, tw1);
    const float scale = 1.0f / (float)NCH;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool early = tau + T * j < job.split;
        va[j].re = (early ? va[j].re : vb[j].re) * scale;
        va[j].im = (early ? va[j].im : vb[j].im) * scale;
    }
    wg_fft<NCH, -1, 0>(va, lds, tau, 0, tw0, tw1);
    if (so.det) {                               // fused detection: add the spectrum's power to its bin
        const long long bin = job.spectrum / so.det_step;
        if (bin >= so.n_out / so.det_step) return;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 pw = detect_pair(va[j], so.det_mode);
            float* dst = so.det + detect_index(bin, tau + T * j, sp, so.lg_chan, npair, so.det_mode);
            if (so.det_step == 1) {                          // (no integration: the value itself)
                if (so.det_mode)
                    *reinterpret_cast<float4*>(dst) = make_float4(pw.x * so.det_scale, pw.y * so.det_scale,
                                                                  pw.z * so.det_scale, pw.w * so.det_scale);
                else
                    *reinterpret_cast<float2*>(dst) = make_float2(pw.x * so.det_scale, pw.y * so.det_scale);
                continue;
            }
            unsafeAtomicAdd(dst + 0, pw.x * so.det_scale);
            unsafeAtomicAdd(dst + 1, pw.y * so.det_scale);
            if (so.det_mode) {
                unsafeAtomicAdd(dst + 2, pw.z * so.det_scale);
                unsafeAtomicAdd(dst + 3, pw.w * so.det_scale);
            }
        }
        return;
    }
    if constexpr (SINGLE) {                              // one stream: the first half of the pair format
        float2* dst = out + (job.spectrum * NCH + tau);
#pragma unroll
        for (int j = 0; j < 16; ++j) dst[T * j] = make_float2(va[j].re.x, va[j].im.x);
        return;
    }
    float2* dst = out + ((job.spectrum * NCH + tau) * S + 2 * sp);
#pragma unroll
    for (int j = 0; j < 16; ++j) st_ext(dst + (long long)T * j * S, va[j]);
}

// Response H[c][k] (natural FFT order) -> Hperm[c][k1o][k1][k2] * scale with
// k = k1o + outer * (k1 + N1 * k2)  (outer == 1: two-level layout [c][k1][k2]).
__global__ void k_permute_resp(const cf* __restrict__ h, cf* __restrict__ hp, int outer, int N1,
                               long long N2, float scale) {
    const long long n = (long long)outer * N1 * N2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const long long k2 = idx % N2, r = idx / N2;
    const long long k1 = r % N1, k1o = r / N1;
    const cf x = h[(long long)blockIdx.y * n + k1o + outer * (k1 + N1 * k2)];
    hp[(long long)blockIdx.y * n + idx] = make_float2(x.x * scale, x.y * scale);
}

// ---------------------------------------------------------------------------
// Polyphase filter bank: y[i, c] = sum_t x[(i + t) N + c] h[t, c], then FFT
// over c.  in: ((n_spec + n_tap - 1) * N, S); out: (n_spec * N, S); taps
// (n_tap, N) float32.  The n_tap-fold re-read of each input row is served by
// the XCD's L2 (xcd_remap keeps neighbouring spectra on one XCD).
template <int N, int FPW>
__global__ __launch_bounds__(FPW* N / 16) void k_pfb(const float2* __restrict__ in,
                                                      float2* __restrict__ out, long long n_spec,
                                                      int S, int n_tap,
                                                      const float* __restrict__ taps,
                                                      const cf* __restrict__ tw0,
                                                      const cf* __restrict__ tw1) {
    typedef FftGeo<N> G;
    constexpr int T = G::T;
    __shared__ v2 lds[FPW * G::LDS_ELEMS];
    const int slot = threadIdx.x / T, tau = threadIdx.x % T;
    const int npair = S >> 1;
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    const long long i = (long long)(vb / npair) * FPW + slot;
    const int sp = vb % npair;
    const bool active = i < n_spec;
    c2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = czero();
    if (active) {
        const float2* src = in + ((i * N + tau) * S + 2 * sp);
        for (int t = 0; t < n_tap; ++t) {
            const float* ht = taps + (long long)t * N + tau;
            const float2* st = src + (long long)t * N * S;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const c2 x = ld_ext(st + (long long)T * j * S);
                const float h = ht[T * j];
                v[j].re += x.re * h;
                v[j].im += x.im * h;
            }
        }
    }
    wg_fft<N, -1, 0, 0, BBT_PFB_TW_POW>(v, lds + slot * G::LDS_ELEMS, tau, 0, tw0, tw1);
    if (active) {
        float2* dst = out + ((i * N + tau) * S + 2 * sp);
#pragma unroll
        for (int j = 0; j < 16; ++j) st_ext(dst + (long long)T * j * S, v[j], S == 2);
    }
}

// Polyphase filter bank on MANY streams, first of two passes: the window alone,
//   y[i, c, s] = sum_t x[(i + t) N + c, s] h[t, c]          (reference pfb.py:91-100),
// as a streaming filter over whole rows -- a row is the N complete samples of one spectrum, N * S
// contiguous complex numbers, and thread `idx` owns one 16-byte piece (a stream pair of one
// column) of every row, so every access is a whole line whatever the stream count (the
// one-pass kernels below take 16 bytes out of every S * 8-byte sample: 330 G stream-samples/s
// for two streams, 144 / 99 / 86 for 16 / 128 / 2048).  A workgroup sweeps NI + NTAP - 1 rows for
// NI spectra, each row loaded once, NTAP running sums in registers (the sum of output i lives in
// register i mod NTAP).  The transform follows in place (k_fft_rows_pp).  Real taps: the external
// format (re_A im_A re_B im_B) is scaled as it is.
//   grid (ceil(row16 / 256), ceil(n_spec / NI));  row16 = N * S / 2 sixteen-byte pieces per row
#ifndef BBT_PFB_FIR_LB
#define BBT_PFB_FIR_LB 0                 // 0: all NTAP rows of a group in flight
#endif
template <int NTAP, int NI>
__global__ __launch_bounds__(256) void k_pfb_fir_rows(const float4* __restrict__ in, float4* __restrict__ out,
                                                      long long n_spec, long long row16, int npair, int N,
                                                      const float* __restrict__ taps) {
    static_assert(NI % NTAP == 0, "rows are swept in groups of NTAP");
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= row16) return;
    const int c = (int)(idx / npair);
    const long long i0 = (long long)blockIdx.y * NI;
    const long long rows = n_spec + NTAP - 1 - i0;           // input rows that exist from i0 on
    const long long outs = n_spec - i0;                       // spectra left from i0 on
    float h[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) h[t] = taps[(long long)t * N + c];
    f4v acc[NTAP];
#pragma unroll
    for (int k = 0; k < NTAP; ++k) acc[k] = f4v{0.f, 0.f, 0.f, 0.f};
    const f4v* src = reinterpret_cast<const f4v*>(in) + i0 * row16 + idx;
    f4v* dst = reinterpret_cast<f4v*>(out) + i0 * row16 + idx;
    // (the rows of a group in batches of LB loads in flight together: BBT_PFB_FIR_LB)
    constexpr int LBD = BBT_PFB_FIR_LB > 0 ? BBT_PFB_FIR_LB : 1;     // (no remainder by a literal zero)
    constexpr int LB = (BBT_PFB_FIR_LB > 0 && NTAP % LBD == 0 && BBT_PFB_FIR_LB < NTAP) ? BBT_PFB_FIR_LB : NTAP;
    for (int g = 0; g < NI / NTAP + 1; ++g) {
#pragma unroll
        for (int u0 = 0; u0 < NTAP; u0 += LB) {
            f4v x[LB];
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                const long long r = (long long)g * NTAP + u0 + i;
                x[i] = (r < rows && r < NI + NTAP - 1) ? src[r * row16] : f4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                const int u = u0 + i;
                const long long r = (long long)g * NTAP + u;
                // row r is tap t of output r - t: register (u - t) mod NTAP
#pragma unroll
                for (int t = 0; t < NTAP; ++t) acc[(u - t + NTAP) % NTAP] += x[i] * h[t];
                // output r - NTAP + 1 is complete (its last tap was this row): register (u + 1) mod NTAP
                const long long k = r - (NTAP - 1);
                if (k >= 0 && k < NI && k < outs) dst[k * row16] = acc[(u + 1) % NTAP];
                acc[(u + 1) % NTAP] = f4v{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// Polyphase filter bank, register sliding window.  A 256-thread workgroup
// computes NG = 4096 / N consecutive spectra of one stream pair: thread t owns
// columns t + 256 c (c < P = N / 256) of every spectrum and, one column at a
// time, loads that column's NTAP taps and its NTAP + NG - 1 input rows (each
// row feeds up to NG accumulators) instead of re-reading NTAP rows per
// spectrum.  The accumulators are then exactly the register layout of the
// fused channelizer (NG groups of P points), so the FFT is radix-P + the tail
// of the 4096-point transform.  BBT_PFB_BATCH caps the rows loaded together
// (default: all of them; 8 and 5 measured 5 % slower for 12 x 1024).
#ifndef BBT_PFB_BATCH
#define BBT_PFB_BATCH 64
#endif
// SINGLE (S == 1): the two transforms side by side are the workgroup's spectra
// [i0, i0 + NG) and [i0 + NG, i0 + 2 NG) of the one stream (rows NG further on).
// SPLIT (with SINGLE): the one stream is z = a + i b of two real streams and the
// half spectra of a and b are written, as in k_fft_rows.
// MANY streams (PP > 1): PP neighbouring stream pairs per workgroup, lanes over the pairs first
// (thread = pp + PP tau), so that a wave's load of one row takes PP * 16 contiguous bytes of
// each of 64 / PP complete samples instead of 16 bytes out of 64 lines; each pair is the
// one-pair kernel on T = GN / 16 threads of a GN-point geometry: NG = GN / N spectra per
// workgroup, its own exchange area.
template <int N, int NTAP, bool SINGLE = false, bool SPLIT = false, int PP = 1, int GN = 4096>
__global__ __launch_bounds__(GN / 16 * PP, (GN / 16 * PP == 512 ? 4 : 1))      // (512 threads: two workgroups per CU)
void k_pfb_window(const float2* __restrict__ in, float2* __restrict__ out, long long n_spec,
                  int S, const float* __restrict__ taps, const cf* __restrict__ tw0,
                  const cf* __restrict__ tw1, int GL = 0) {
    static_assert((PP == 1 && GN == 4096) || (!SINGLE && !SPLIT), "several pairs: plain stream pairs only");
    typedef FftGeo<GN> G;
    constexpr int T = GN / 16;
    constexpr int NG = GN / N;         // spectra per workgroup
    constexpr int P = 16 / NG;         // columns per thread
    static_assert(P >= 1 && P <= 8, "GN = N NG with 2 <= NG <= 16");
    // (several pairs: their exchange areas interleaved element by element, the pair fastest like the
    // lanes -- COLMODE = PP of fft_core.hpp: side by side in areas of their own 48 % (PP = 4) and 74 % (8)
    // of the LDS cycles were bank conflicts)
    constexpr int CM = PP > 1 ? PP : 0;
    __shared__ v2 lds[PP * G::LDS_ELEMS];
    const int tau = threadIdx.x / PP, pp = threadIdx.x % PP;
    const int npair = SINGLE ? 1 : S >> 1;
    const int ngrp = npair / PP;                                      // (the host checks npair % PP == 0)
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    // Order of the workgroups: the GL pair groups that share 128-byte lines first, then the
    // spectra, then the further pair groups -- an XCD then walks along the spectra of one slab of
    // columns and finds the NTAP - 1 rows it shares with the previous workgroup in its L2 (GL = 0:
    // all pair groups first, as the one-pair kernel always did).
    const unsigned gl = (PP > 1 && GL > 0) ? (unsigned)GL : (unsigned)ngrp;
    const unsigned n_sg = gridDim.x / ngrp;
    const unsigned lo = vb % gl, rest = vb / gl;
    const unsigned grp = (rest / n_sg) * gl + lo;
    const long long i0 = (long long)(rest % n_sg) * (SINGLE ? 2 * NG : NG);   // first spectrum of this workgroup
    const int sp = grp * PP + pp;
    c2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = czero();
    const long long rows_left = n_spec + NTAP - 1 - i0;     // input rows that exist from i0 on
    const float2* src = SINGLE ? in + (i0 * N + tau) : in + ((i0 * N + tau) * S + 2 * sp);
    // One column of the thread at a time: all NTAP + NG - 1 row loads of the
    // column are issued together (the kernel is bound by load latency at 2
    // waves/SIMD: this keeps 15 loads per thread in flight instead of ~4), then
    // each row feeds the accumulators of the spectra it belongs to.
    // taps: the window variants read them as float4 over four consecutive taps of a column,
    // permuted by the host to [column group c][tap quad][thread] (bbt_pfb_plan_create): NTAP / 4
    // fully coalesced 16-byte loads per column instead of NTAP 4-byte ones -- the kernel is short
    // of vector-memory issue slots, not of bytes.
    static_assert(NTAP % 4 == 0, "window kernels: taps in quads");
    const float4* taps4 = reinterpret_cast<const float4*>(taps);
#pragma unroll
    for (int c = 0; c < P; ++c) {
        float h[NTAP];
#pragma unroll
        for (int q = 0; q < NTAP / 4; ++q) {
            const float4 hq = taps4[(c * (NTAP / 4) + q) * T + tau];
            h[4 * q] = hq.x;
            h[4 * q + 1] = hq.y;
            h[4 * q + 2] = hq.z;
            h[4 * q + 3] = hq.w;
        }
        // rows past the end of the stream (last workgroup only): read the last
        // existing row instead and zero it, so the loads stay branch free
        constexpr int NR = NTAP + NG - 1;
        // (512 threads and more: 128 registers, so the rows in two or three batches)
        constexpr int RBMAX = (GN / 16 * PP >= 512) ? ((NR + 1) / 2 < 8 ? (NR + 1) / 2 : 8) : BBT_PFB_BATCH;
        constexpr int RB = (RBMAX < NR) ? RBMAX : NR;                      // rows in flight per batch
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += RB) {
            c2 x[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = r0 + i;
                if (r < NR) {
                    const long long rr = r < rows_left ? r : rows_left - 1;
                    if constexpr (SINGLE) {
                        const long long rb = r + NG < rows_left ? r + NG : rows_left - 1;
                        const float2 a = src[rr * N + T * c], b = src[rb * N + T * c];
                        x[i] = c2{v2{a.x, b.x}, v2{a.y, b.y}};
                    } else {
                        x[i] = ld_ext(src + (rr * N + T * c) * S);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = r0 + i;
                if (r < NR) {
                    // rows past the end of the stream count as zeros (once per row, not per tap)
                    const float keep = r < rows_left ? 1.f : 0.f;
                    const float keep_b = (!SINGLE || r + NG < rows_left) ? 1.f : 0.f;
                    const v2 kk = SINGLE ? v2{keep, keep_b} : v2{keep, keep};
                    const c2 xr = c2{x[i].re * kk, x[i].im * kk};
#pragma unroll
                    for (int q = 0; q < NG; ++q) {
                        const int t = r - q;                  // tap index for spectrum i0 + q
                        if (t >= 0 && t < NTAP) {
                            v[q * P + c].re += xr.re * h[t];
                            v[q * P + c].im += xr.im * h[t];
                        }
                    }
                }
            }
            asm volatile("" ::: "memory");
        }
        // keep the next column's loads behind this column's arithmetic: hoisting
        // them all (the loop is unrolled) needs 4 x 60 registers
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // FFT over the columns of each spectrum: radix-P, twiddle W_N^{tau c}, tail
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        if constexpr (P > 1) {
            c2 t[P];
#pragma unroll
            for (int c = 0; c < P; ++c) t[c] = v[q * P + c];
            radixR<-1, P>(t);
#pragma unroll
            for (int c = 0; c < P; ++c) v[q * P + c] = t[c];
        }
    }
#pragma unroll
    for (int c = 1; c < P; ++c) {
        const cf w = tw0[(c * NG) * T + tau];
#pragma unroll
        for (int q = 0; q < NG; ++q) v[q * P + c] = twmul<-1>(v[q * P + c], w);
    }
    wg_fft_tail<GN, -1, CM, 0, BBT_PFB_TW_POW>(v, lds, tau, pp, tw1);
    const int f = tau & 15, g = tau >> 4;
    const int q = f / P, c = f - q * P;
    if constexpr (SINGLE && SPLIT) {
        // the NG spectra of each half of the pair through the exchange area (real parts, then
        // imaginary parts) at [q N + k]; then channel k meets N - k of the same spectrum
        constexpr int HALF = N / 2 + 1, PER = N / 2;          // outputs per spectrum handled in the main loop
        v2 zk_re[8], zm_re[8], zk_im[8], zm_im[8], ny_re, ny_im;
        __syncthreads();
#pragma unroll
        for (int c2i = 0; c2i < 16; ++c2i) lds[q * N + c + P * (g + 16 * c2i)] = v[c2i].re;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tau + 256 * j, qq = idx / PER, k = idx - qq * PER;
            zk_re[j] = lds[qq * N + k];
            zm_re[j] = lds[qq * N + ((N - k) & (N - 1))];
        }
        ny_re = lds[(tau & (NG - 1)) * N + N / 2];
        __syncthreads();
#pragma unroll
        for (int c2i = 0; c2i < 16; ++c2i) lds[q * N + c + P * (g + 16 * c2i)] = v[c2i].im;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tau + 256 * j, qq = idx / PER, k = idx - qq * PER;
            zk_im[j] = lds[qq * N + k];
            zm_im[j] = lds[qq * N + ((N - k) & (N - 1))];
        }
        ny_im = lds[(tau & (NG - 1)) * N + N / 2];
        float4* spectra = reinterpret_cast<float4*>(out);      // (spectrum, k) -> (a, b): 16 bytes
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j == 8 && tau >= NG) break;
            const int idx = tau + 256 * j;
            const int qq = j < 8 ? idx / PER : tau, k = j < 8 ? idx - qq * PER : N / 2;
            const v2 kr = j < 8 ? zk_re[j] : ny_re, mr = j < 8 ? zm_re[j] : ny_re;
            const v2 ki = j < 8 ? zk_im[j] : ny_im, mi = j < 8 ? zm_im[j] : ny_im;
            const v2 ar = 0.5f * (kr + mr), ai = 0.5f * (ki - mi), br = 0.5f * (ki + mi), bi = -0.5f * (kr - mr);
            const long long sa = i0 + qq, sb = i0 + NG + qq;
            if (sa < n_spec) spectra[sa * HALF + k] = make_float4(ar.x, ai.x, br.x, bi.x);
            if (sb < n_spec) spectra[sb * HALF + k] = make_float4(ar.y, ai.y, br.y, bi.y);
        }
        return;
    }
    if constexpr (!SINGLE && SPLIT) {
        // stream pairs, each stream z = a + i b of two real streams: the same pairing of k with
        // N - k; the two halves of the registers are the pair's two streams, same spectra
        constexpr int HALF = N / 2 + 1, PER = N / 2;
        v2 zk_re[8], zm_re[8], zk_im[8], zm_im[8], ny_re, ny_im;
        __syncthreads();
#pragma unroll
        for (int c2i = 0; c2i < 16; ++c2i) lds[q * N + c + P * (g + 16 * c2i)] = v[c2i].re;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tau + 256 * j, qq = idx / PER, k = idx - qq * PER;
            zk_re[j] = lds[qq * N + k];
            zm_re[j] = lds[qq * N + ((N - k) & (N - 1))];
        }
        ny_re = lds[(tau & (NG - 1)) * N + N / 2];
        __syncthreads();
#pragma unroll
        for (int c2i = 0; c2i < 16; ++c2i) lds[q * N + c + P * (g + 16 * c2i)] = v[c2i].im;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tau + 256 * j, qq = idx / PER, k = idx - qq * PER;
            zk_im[j] = lds[qq * N + k];
            zm_im[j] = lds[qq * N + ((N - k) & (N - 1))];
        }
        ny_im = lds[(tau & (NG - 1)) * N + N / 2];
        float4* spectra = reinterpret_cast<float4*>(out) + 2 * sp;       // (spectrum, k, S complex streams)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j == 8 && tau >= NG) break;
            const int idx = tau + 256 * j;
            const int qq = j < 8 ? idx / PER : tau, k = j < 8 ? idx - qq * PER : N / 2;
            const v2 kr = j < 8 ? zk_re[j] : ny_re, mr = j < 8 ? zm_re[j] : ny_re;
            const v2 ki = j < 8 ? zk_im[j] : ny_im, mi = j < 8 ? zm_im[j] : ny_im;
            const v2 ar = 0.5f * (kr + mr), ai = 0.5f * (ki - mi), br = 0.5f * (ki + mi), bi = -0.5f * (kr - mr);
            const long long sa = i0 + qq;
            if (sa < n_spec) {
                float4* dst = spectra + (sa * HALF + k) * S;
                dst[0] = make_float4(ar.x, ai.x, br.x, bi.x);
                dst[1] = make_float4(ar.y, ai.y, br.y, bi.y);
            }
        }
        return;
    }
    if constexpr (SINGLE) {
        float2* dst = out + ((i0 + q) * N + c);
        const bool act_a = i0 + q < n_spec, act_b = i0 + NG + q < n_spec;
#pragma unroll
        for (int c2i = 0; c2i < 16; ++c2i) {
            const long long k = P * (g + 16 * c2i);
            if (act_a) dst[k] = make_float2(v[c2i].re.x, v[c2i].im.x);
            if (act_b) dst[(long long)NG * N + k] = make_float2(v[c2i].re.y, v[c2i].im.y);
        }
        return;
    }
    if (i0 + q < n_spec) {
        // register u + U c2 holds output k' = g + R2 u + 16 c2 of the thread's sequence (wg_fft_tail)
        constexpr int R2 = G::R2, U = 16 / R2;
        float2* dst = out + (((i0 + q) * N + c) * S + 2 * sp);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            st_ext(dst + (long long)(P * (g + R2 * (reg % U) + 16 * (reg / U))) * S, v[reg], S == 2);
    }
}

// ---------------------------------------------------------------------------
// Detection + integration (reference functions.py:15-16, 131-143 and
// integration.py:252-303 for an integer step): out[i] = scale * sum over the
// `step` input samples [i*step, (i+1)*step) of
//   MODE 0  |z|^2 per complex element            (Square)   float2 -> float
//   MODE 1  |X|^2, |Y|^2, Re X Y*, Im X Y*        (Power)    float4 (X, Y) -> float4
//   MODE 2  the element itself                    (Integrate of a real stream) float -> float
// `q` = elements per complete sample in units of the mode's input type.  The
// sum runs sequentially in time, in float32, like np.add.reduceat along axis 0.
template <int MODE>
__global__ __launch_bounds__(256) void k_detect_integrate(const void* __restrict__ in_,
                                                          void* __restrict__ out_, long long n_out,
                                                          long long step, long long q, float scale) {
    const long long tiles = (q + 255) / 256;
    const long long tile = (long long)blockIdx.x;
    const long long i = tile / tiles;
    const long long e = (tile - i * tiles) * 256 + threadIdx.x;
    if (i >= n_out || e >= q) return;
    const long long first = i * step * q + e;
    if (MODE == 0) {
        const float2* in = (const float2*)in_ + first;
        float acc = 0.f;
        for (long long s = 0; s < step; ++s) {
            const float2 z = in[s * q];
            acc += z.x * z.x + z.y * z.y;
        }
        ((float*)out_)[i * q + e] = acc * scale;
    } else if (MODE == 1) {
        const float4* in = (const float4*)in_ + first;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (long long s = 0; s < step; ++s) {
            const float4 z = in[s * q];              // X = (x, y), Y = (z, w)
            acc.x += z.x * z.x + z.y * z.y;
            acc.y += z.z * z.z + z.w * z.w;
            acc.z += z.x * z.z + z.y * z.w;
            acc.w += z.y * z.z - z.x * z.w;
        }
        ((float4*)out_)[i * q + e] =
            make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
    } else {
        const float* in = (const float*)in_ + first;
        float acc = 0.f;
        for (long long s = 0; s < step; ++s) acc += in[s * q];
        ((float*)out_)[i * q + e] = acc * scale;
    }
}

// ---------------------------------------------------------------------------
// Per-stream integer sample shifts (reference sampling.py:380-425,
// ShiftSamples.task: data[self._indices]): out[i, e] = in[i + offset[e], e]
// for the E elements of a complete sample, offset[e] = shift.max() - shift[e].
// Element size 4, 8 or 16 bytes (the host merges neighbouring elements with the
// same offset -- the polarisations of a sub-band -- into one wider element).
// A workgroup covers 256 / LE rows x LE (a power of two) neighbouring elements
// per step, ITER steps: no division, 32-bit index arithmetic inside a row.
template <typename T, int ITER>
__global__ __launch_bounds__(256) void k_shift_samples(const T* __restrict__ in, T* __restrict__ out,
                                                       long long n_rows, int n_elem, int lg_le,
                                                       const int* __restrict__ offset) {
    const int le = 1 << lg_le;
    const int e = blockIdx.y * le + (threadIdx.x & (le - 1));
    if (e >= n_elem) return;
    const int rows_per_step = 256 >> lg_le;
    const long long off = offset[e];
    // consecutive row chunks on one XCD: a line of the input is wanted by its elements at
    // different rows (within the span of the offsets), which should find it in that L2
    long long r = (long long)xcd_remap(blockIdx.x, gridDim.x) * (rows_per_step * ITER) + (threadIdx.x >> lg_le);
#pragma unroll
    for (int it = 0; it < ITER; ++it, r += rows_per_step)
        if (r < n_rows) out[r * n_elem + e] = in[(r + off) * n_elem + e];
}

// The same, tiled: a workgroup takes one GROUP of neighbouring elements that fill a cache line
// (G = 128 / sizeof(T)) over `rows` output rows.  Its input lines -- rows [r0 + lo, r0 + rows + hi),
// lo / hi the smallest / largest offset in the group -- are staged in LDS, each read once and
// whole (the gather above has every lane ask for its own 16 bytes of a different line, and
// every line asked for G times), then the output rows are assembled from there.  Worth it while
// the group's offsets span less than half the window (neighbouring sub-bands of a dispersed
// band do); a group that spans more takes the gather path in the same launch.
//   group_lo[g], group_span[g] : smallest offset and hi - lo of group g;  window: LDS rows
template <typename T>
__global__ __launch_bounds__(256) void k_shift_tiled(const T* __restrict__ in, T* __restrict__ out,
                                                     long long n_rows, int n_elem,
                                                     const int* __restrict__ offset,
                                                     const int* __restrict__ group_lo,
                                                     const int* __restrict__ group_span, int window) {
    constexpr int G = 128 / (int)sizeof(T);
    constexpr int STEP = 256 / G;                      // rows a workgroup touches per instruction
    extern __shared__ unsigned char shift_lds_raw[];
    T* lds = reinterpret_cast<T*>(shift_lds_raw);
    const int g = blockIdx.y, elem = threadIdx.x % G, row0 = threadIdx.x / G;
    const int e = g * G + elem;
    const int lo = group_lo[g], span = group_span[g];
    // output rows per workgroup: what the window leaves beside the span of the group's offsets
    // (the grid is sized for the smallest case, window / 2; surplus workgroups leave at once)
    const bool tiled = span <= window / 2;
    const int rows = tiled ? window - span : window / 2;
    const long long r0 = (long long)xcd_remap(blockIdx.x, gridDim.x) * rows;
    if (r0 >= n_rows) return;
    const int nr = (int)(n_rows - r0 < rows ? n_rows - r0 : rows);
    const long long off = offset[e];
    if (!tiled) {                                      // offsets too far apart: gather
        for (int r = row0; r < nr; r += STEP) out[(r0 + r) * n_elem + e] = in[(r0 + r + off) * n_elem + e];
        return;
    }
    const int wrows = nr + span;
    const T* src = in + ((r0 + lo) * n_elem + e);
#pragma unroll 8
    for (int w = row0; w < wrows; w += STEP) lds[w * G + elem] = src[(long long)w * n_elem];
    __syncthreads();
    const int d = (int)off - lo;
    T* dst = out + (r0 * n_elem + e);
#pragma unroll 8
    for (int r = row0; r < nr; r += STEP) dst[(long long)r * n_elem] = lds[(r + d) * G + elem];
}

// Power with the polarization axis anywhere in the sample (reference
// functions.py:131-143 takes any axis): a complete sample is (outer, 2, inner)
// complex, X = [:, 0, :], Y = [:, 1, :]; out (n_out, outer, 4, inner) float32 =
// scale * sum over `step` samples of |X|^2, |Y|^2, Re X conj(Y), Im X conj(Y).
// One thread per (output sample, outer, inner) position.
__global__ __launch_bounds__(256) void k_power_axis(const float2* __restrict__ in, float* __restrict__ out,
                                                    long long n_out, long long step, int outer, int inner,
                                                    float scale) {
    const long long per = (long long)outer * inner;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_out * per) return;
    const long long i = t / per;
    const int r = (int)(t - i * per);
    const int o = r / inner, k = r - o * inner;
    const long long row = 2 * per;                       // complex elements per input sample
    const float2* src = in + i * step * row + ((long long)o * 2) * inner + k;
    float xx = 0.f, yy = 0.f, re = 0.f, im = 0.f;
    for (long long s = 0; s < step; ++s) {
        const float2 x = src[s * row], y = src[s * row + inner];
        xx += x.x * x.x + x.y * x.y;
        yy += y.x * y.x + y.y * y.y;
        re += x.x * y.x + x.y * y.y;
        im += x.y * y.x - x.x * y.y;
    }
    float* dst = out + (i * outer + o) * 4ll * inner + k;
    dst[0] = xx * scale;
    dst[inner] = yy * scale;
    dst[2 * inner] = re * scale;
    dst[3 * inner] = im * scale;
}

// Pitched device-to-device copy: `rows` runs of `wpr` elements of type T (4, 8
// or 16 bytes), run r from src + r * spitch to dst + r * dpitch (pitches in
// elements).  hipMemcpy2DAsync moves 8-byte-wide rows -- padding an odd stream
// count to even -- at 0.16 TB/s; this is a plain coalesced copy.  Lanes run along
// the flattened (row, element) index; a workgroup covers 256 / LE rows x LE
// elements (LE a power of two >= wpr, or 256 with the row tiled in blockIdx.y).
// Columns from `src_wpr` on (bbt_pad_streams) are written as zeros.
template <typename T>
__global__ __launch_bounds__(256) void k_copy2d(const T* __restrict__ src, T* __restrict__ dst,
                                                long long rows, int wpr, int src_wpr,
                                                long long spitch, long long dpitch, int lg_le) {
    const int le = 1 << lg_le;
    const int c = blockIdx.y * le + (threadIdx.x & (le - 1));
    if (c >= wpr) return;
    const int rows_per_step = 256 >> lg_le;
    long long r = (long long)blockIdx.x * (rows_per_step * 4) + (threadIdx.x >> lg_le);
    T zero;
    __builtin_memset(&zero, 0, sizeof(T));
#pragma unroll
    for (int it = 0; it < 4; ++it, r += rows_per_step)
        if (r < rows) dst[r * dpitch + c] = c < src_wpr ? src[r * spitch + c] : zero;
}

// ---------------------------------------------------------------------------
// Glue for real-valued (float32) streams, which run through the complex
// kernels (reference: the rfft/irfft engine paths, fourier/numpy.py:41-49).
//   MODE 0  real -> complex (zero imaginary part)            out[t] = (in[t], 0)
//   MODE 1  complex -> real part                             out[t] = in[t].x
//   MODE 2  half spectrum (n/2+1 channels) -> Hermitian full spectrum of n
//           channels, as irfft interprets it (imaginary parts of the DC and
//           Nyquist channels ignored); rows = spectra, `s` streams innermost
//   MODE 3  x -> x^2 (Square of a real stream)
template <int MODE, bool PADDED = false>
__global__ __launch_bounds__(256) void k_real_ops(const void* __restrict__ in_,
                                                  void* __restrict__ out_, long long n_total, int n,
                                                  int s) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_total) return;
    if (MODE == 0) {
        ((float2*)out_)[t] = make_float2(((const float*)in_)[t], 0.f);
    } else if (MODE == 1) {
        ((float*)out_)[t] = ((const float2*)in_)[t].x;
    } else if (MODE == 3) {
        const float x = ((const float*)in_)[t];
        ((float*)out_)[t] = x * x;
    } else if (MODE == 4) {
        // spectra of two real streams from the transform of z = a + i b:
        // A[k] = (Z[k] + conj Z[n-k]) / 2,  B[k] = (Z[k] - conj Z[n-k]) / 2i.
        // in (spectrum, n, s/2) complex, out (spectrum, n/2+1, s); t indexes out
        // (PADDED: the input carries one more, unused, pair stream -- an odd number
        // of pairs went through the transform padded to even)
        const int half = n / 2 + 1, np2 = (s >> 1) + (PADDED ? 1 : 0);
        const int st = (int)(t % s);
        const long long r = t / s;
        const int k = (int)(r % half);
        const long long spec = r / half;
        const float2* row = (const float2*)in_ + spec * n * np2 + (st >> 1);
        const float2 zk = row[(long long)k * np2];
        const float2 zm = row[(long long)((n - k) % n) * np2];
        ((float2*)out_)[t] = (st & 1) ? make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x))
                                      : make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    } else if (MODE == 5) {
        // the inverse: Z[k] = A[k] + i B[k] from the half spectra of two real
        // streams (irfft conventions: imaginary parts of DC and Nyquist ignored).
        // in (spectrum, n/2+1, s) complex, out (spectrum, n, s/2); t indexes out
        const int half = n / 2 + 1, np2 = s >> 1;
        const int p = (int)(t % np2);
        const long long r = t / np2;
        const int k = (int)(r % n);
        const long long spec = r / n;
        const int kk = k <= n / 2 ? k : n - k;
        const float2* src = (const float2*)in_ + (spec * half + kk) * s + 2 * p;
        float2 a = src[0], b = src[1];
        if (kk == 0 || 2 * kk == n) a.y = b.y = 0.f;
        ((float2*)out_)[t] = k <= n / 2 ? make_float2(a.x - b.y, a.y + b.x)
                                        : make_float2(a.x + b.y, b.x - a.y);
    } else {
        // t indexes out (spectrum, k, stream)
        const int st = (int)(t % s);
        const long long r = t / s;
        const int k = (int)(r % n);
        const long long spec = r / n;
        const int half = n / 2 + 1;
        const float2* row = (const float2*)in_ + spec * half * s;
        float2 z;
        if (k <= n / 2) {
            z = row[(long long)k * s + st];
            if (k == 0 || 2 * k == n) z.y = 0.f;
        } else {
            z = row[(long long)(n - k) * s + st];
            z.y = -z.y;
        }
        ((float2*)out_)[t] = z;
    }
}

// The dispersion chirp, Disperse.phase_factor (reference dispersion.py:115-129, dm.py:78-105), in
// float64 on the GPU, cast to complex64 as the reference casts it:
//   f = freq_c + s_c fftfreq(n, 1/rate)[k]   (Hz)
//   phase = s_c d f_MHz (1/fref_MHz - 1/f_MHz)^2 1e6 + offset fftfreq[k]   (cycles; d = D DM)
//   out[c][k] = exp(2 pi i phase)
// col[c] = (freq_hz, sideband, ref_hz, unused).  The phase runs to 1e6 cycles and more (config 4):
// its fractional part is taken in float64 before the sine and cosine.
__global__ __launch_bounds__(256) void k_chirp(float2* __restrict__ out, long long n, const double4* __restrict__ col,
                                               double rate_hz, double d, double offset) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int c = blockIdx.y;
    const double4 p = col[c];
    const double fft_freq = (double)(k < (n + 1) / 2 ? k : k - n) * rate_hz / (double)n;      // numpy.fft.fftfreq
    const double f = (p.x + fft_freq * p.y) / 1e6, fr = p.z / 1e6;
    const double t = 1. / fr - 1. / f;
    double phase = d * f * (t * t) * 1e6 * p.y;
    phase += offset * fft_freq;
    const double frac = phase - floor(phase);
    double sn, cs;
    sincospi(2. * frac, &sn, &cs);
    out[(long long)c * n + k] = make_float2((float)cs, (float)sn);
}

// Per-stream complex factor (reference sampling.py:374-377, TimeDelay.task:
// data *= phase_factor): out[i, e] = in[i, e] * factor[e].
__global__ __launch_bounds__(256) void k_scale_streams(const float2* __restrict__ in,
                                                       float2* __restrict__ out, long long n_total,
                                                       int n_elem, const float2* __restrict__ factor) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_total) return;
    out[t] = cmul(in[t], factor[t % n_elem]);
}

// ---------------------------------------------------------------------------
// Time-domain FIR for short responses: Convolve.task (reference
// convolution.py:116-120) keeps ifft(fft(x) * fft(response))[n_tap - 1:], i.e.
//   out[i, s] = sum_{m < n_tap} in[i + m, s] * g[s][m],   g[s][m] = response[n_tap - 1 - m, s],
// the exact linear convolution, which is block independent.  For a windowed
// sinc of 129 taps (Resample / ShiftAndResample) doing it directly costs one
// read and one write of the stream instead of three FFT passes.
//
// A thread computes R consecutive outputs of one stream pair from a sliding
// window: per chunk of R inputs (read once from LDS) it does R*R multiply-adds
// with the 2R-1 taps the chunk touches (uniform: scalar loads).  The input
// tile sits in LDS as R rows of PITCH 16-byte elements, sample s at row s % R,
// column s / R, so both the coalesced fill (16 consecutive samples -> 16
// different 16-byte bank groups because PITCH = 2 mod 16) and the per-thread
// reads (column t + q for thread t) are conflict free.
//   tre/tim : [npair][tap_pitch] float2 (g_A, g_B) real / imaginary parts,
//             R-1 zeros in front and >= R zeros behind the n_tap taps.
// `half` > 0 (one stream, S == 1): the two filters a thread carries side by side
// are the first and the second half of the one stream -- outputs [0, half) and
// [half, n_out) -- instead of the two streams of a pair.
template <int R, bool CPLX>
__device__ __forceinline__ void fir_tile(const float2* __restrict__ in, float2* __restrict__ out,
                                         long long n_in, long long n_out, int S, int sp,
                                         long long base, const float2* __restrict__ tre,
                                         const float2* __restrict__ tim, int tap_pitch,
                                         int n_chunks, int pitch, float4* __restrict__ fir_tile_lds,
                                         long long half = 0) {
    constexpr int T = 256;
    const int t = threadIdx.x;
    const int n_tile = R * (T + n_chunks);
    for (int s = t; s < n_tile; s += T) {
        const long long g = base + s;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (half > 0) {
            if (g < n_in) {
                const float2 a = in[g];
                x.x = a.x;
                x.y = a.y;
            }
            if (g + half < n_in) {
                const float2 b = in[g + half];
                x.z = b.x;
                x.w = b.y;
            }
        } else if (g < n_in) x = *reinterpret_cast<const float4*>(in + (g * S + 2 * sp));
        // the tile holds register order (re_A re_B im_A im_B): the swizzle is done once per
        // sample here instead of once per use in the tap loop (3 v_mov per input, 15 % of its VALU)
        fir_tile_lds[(s % R) * pitch + s / R] = make_float4(x.x, x.z, x.y, x.w);
    }
    __syncthreads();
    const float2* gre = tre + (long long)sp * tap_pitch;
    const float2* gim = tim + (long long)sp * tap_pitch;
    c2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = czero();
    for (int q = 0; q < n_chunks; ++q) {
        c2 x[R];
#pragma unroll
        for (int p = 0; p < R; ++p) {
            const float4 v = fir_tile_lds[p * pitch + t + q];
            x[p] = c2{v2{v.x, v.y}, v2{v.z, v.w}};
        }
        v2 wr[2 * R - 1], wi[2 * R - 1];
#pragma unroll
        for (int k = 0; k < 2 * R - 1; ++k) {
            const float2 a = gre[R * q + k];
            wr[k] = v2{a.x, a.y};
            if (CPLX) {
                const float2 b = gim[R * q + k];
                wi[k] = v2{b.x, b.y};
            }
        }
#pragma unroll
        for (int p = 0; p < R; ++p)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int k = p - r + R - 1;            // tap g[R q + p - r]
                acc[r].re += x[p].re * wr[k];
                acc[r].im += x[p].im * wr[k];
                if (CPLX) {
                    acc[r].re -= x[p].im * wi[k];
                    acc[r].im += x[p].re * wi[k];
                }
            }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r)
        fir_tile_lds[r * pitch + t] = make_float4(acc[r].re.x, acc[r].im.x, acc[r].re.y, acc[r].im.y);
    __syncthreads();
    for (int s = t; s < T * R; s += T) {
        const long long g = base + s;
        if (half > 0) {
            const float4 y = fir_tile_lds[(s % R) * pitch + s / R];
            if (g < half && g < n_out) out[g] = make_float2(y.x, y.y);
            if (g + half < n_out) out[g + half] = make_float2(y.z, y.w);
        } else if (g < n_out) {
            *reinterpret_cast<float4*>(out + (g * S + 2 * sp)) = fir_tile_lds[(s % R) * pitch + s / R];
        }
    }
}

template <int R, bool CPLX>
__global__ __launch_bounds__(256) void k_fir(const float2* __restrict__ in, float2* __restrict__ out,
                                             long long n_in, long long n_out, int S,
                                             const float2* __restrict__ tre,
                                             const float2* __restrict__ tim, int tap_pitch,
                                             int n_chunks, int pitch) {
    extern __shared__ float4 fir_tile_mem[];
    // the pairs of one tile share cache lines: consecutive virtual ids, one XCD
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    if (S == 1) {               // one stream: the grid covers the first half, see fir_tile
        const long long tiles = gridDim.x;
        fir_tile<R, CPLX>(in, out, n_in, n_out, 1, 0, (long long)vb * (256 * R), tre, tim, tap_pitch, n_chunks,
                          pitch, fir_tile_mem, tiles * (256 * R));
        return;
    }
    const int npair = S >> 1;
    fir_tile<R, CPLX>(in, out, n_in, n_out, S, vb % npair, (long long)(vb / npair) * (256 * R), tre, tim,
                      tap_pitch, n_chunks, pitch, fir_tile_mem);
}

// ---------------------------------------------------------------------------
// Unpacking of sampler frames (the decode step of the
// `baseband` readers in front of the path -- VDIF / DADA payloads).  The
// reference itself holds no decoder (it takes `baseband` stream readers, a
// package that is not in this image) and no sample files: this follows the
// published formats (VDIF 1.1.1: little-endian 32-bit words, the first sample
// in the least significant bits, channels of a complete sample adjacent, I
// before Q; DADA: signed 8-bit).  PARITY UNPINNED -- see ingest.py.
//   raw   : n_frames frames of frame_bytes, payload after header_bytes
//   out   : float32 [(set * spf + t) * n_thread + thread][e], e < E; frame f
//           belongs to set f / n_thread, thread f % n_thread
//   code 0: VDIF levels as `baseband` decodes them (its base/encoding.py, restated from the
//           published source -- the package is not in this image): 1 bit {-1, +1}; 2 bits
//           {-3.3359, -1, +1, +3.3359} (OPTIMAL_2BIT_HIGH); 4 bits (v - 8) / 2.95
//           (FOUR_BIT_1_SIGMA); 8 bits (v - 127.5) / 35.5 (EIGHT_BIT_1_SIGMA: 0..255 encode
//           -127.5..127.5, scaled to look like 2-bit data); 16 bits offset binary v - 2^15
//           (no `baseband` decoder to follow)
//   code 1: two's complement integers (8 or 16 bits)
// One frame per blockIdx.x, 256 * G consecutive components of it per
// blockIdx.y: a thread decodes G (1, 2 or 4, dividing E) components that are
// adjacent in the payload (G * bits <= 64 of its bits) and in the output, and
// stores them at once (16 bytes for G == 4).  Index arithmetic is 32-bit
// inside a frame; no 64-bit division anywhere.
__device__ __forceinline__ float unpack_level(unsigned v, int bits, int code) {
    if (code == 1) return bits == 8 ? (float)(signed char)v : (float)(short)v;
    if (bits == 1) return v ? 1.f : -1.f;
    if (bits == 2) return v == 0 ? -3.3359f : (v == 1 ? -1.f : (v == 2 ? 1.f : 3.3359f));
    if (bits == 4) return ((float)v - 8.f) / 2.95f;
    if (bits == 8) return ((float)v - 127.5f) / 35.5f;
    return (float)v - (float)(1u << (bits - 1));
}
#define BBT_UNPACK_ITER 4     // groups of G components per thread
// `valid` (optional): one byte per frame; frames flagged 0 (invalid or missing in
// the file: their bytes are whatever the host put there) are written as zeros,
// the fill value of `baseband`'s readers.
// BITS: the width as a compile-time constant (0: run-time `bits`), so that the level of a code
// is a few selects or one fused multiply-add instead of a ladder of run-time tests; lg_e >= 0:
// E is that power of two (the usual case) and the component -> (sample, element) split is a
// shift instead of a 32-bit division per group.  (2-bit VDIF x 8 channels: 0.41 -> see DESIGN.)
template <int G, int BITS = 0>
__global__ __launch_bounds__(256) void k_unpack(const unsigned char* __restrict__ raw,
                                                float* __restrict__ out, int frame_bytes,
                                                int header_bytes, int bits_rt, int spf, int n_thread,
                                                int E, int code,
                                                const unsigned char* __restrict__ valid, int lg_e) {
    const int bits = BITS ? BITS : bits_rt;
    const unsigned frame = blockIdx.x;
    const bool good = valid == nullptr || valid[frame] != 0;
    const unsigned set = frame / (unsigned)n_thread, thr = frame - set * (unsigned)n_thread;
    const unsigned* payload =
        reinterpret_cast<const unsigned*>(raw + (long long)frame * frame_bytes + header_bytes);
    const unsigned n_comp = (unsigned)spf * (unsigned)E;
    const unsigned mask = (1u << bits) - 1u;                        // bits <= 16
    float* frame_out = out + ((long long)set * spf * n_thread + thr) * E;
#pragma unroll
    for (int it = 0; it < BBT_UNPACK_ITER; ++it) {
        const unsigned q = ((blockIdx.y * BBT_UNPACK_ITER + it) * 256u + threadIdx.x) * G;   // component in the frame
        if (q >= n_comp) return;
        const unsigned t = lg_e >= 0 ? q >> lg_e : q / (unsigned)E, e = q - t * (unsigned)E;
        const unsigned bit = q * (unsigned)bits;                    // G * bits divides 32, or is 64
        unsigned long long w = payload[bit >> 5];
        if (G * bits > 32) w |= (unsigned long long)payload[(bit >> 5) + 1] << 32;
        w >>= (bit & 31);
        float x[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
            x[g] = good ? unpack_level((unsigned)(w >> (g * bits)) & mask, bits, code) : 0.f;
        float* dst = frame_out + (long long)t * n_thread * E + e;
        if (G == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[G > 1 ? 1 : 0], x[G > 2 ? 2 : 0], x[G - 1]);
        } else if (G == 2) {
            *reinterpret_cast<float2*>(dst) = make_float2(x[0], x[G - 1]);
        } else {
            dst[0] = x[0];
        }
    }
}

}  // namespace bbt
