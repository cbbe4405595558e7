// The kernels of gen_kernels.hpp on the register-resident mixed-radix engine (fft_gen2.hpp),
// templated on the compile-time geometry: same data contract, same sources / multipliers /
// sinks, same three-pass structure.
//
//   k_g2_osm_small   N <= 8192: ifft(fft(x) * H)[valid] in one workgroup
//                    (reference dispersion.py:135-139, convolution.py:116-120)
//   k_g2_col         N = N1 x N2: column transforms over n1 for a tile of G::ct columns
//   k_g2_row         row k1: four-step twiddle, forward over n2, * H, inverse, conjugate twiddle
//   k_g2_fft_rows    Channelize.task / Dechannelize.task (reference channelize.py:73-74, 164-165)
//
// These are the kernels' BODIES (device functions templated on the geometry traits); the
// __global__ entry points are defined per geometry by the BBT_G2_KERNEL_* macros at the end, in
// the translation unit that the library compiles for a plan at run time (rtc.hpp: the traits come
// from -D options) or in a dev harness (tools/gen2_bench.hip).
// Launch: blockDim.x = G2Info<G>::THREADS; the exchange area is static LDS.
#pragma once
#if !defined(__HIPCC_RTC__)          // (hipRTC provides the runtime's declarations itself)
#include <hip/hip_runtime.h>
#endif
#include "gen_functors.hpp"
#include "fft_gen2.hpp"

namespace bbt {

// BBT_G2_BOUNDS(G, W): W = 0: the registers the kernel needs; W = 4: at most 128, so that two
// workgroups of 7 or 8 waves share a CU (rows of 8100 points: 56.7 -> 45.2 us per 3.9 M-point block
// with 5 spilled dwords; workgroups of up to 4 waves lose: Channelize(1000) 145 -> 128 G)
#define BBT_G2_BOUNDS_0(G) __launch_bounds__(bbt::G2Info<G>::THREADS)
#define BBT_G2_BOUNDS_4(G) __launch_bounds__(bbt::G2Info<G>::THREADS, 4)
#define BBT_G2_BOUNDS(G, W) BBT_G2_BOUNDS_##W(G)
#define BBT_G2_LDS2(G, GR) (bbt::G2Info<G>::LDS > bbt::G2Info<GR>::LDS ? bbt::G2Info<G>::LDS : bbt::G2Info<GR>::LDS)

template <class G, class GR>
__device__ __forceinline__ void g2_osm_small(v2* lds, const float2* __restrict__ in, float2* __restrict__ out,
                                             const OsmChunk& ch, int S, const cf* __restrict__ resp,
                                             const int* __restrict__ resp_index,
                                             const cf* __restrict__ wn, const cf* __restrict__ wnr) {
    constexpr int n = G::n;
    const int npair = S >> 1;
    const int sp = blockIdx.x % npair;
    const OsmBlock blk = osm_block(ch, blockIdx.x / npair);
    GenStreamSrc src{in + (blk.in_off * S + 2 * sp), S, true};
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    GenRespMul mul{resp + (long long)c0 * n, resp + (long long)c1 * n, c0 == c1};
    GenValidDst dst{out + (blk.out_off * S + 2 * sp), S, 0, 1, blk.valid_start, blk.valid_count, true};
    g2_conv_open<G, GR>(lds, wn, wnr, threadIdx.x, src, mul, dst);
}

// Column pass: tile of G::ct columns n2 of one (block, pair), all N1 = G::n rows.
//   grid (tiles * npair * blocks), XCD-contiguous: a tile's 128-byte runs straddle cache lines
//   whenever N2 is not a multiple of 8, so neighbouring tiles share lines and run on one XCD.
//   work element (k1, n2) at ((b*npair+sp)*N1 + k1)*N2p + n2: rows padded to whole lines (N2p a
//   multiple of 8), so that on the work side a tile's run IS one line.
#ifndef BBT_G2_XCD
#define BBT_G2_XCD 1
#endif
template <class G, bool FIRST>
__device__ __forceinline__ void g2_col(v2* lds, const float2* __restrict__ in, float2* __restrict__ out,
                                       float2* __restrict__ work, const OsmChunk& ch, int S, int N2, int N2p,
                                       const cf* __restrict__ wn) {
    constexpr int N1 = G::n, ct = G::ct;
    const int npair = S >> 1;
    const int tid = threadIdx.x;
    const unsigned per_block = gridDim.x / ch.nblk;          // tiles * npair
    const unsigned vb = BBT_G2_XCD ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int b = vb / per_block;
    const unsigned rest = vb - b * per_block;
    const int sp = rest % npair, n2_0 = (rest / npair) * ct;
    const OsmBlock blk = osm_block(ch, b);
    const int n2 = n2_0 + (tid & (ct - 1));
    const bool live = n2 < N2;
    f4* w = reinterpret_cast<f4*>(work) + ((long long)(b * npair + sp) * N1) * N2p + n2;
    if (FIRST) {
        GenStreamSrc src{in + ((blk.in_off + n2) * S + 2 * sp), (long long)N2 * S, live, BBT_G2_NT_LOAD && S == 2};
        GenWorkDst dst{w, N2p, live};
        g2_fft_open<G, -1>(lds, wn, tid, src, dst);
    } else {
        GenWorkSrc src{w, N2p, live};
        GenValidDst dst{out + (blk.out_off * S + 2 * sp), S, n2, N2, blk.valid_start, blk.valid_count, live,
                        BBT_G2_NT_STORE && S == 2};
        g2_fft_open<G, +1>(lds, wn, tid, src, dst);
    }
}

// Row pass, in place on rows k1 of a (block, pair): G::ct neighbouring rows per workgroup (lanes
// over the rows first; gen2_host.hpp g2_row_ct).  grid (ceil(N1 / ct), blocks * npair); see
// k_gen_row.  Threads of rows past N1 run along on the last row and store nothing.
template <class G, class GR>
__device__ __forceinline__ void g2_row(v2* lds, float2* __restrict__ work, int N1, int N2p, const cf* __restrict__ resp,
                                       const int* __restrict__ resp_index, int npair,
                                       const cf* __restrict__ wn, const cf* __restrict__ wnr,
                                       const cf* __restrict__ tlo, const cf* __restrict__ thi,
                                       const cf* __restrict__ tws) {
    constexpr int N2 = G::n, ct = G::ct;
    const int k1_mine = blockIdx.x * ct + (threadIdx.x & (ct - 1));
    const bool live = k1_mine < N1;
    const int k1 = live ? k1_mine : N1 - 1, sp = blockIdx.y % npair;
    f4* row = reinterpret_cast<f4*>(work) + ((long long)blockIdx.y * N1 + k1) * N2p;
    const cf* srow = tws + (long long)k1 * G::fac[0];
    GenRowSrc src{row, tlo, thi, srow, k1};
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    GenRespMul mul{resp + ((long long)c0 * N1 + k1) * N2, resp + ((long long)c1 * N1 + k1) * N2, c0 == c1};
    GenRowDst dst{row, tlo, thi, srow, k1, live};
    g2_conv_open<G, GR>(lds, wn, wnr, threadIdx.x, src, mul, dst);
}

// Batched transforms over contiguous groups of n = G::n complete samples.  A workgroup takes a
// tile of G::ct "columns": cp neighbouring stream pairs (cp * 16 contiguous bytes of every
// complete sample) of G::ct / cp consecutive transforms -- short transforms fill their waves with
// several of them.  grid (ceil(n_fft / (ct / cp)) * (npair / cp)); cp a power of two dividing npair.
template <class G, int SIGN>
__device__ __forceinline__ void g2_fft_rows(v2* lds, const float2* __restrict__ in, float2* __restrict__ out, int S,
                                            int cp, long long n_fft, float scale, const cf* __restrict__ wn) {
    constexpr int n = G::n, ct = G::ct;
    const int npair = S >> 1, npg = npair / cp, lgcp = __ffs(cp) - 1;
    const int tid = threadIdx.x, col = tid & (ct - 1);
    const long long i = (long long)(blockIdx.x / npg) * (ct >> lgcp) + (col >> lgcp);
    const int sp = (blockIdx.x % npg) * cp + (col & (cp - 1));
    const bool live = i < n_fft;
    GenStreamSrc src{in + (i * n * S + 2 * sp), S, live};
    GenScaledDst dst{out + (i * n * S + 2 * sp), S, scale, live};
    g2_fft_open<G, SIGN>(lds, wn, tid, src, dst);
}

}  // namespace bbt

// ---- geometry traits from macros, and the entry points -------------------------------------
//   BBT_G2_TRAIT(GA, n, nfac, tj, ct, (fac...), (pitch...))
#define BBT_G2_UNPAREN(...) __VA_ARGS__
#define BBT_G2_TRAIT(NAME, N_, NFAC_, TJ_, CT_, FAC_, PITCH_)                              \
    struct NAME {                                                                          \
        static constexpr int n = N_, nfac = NFAC_, tj = TJ_, ct = CT_;                     \
        static constexpr int fac[BBT_G2_MAXS] = {BBT_G2_UNPAREN FAC_};                     \
        static constexpr int pitch[BBT_G2_MAXS] = {BBT_G2_UNPAREN PITCH_};                 \
    };
#define BBT_G2_KERNEL_OSM_SMALL(NAME, G, GR, W)                                                            \
    extern "C" __global__ BBT_G2_BOUNDS(G, W) void NAME(const float2* __restrict__ in, float2* __restrict__ out, \
            bbt::OsmChunk ch, int S, const bbt::cf* __restrict__ resp, const int* __restrict__ resp_index,  \
            const bbt::cf* __restrict__ wn, const bbt::cf* __restrict__ wnr) {                              \
        __shared__ bbt::v2 lds[BBT_G2_LDS2(G, GR)];                                                         \
        bbt::g2_osm_small<G, GR>(lds, in, out, ch, S, resp, resp_index, wn, wnr);                           \
    }
#define BBT_G2_KERNEL_COL(NAME, G, FIRST, W)                                                               \
    extern "C" __global__ BBT_G2_BOUNDS(G, W) void NAME(const float2* __restrict__ in, float2* __restrict__ out, \
            float2* __restrict__ work, bbt::OsmChunk ch, int S, int N2, int N2p,                           \
            const bbt::cf* __restrict__ wn) {                                                              \
        __shared__ bbt::v2 lds[bbt::G2Info<G>::LDS];                                                        \
        bbt::g2_col<G, FIRST>(lds, in, out, work, ch, S, N2, N2p, wn);                                      \
    }
#define BBT_G2_KERNEL_ROW(NAME, G, GR, W)                                                                  \
    extern "C" __global__ BBT_G2_BOUNDS(G, W) void NAME(float2* __restrict__ work, int N1, int N2p,             \
            const bbt::cf* __restrict__ resp, const int* __restrict__ resp_index, int npair,               \
            const bbt::cf* __restrict__ wn, const bbt::cf* __restrict__ wnr, const bbt::cf* __restrict__ tlo, \
            const bbt::cf* __restrict__ thi, const bbt::cf* __restrict__ tws) {                             \
        __shared__ bbt::v2 lds[BBT_G2_LDS2(G, GR)];                                                         \
        bbt::g2_row<G, GR>(lds, work, N1, N2p, resp, resp_index, npair, wn, wnr, tlo, thi, tws);              \
    }
#define BBT_G2_KERNEL_FFT_ROWS(NAME, G, SIGN, W)                                                           \
    extern "C" __global__ BBT_G2_BOUNDS(G, W) void NAME(const float2* __restrict__ in, float2* __restrict__ out, \
            int S, int cp, long long n_fft, float scale, const bbt::cf* __restrict__ wn) {                 \
        __shared__ bbt::v2 lds[bbt::G2Info<G>::LDS];                                                        \
        bbt::g2_fft_rows<G, SIGN>(lds, in, out, S, cp, n_fft, scale, wn);                                   \
    }
