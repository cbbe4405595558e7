// Kernels for block / channel counts that are not powers of two
// (n = 2^a 3^b 5^c 7^d), built on the LDS Stockham transform of
// fft_generic.hpp.  Same data contract and the same three-pass structure as the
// power-of-two path in bbt_kernels.hpp:
//
//   k_gen_osm_small   N <= 8192: ifft(fft(x) * H)[valid] in one workgroup
//                     (reference dispersion.py:135-139, convolution.py:116-120)
//   k_gen_col         N = N1 x N2: column transforms over n1 (forward: stream ->
//                     work; inverse: work -> valid output samples)
//   k_gen_row         row k1: four-step twiddle, forward over n2, * H, inverse
//                     over k2, conjugate twiddle, in place
//   k_gen_fft_rows    Channelize.task / Dechannelize.task for such n
//                     (reference channelize.py:73-74, 164-165)
#pragma once
#include <hip/hip_runtime.h>
#include "bbt_kernels.hpp"
#include "fft_generic.hpp"
#include "gen_functors.hpp"

namespace bbt {

// A thread's share of a tile of `total` <= BBT_GEN_EPT * nthr elements: idx = tid + e * nthr.
// The loops below are written over e with a fixed trip count so that all of a thread's global
// loads are in flight together (a loop over idx with a run-time bound issued them one by one,
// each waiting for the one before: these kernels hold one to three workgroups per CU and have
// nothing else to hide that latency with).
#define BBT_GEN_FOR(e, idx, total) \
    _Pragma("unroll") for (int e = 0, idx = tid; e < BBT_GEN_EPT; ++e, idx += nthr)

// One workgroup per (block, pair): n = g.n <= 8192 elements of dynamic LDS.
//   g / gr: the stages and their reversal (forward and inverse transform), wn / wnr their tables
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_gen_osm_small(const float2* __restrict__ in,
                                                        float2* __restrict__ out, OsmChunk ch, int S,
                                                        const cf* __restrict__ resp,
                                                        const int* __restrict__ resp_index, GenGeo g,
                                                        const cf* __restrict__ wn, GenGeo gr,
                                                        const cf* __restrict__ wnr) {
    extern __shared__ f4 gen_lds[];
    const int npair = S >> 1, n = g.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int sp = blockIdx.x % npair;
    const OsmBlock blk = osm_block(ch, blockIdx.x / npair);
    GenStreamSrc src{in + (blk.in_off * S + 2 * sp), S, true};
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    GenRespMul mul{resp + (long long)c0 * n, resp + (long long)c1 * n, c0 == c1};
    GenValidDst dst{out + (blk.out_off * S + 2 * sp), S, 0, 1, blk.valid_start, blk.valid_count, true};
    gen_conv_open(gen_lds, g, gr, 1, wn, wnr, tid, nthr, src, mul, dst);
}

// Column pass: tile of `ct` columns n2 of one (block, pair), all N1 = g.n rows.
//   grid (tiles * npair, blocks); work element (k1, n2) at ((b*npair+sp)*N1 + k1)*N2 + n2.
template <bool FIRST>
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_gen_col(const float2* __restrict__ in,
                                                  float2* __restrict__ out,
                                                  float2* __restrict__ work, OsmChunk ch, int S,
                                                  int N2, int ct, GenGeo g,
                                                  const cf* __restrict__ wn) {
    extern __shared__ f4 gen_lds[];
    const int npair = S >> 1, N1 = g.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int sp = blockIdx.x % npair, n2_0 = (blockIdx.x / npair) * ct;
    const int b = blockIdx.y;
    const OsmBlock blk = osm_block(ch, b);
    const int n2 = n2_0 + (tid & (ct - 1));        // (ct is a power of two that divides nthr)
    const bool live = n2 < N2;
    f4* w = reinterpret_cast<f4*>(work) + ((long long)(b * npair + sp) * N1) * N2 + n2;
    if (FIRST) {
        GenStreamSrc src{in + ((blk.in_off + n2) * S + 2 * sp), (long long)N2 * S, live};
        GenWorkDst dst{w, N2, live};
        gen_fft_open<-1>(gen_lds, g, ct, wn, tid, nthr, src, dst);
    } else {
        GenWorkSrc src{w, N2, live};
        GenValidDst dst{out + (blk.out_off * S + 2 * sp), S, n2, N2, blk.valid_start, blk.valid_count, live};
        gen_fft_open<+1>(gen_lds, g, ct, wn, tid, nthr, src, dst);
    }
}

// Row pass, in place on row k1 of a (block, pair).  grid (N1, blocks * npair); sources and sinks: gen_functors.hpp
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_gen_row(float2* __restrict__ work, int N1,
                                                  const cf* __restrict__ resp,
                                                  const int* __restrict__ resp_index, int npair,
                                                  GenGeo g, const cf* __restrict__ wn, GenGeo gr,
                                                  const cf* __restrict__ wnr,
                                                  const cf* __restrict__ tlo,
                                                  const cf* __restrict__ thi,
                                                  const cf* __restrict__ tws) {
    extern __shared__ f4 gen_lds[];
    const int N2 = g.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int k1 = blockIdx.x, sp = blockIdx.y % npair;
    f4* row = reinterpret_cast<f4*>(work) + ((long long)blockIdx.y * N1 + k1) * N2;
    // tws [N1][fac[0]]: the first forward and the last inverse stage have that radix (and m = N2 / fac[0])
    const cf* srow = tws + (long long)k1 * g.fac[0];
    GenRowSrc src{row, tlo, thi, srow, k1};
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    GenRespMul mul{resp + ((long long)c0 * N1 + k1) * N2, resp + ((long long)c1 * N1 + k1) * N2, c0 == c1};
    GenRowDst dst{row, tlo, thi, srow, k1};
    gen_conv_open(gen_lds, g, gr, 1, wn, wnr, tid, nthr, src, mul, dst);
}

// Batched transforms over contiguous groups of n = g.n complete samples, for a
// tile of `ct` stream pairs (ct * 16 contiguous bytes per complete sample).
//   grid (n_fft * (npair / ct))
template <int SIGN>
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_gen_fft_rows(const float2* __restrict__ in,
                                                       float2* __restrict__ out, int S, int ct,
                                                       float scale, GenGeo g,
                                                       const cf* __restrict__ wn) {
    extern __shared__ f4 gen_lds[];
    const int npair = S >> 1, n = g.n, npg = npair / ct;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long long i = blockIdx.x / npg;
    const int sp = (blockIdx.x % npg) * ct + (tid & (ct - 1));       // (ct: a power of two dividing nthr)
    GenStreamSrc src{in + (i * n * S + 2 * sp), S, true};
    GenScaledDst dst{out + (i * n * S + 2 * sp), S, scale};
    gen_fft_open<SIGN>(gen_lds, g, ct, wn, tid, nthr, src, dst);
}

// Splice the spectrum that straddles a block seam (see k_seam_fix) for any
// channel count n = g.n: both blocks' versions go through the inverse transform
// side by side (tile of two), samples [0, split) are taken from the earlier
// block, the rest from the later one, and the result is transformed again.
// One workgroup per (seam, pair); 2 n elements of dynamic LDS.
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_seam_fix_gen(const float2* __restrict__ seam,
                                                       float2* __restrict__ out, SeamJobs jobs, int S,
                                                       int npair, GenGeo g,
                                                       const cf* __restrict__ wn, SpecOut so) {
    extern __shared__ f4 gen_lds[];
    const int n = g.n, tid = threadIdx.x, nthr = blockDim.x, sp = blockIdx.y;
    const SeamJob job = jobs.j[blockIdx.x];
    const float2* za = seam + ((((long long)job.first_block * 2 + 1) * npair + sp) * n) * 2;
    const float2* zb = seam + ((((long long)(job.first_block + 1) * 2 + 0) * npair + sp) * n) * 2;
    for (int i = tid; i < n; i += nthr) {
        gen_lds[2 * i] = ld_ext_f4(za + 2 * i);
        gen_lds[2 * i + 1] = ld_ext_f4(zb + 2 * i);
    }
    __syncthreads();
    gen_fft<+1>(gen_lds, g, 2, wn, tid, nthr);
    const float scale = 1.0f / (float)n;
    for (int i = tid; i < n; i += nthr) gen_lds[2 * i] = gen_lds[2 * i + (i < job.split ? 0 : 1)] * scale;
    __syncthreads();
    gen_fft<-1>(gen_lds, g, 2, wn, tid, nthr);
    if (so.det) {
        const long long bin = job.spectrum / so.det_step;
        if (bin >= so.n_out / so.det_step) return;
        for (int i = tid; i < n; i += nthr) {
            const float4 pw = detect_pair(f4_to_c2(gen_lds[2 * i]), so.det_mode);
            float* dst = so.det + detect_index(bin, i, sp, so.lg_chan, npair, so.det_mode);
            unsafeAtomicAdd(dst + 0, pw.x * so.det_scale);
            unsafeAtomicAdd(dst + 1, pw.y * so.det_scale);
            if (so.det_mode) {
                unsafeAtomicAdd(dst + 2, pw.z * so.det_scale);
                unsafeAtomicAdd(dst + 3, pw.w * so.det_scale);
            }
        }
        return;
    }
    float2* dst = out + ((job.spectrum * n) * S + 2 * sp);
    for (int i = tid; i < n; i += nthr) st_ext_f4(dst + (long long)i * S, gen_lds[2 * i]);
}

}  // namespace bbt
