// One-kernel overlap-save for blocks of 8192 / 16384 samples: k_osm_small (bbt_kernels.hpp) with the
// four-stage transform of fft_big.hpp -- ifft(fft(x) * H)[valid] of one (block, pair) per workgroup
// (reference dispersion.py:135-139, convolution.py:116-120), one pass over the stream where the
// 16 x 512 / 16 x 1024 plans take three.
#pragma once
#include "bbt_kernels.hpp"
#include "fft_big.hpp"

namespace bbt {

// SINGLE (one stream): the two transforms a thread carries are two consecutive blocks (SinglePair).
template <int N, bool SINGLE = false>
__global__ __launch_bounds__(N / 16, 4) void k_osm_small_big(const float2* __restrict__ in,
                                                              float2* __restrict__ out, OsmChunk ch, int S,
                                                              const cf* __restrict__ resp,
                                                              const int* __restrict__ resp_index,
                                                              const cf* __restrict__ tw) {
    constexpr int T = BigGeo<N>::T;
    extern __shared__ v2 big_lds[];
    const int tau = threadIdx.x, npair = S >> 1;
    // the pairs of one block share cache lines: consecutive virtual ids, one XCD
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    c2 v[16];
    if constexpr (SINGLE) {                        // workgroup vb = pair of blocks
        const SinglePair pr = single_pair(ch, vb);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld_single(in, pr, tau + T * j);
        wg_fft_big<N, -1>(v, big_lds, tau, tw);
        const cf* h = resp + (long long)resp_index[0] * N + tau;
        __builtin_amdgcn_sched_barrier(0);
        apply_resp<T>(v, h, h, true);
        __builtin_amdgcn_sched_barrier(0);
        const cf* twb = tw;
        asm volatile("" : "+s"(twb));
        wg_fft_big<N, +1>(v, big_lds, tau, twb);
#pragma unroll
        for (int j = 0; j < 16; ++j) st_single(out, pr, tau + T * j, v[j]);
        return;
    }
    const int sp = vb % npair;
    const OsmBlock blk = osm_block(ch, vb / npair);
    const float2* src = in + ((blk.in_off + tau) * S + 2 * sp);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = ld_ext(src + (long long)T * j * S);
    wg_fft_big<N, -1>(v, big_lds, tau, tw);
    const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
    const cf* h0 = resp + (long long)c0 * N + tau;
    const cf* h1 = resp + (long long)c1 * N + tau;
    __builtin_amdgcn_sched_barrier(0);
    apply_resp<T>(v, h0, h1, c0 == c1);
    __builtin_amdgcn_sched_barrier(0);
    {
        const cf* twb = tw;
        asm volatile("" : "+s"(twb));          // (or the first transform's table values stay in registers)
        wg_fft_big<N, +1>(v, big_lds, tau, twb);
    }
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

}  // namespace bbt
