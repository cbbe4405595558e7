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

namespace bbt {

// A thread's share of a tile of `total` <= BBT_GEN_EPT * nthr elements: idx = tid + e * nthr.
// The loops below are written over e with a fixed trip count so that all of a thread's global
// loads are in flight together (a loop over idx with a run-time bound issued them one by one,
// each waiting for the one before: these kernels hold one to three workgroups per CU and have
// nothing else to hide that latency with).
#define BBT_GEN_FOR(e, idx, total) \
    _Pragma("unroll") for (int e = 0, idx = tid; e < BBT_GEN_EPT; ++e, idx += nthr)

__device__ __forceinline__ f4 ld_ext_f4(const float2* p) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    return f4{x.x, x.z, x.y, x.w};
}
__device__ __forceinline__ void st_ext_f4(float2* p, f4 a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.x, a.z, a.y, a.w);
}
__device__ __forceinline__ f4 f4_mul_resp(f4 a, cf x, cf y) {     // stream A times x, stream B times y
    return f4{a.x * x.x - a.z * x.y, a.y * y.x - a.w * y.y, a.x * x.y + a.z * x.x, a.y * y.y + a.w * y.x};
}
__device__ __forceinline__ f4 f4_twmul(f4 a, cf w) { return f4_mul_resp(a, w, w); }

// One workgroup per (block, pair): n = g.n <= 8192 elements of dynamic LDS.
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_gen_osm_small(const float2* __restrict__ in,
                                                        float2* __restrict__ out, OsmChunk ch, int S,
                                                        const cf* __restrict__ resp,
                                                        const int* __restrict__ resp_index, GenGeo g,
                                                        const cf* __restrict__ wn) {
    extern __shared__ f4 gen_lds[];
    const int npair = S >> 1, n = g.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int sp = blockIdx.x % npair;
    const OsmBlock blk = ch.b[blockIdx.x / npair];
    const float2* src = in + (blk.in_off * S + 2 * sp);
    {
        f4 x[BBT_GEN_EPT];
        BBT_GEN_FOR(e, i, n) x[e] = i < n ? ld_ext_f4(src + (long long)i * S) : f4{0.f, 0.f, 0.f, 0.f};
        BBT_GEN_FOR(e, i, n) if (i < n) gen_lds[i] = x[e];
    }
    __syncthreads();
    const cf* h0 = resp + (long long)resp_index[2 * sp] * n;
    const cf* h1 = resp + (long long)resp_index[2 * sp + 1] * n;
    gen_fft<-1>(gen_lds, g, 1, wn, tid, nthr);
    {
        cf ha[BBT_GEN_EPT], hb[BBT_GEN_EPT];
        BBT_GEN_FOR(e, i, n) {
            ha[e] = i < n ? h0[i] : make_float2(0.f, 0.f);
            hb[e] = i < n ? h1[i] : make_float2(0.f, 0.f);
        }
        BBT_GEN_FOR(e, i, n) if (i < n) gen_lds[i] = f4_mul_resp(gen_lds[i], ha[e], hb[e]);
    }
    __syncthreads();
    gen_fft<+1>(gen_lds, g, 1, wn, tid, nthr);
    BBT_GEN_FOR(e, i, n) {
        const int r = i - blk.valid_start;
        if (i < n && r >= 0 && r < blk.valid_count) st_ext_f4(out + ((blk.out_off + r) * S + 2 * sp), gen_lds[i]);
    }
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
    const OsmBlock blk = ch.b[b];
    f4* w = reinterpret_cast<f4*>(work) + ((long long)(b * npair + sp) * N1) * N2;
    const int total = N1 * ct;
    const int lg = __ffs(ct) - 1;                  // (ct is a power of two that divides nthr)
    const int n2 = n2_0 + (tid & (ct - 1)), row_step = nthr >> lg;
    if (n2 < N2) {
        // (four loads in flight per thread; with all eight the allocator spilled 700 dwords)
        const long long step = (long long)row_step * N2;
        if (FIRST) {
            const float2* src = in + ((blk.in_off + (long long)(tid >> lg) * N2 + n2) * S + 2 * sp);
#pragma unroll 4
            for (int idx = tid; idx < total; idx += nthr, src += step * S) gen_lds[idx] = ld_ext_f4(src);
        } else {
            const f4* src = w + (long long)(tid >> lg) * N2 + n2;
#pragma unroll 4
            for (int idx = tid; idx < total; idx += nthr, src += step) gen_lds[idx] = *src;
        }
    } else {
        for (int idx = tid; idx < total; idx += nthr) gen_lds[idx] = f4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    gen_fft<FIRST ? -1 : +1>(gen_lds, g, ct, wn, tid, nthr);
    if (n2 >= N2) return;
    int n1 = tid >> lg;
    for (int idx = tid; idx < total; idx += nthr, n1 += row_step) {
        if (FIRST) {
            w[(long long)n1 * N2 + n2] = gen_lds[idx];
        } else {
            const long long r = (long long)n1 * N2 + n2 - blk.valid_start;
            if (r >= 0 && r < blk.valid_count) st_ext_f4(out + ((blk.out_off + r) * S + 2 * sp), gen_lds[idx]);
        }
    }
}

// Row pass, in place on row k1 of a (block, pair).  grid (N1, blocks * npair).
//   resp  : [C][N1][N2] = H[c][k1 + N1 k2] / N
//   wn    : W_{N2}^k ;  tlo / thi : W_N^m tables (big_twiddle)
__global__ __launch_bounds__(BBT_GEN_MAX_THREADS) void k_gen_row(float2* __restrict__ work, int N1,
                                                  const cf* __restrict__ resp,
                                                  const int* __restrict__ resp_index, int npair,
                                                  GenGeo g, const cf* __restrict__ wn,
                                                  const cf* __restrict__ tlo,
                                                  const cf* __restrict__ thi) {
    extern __shared__ f4 gen_lds[];
    const int N2 = g.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int k1 = blockIdx.x, sp = blockIdx.y % npair;
    f4* row = reinterpret_cast<f4*>(work) + ((long long)blockIdx.y * N1 + k1) * N2;
    // The four-step twiddles W_N^{k1 n2} of a thread's elements n2 = tid + e nthr are
    // a s^e with a = W_N^{k1 tid} and s = W_N^{k1 nthr} (the same for the whole workgroup): two
    // table look-ups and products at most four roundings deep instead of a look-up (two loads)
    // per element and direction -- like the stage twiddles, loads were what this kernel waited for.
    auto four_step = [&](cf (&tw)[BBT_GEN_EPT]) {
        const cf a = big_twiddle(tlo, thi, k1 * tid);
        cf sp_[BBT_GEN_EPT];
        sp_[1] = big_twiddle(tlo, thi, (int)(((long long)k1 * nthr) % ((long long)N1 * N2)));
#pragma unroll
        for (int e = 2; e < BBT_GEN_EPT; ++e) sp_[e] = cmul(sp_[(e + 1) / 2], sp_[e / 2]);
        tw[0] = a;
#pragma unroll
        for (int e = 1; e < BBT_GEN_EPT; ++e) tw[e] = cmul(a, sp_[e]);
    };
    {
        f4 x[BBT_GEN_EPT];
        cf tw[BBT_GEN_EPT];
        BBT_GEN_FOR(e, i, N2) x[e] = i < N2 ? row[i] : f4{0.f, 0.f, 0.f, 0.f};
        four_step(tw);
        BBT_GEN_FOR(e, i, N2) if (i < N2) gen_lds[i] = f4_twmul(x[e], tw[e]);
    }
    __syncthreads();
    gen_fft<-1>(gen_lds, g, 1, wn, tid, nthr);
    {
        const int c0 = resp_index[2 * sp], c1 = resp_index[2 * sp + 1];
        const cf* h0 = resp + ((long long)c0 * N1 + k1) * N2;
        const cf* h1 = resp + ((long long)c1 * N1 + k1) * N2;
        cf ha[BBT_GEN_EPT], hb[BBT_GEN_EPT];
        if (c0 == c1) {                                  // (both streams of the pair: one column)
            BBT_GEN_FOR(e, i, N2) ha[e] = i < N2 ? h0[i] : make_float2(0.f, 0.f);
            BBT_GEN_FOR(e, i, N2) if (i < N2) gen_lds[i] = f4_mul_resp(gen_lds[i], ha[e], ha[e]);
        } else {
            BBT_GEN_FOR(e, i, N2) {
                ha[e] = i < N2 ? h0[i] : make_float2(0.f, 0.f);
                hb[e] = i < N2 ? h1[i] : make_float2(0.f, 0.f);
            }
            BBT_GEN_FOR(e, i, N2) if (i < N2) gen_lds[i] = f4_mul_resp(gen_lds[i], ha[e], hb[e]);
        }
    }
    __syncthreads();
    gen_fft<+1>(gen_lds, g, 1, wn, tid, nthr);
    {
        cf tw[BBT_GEN_EPT];
        four_step(tw);
        BBT_GEN_FOR(e, i, N2) if (i < N2) row[i] = f4_twmul(gen_lds[i], make_float2(tw[e].x, -tw[e].y));
    }
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
    const int sp0 = (blockIdx.x % npg) * ct;
    const int total = n * ct;
    const float2* src = in + (i * n * S + 2 * sp0);
    const int lg = __ffs(ct) - 1;                  // (ct is a power of two that divides nthr)
    const int c = tid & (ct - 1), row0 = tid >> lg, row_step = nthr >> lg;
    {
        f4 x[BBT_GEN_EPT];
        BBT_GEN_FOR(e, idx, total)
            x[e] = idx < total ? ld_ext_f4(src + ((long long)(row0 + e * row_step) * S + 2 * c)) : f4{0.f, 0.f, 0.f, 0.f};
        BBT_GEN_FOR(e, idx, total) if (idx < total) gen_lds[idx] = x[e];
    }
    __syncthreads();
    gen_fft<SIGN>(gen_lds, g, ct, wn, tid, nthr);
    float2* dst = out + (i * n * S + 2 * sp0);
    BBT_GEN_FOR(e, idx, total)
        if (idx < total) st_ext_f4(dst + ((long long)(row0 + e * row_step) * S + 2 * c), gen_lds[idx] * scale);
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
