// Sources, multipliers and sinks of the open-ended generic-length transforms (fft_generic.hpp,
// fft_gen2.hpp): where a transform's first stage takes its butterflies from and where its last
// stage puts them.  Elements i = j + r m of one column of stream pairs, m apart:
//
//   Src:  template <int R> void load(int j, int m, c2 (&v)[R])    v[r] = x[j + r m]
//   Mul:  template <int R> void apply(int j, int m, c2 (&v)[R])   v[r] *= h[j + r m]
//   Dst:  template <int R> void store(int j, int m, c2 (&v)[R])   y[j + r m] = v[r]
#pragma once
#if !defined(__HIPCC_RTC__)          // (hipRTC provides the runtime's declarations itself)
#include <hip/hip_runtime.h>
#endif
#include "osm_chunk.hpp"
#include "fft_generic.hpp"

namespace bbt {

// (dev switches of the run-time compiled kernels: BBT_RTC_DEFINES="-DBBT_G2_NT_LOAD=1 ...")
#ifndef BBT_G2_NT_LOAD
#define BBT_G2_NT_LOAD 0                 // stream side of the first column pass non-temporal when S == 2
#endif
#ifndef BBT_G2_NT_STORE
#define BBT_G2_NT_STORE 0                // ... of the last one
#endif
#ifndef BBT_G2_WORK_ST
#define BBT_G2_WORK_ST 0                 // work-buffer stores write-through (st_int)
#endif

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

// ---- sources, multipliers and sinks of the open-ended transforms (fft_generic.hpp) ----------
// elements i = j + r m of one column of stream pairs, m apart

// rows of a (n, S) stream: element i at base[i * stride] (stride in float2 units), external format
struct GenStreamSrc {
    const float2* base;
    long long stride;
    bool live;                           // (a column past the edge of the last tile reads zeros)
    bool nt = false;                     // a stream read once in whole lines (S == 2): non-temporal, as ld_ext_nt
    template <int R>
    __device__ __forceinline__ void load(int j, int m, c2 (&v)[R]) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float2* p = base + (long long)(j + r * m) * stride;
#if defined(BBT_DBG_NOLOAD)              // (timing experiment: what the transform costs without its loads)
            v[r] = c2{v2{(float)j, (float)r}, v2{(float)m, 1.f}};
#else
            v[r] = !live ? czero() : nt ? ld_ext_nt(p) : f4_to_c2(ld_ext_f4(p));
#endif
        }
    }
};
// the same for the work buffer (internal format, f4 units)
struct GenWorkSrc {
    const f4* base;
    long long stride;
    bool live;
    template <int R>
    __device__ __forceinline__ void load(int j, int m, c2 (&v)[R]) const {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = live ? f4_to_c2(base[(long long)(j + r * m) * stride]) : czero();
    }
};
struct GenWorkDst {
    f4* base;
    long long stride;
    bool live;
    template <int R>
    __device__ __forceinline__ void store(int j, int m, c2 (&v)[R]) const {
        if (!live) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f4* p = base + (long long)(j + r * m) * stride;
#if defined(BBT_DBG_NOSTORE)
            if (v[r].re.x != 12345.678f) continue;
#endif
#if BBT_G2_WORK_ST
            st_int(reinterpret_cast<float2*>(p), v[r]);        // (write-through, as the power-of-two passes)
#else
            *p = c2_to_f4(v[r]);
#endif
        }
    }
};
// kept samples of an overlap-save block: element i of the block goes to output row i - valid_start
struct GenValidDst {
    float2* out;                         // out + (out_off * S + 2 sp), external format
    long long stride;                    // S
    long long first, step;               // block sample of element i: first + i * step
    int valid_start, valid_count;
    bool live;
    bool nt = false;                     // whole lines written once (S == 2): non-temporal
    template <int R>
    __device__ __forceinline__ void store(int j, int m, c2 (&v)[R]) const {
        if (!live) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long q = first + (long long)(j + r * m) * step - valid_start;
            if (q >= 0 && q < valid_count) st_ext(out + q * stride, v[r], nt);
        }
    }
};
// spectral multiply: element i times the response columns of the pair's two streams
struct GenRespMul {
    const cf* h0;
    const cf* h1;
    bool same;
    template <int R>
    __device__ __forceinline__ void apply(int j, int m, c2 (&v)[R]) const {
        if (same) {                      // (both streams of the pair share a column: a twiddle-like product)
            cf x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = h0[j + r * m];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = twmul_v<-1>(v[r], x[r]);
            return;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const cf x = h0[j + r * m], y = h1[j + r * m];
            v[r] = cmul2(v[r], c2{v2{x.x, y.x}, v2{x.y, y.y}});
        }
    }
};

// Row pass of a two-level plan: in place on row k1 of a (block, pair).
//   resp  : [C][N1][N2] = H[c][k1 + N1 k2] / N
//   g / gr, wn / wnr: stages of the N2-point transform and their reversal;  tlo / thi : W_N^m
// The four-step twiddles W_N^{k1 i} of a butterfly's elements i = j + r m are a s_r with
// a = W_N^{k1 j} (one look-up per butterfly) and s_r = W_N^{k1 m r}, the same for the whole
// workgroup: a row of a small table made in double at plan creation (srow[r], r < R: scalar
// loads; forming them as powers of s_1 cost R - 2 complex products per butterfly and three
// more roundings).  Folded into the source and, conjugated, into the sink.
struct GenRowSrc {
    const f4* row;
    const cf* tlo;
    const cf* thi;
    const cf* srow;
    int k1;
    template <int R>
    __device__ __forceinline__ void load(int j, int m, c2 (&v)[R]) const {
        const cf a = big_twiddle(tlo, thi, k1 * j);
        v[0] = twmul_v<-1>(f4_to_c2(row[j]), a);
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = twmul_v<-1>(f4_to_c2(row[j + r * m]), cmul(a, srow[r]));
    }
};
struct GenRowDst {
    f4* row;
    const cf* tlo;
    const cf* thi;
    const cf* srow;
    int k1;
    bool live = true;
    template <int R>
    __device__ __forceinline__ void store(int j, int m, c2 (&v)[R]) const {
        if (!live) return;
        const cf a = big_twiddle(tlo, thi, k1 * j);
#if BBT_G2_WORK_ST
        st_int(reinterpret_cast<float2*>(row + j), twmul_v<+1>(v[0], a));
#pragma unroll
        for (int r = 1; r < R; ++r) st_int(reinterpret_cast<float2*>(row + j + r * m), twmul_v<+1>(v[r], cmul(a, srow[r])));
#else
        row[j] = c2_to_f4(twmul_v<+1>(v[0], a));
#pragma unroll
        for (int r = 1; r < R; ++r) row[j + r * m] = c2_to_f4(twmul_v<+1>(v[r], cmul(a, srow[r])));
#endif
    }
};
struct GenScaledDst {
    float2* base;
    long long stride;
    float scale;
    bool live = true;
    template <int R>
    __device__ __forceinline__ void store(int j, int m, c2 (&v)[R]) const {
        if (!live) return;
#pragma unroll
        for (int r = 0; r < R; ++r)
#if defined(BBT_DBG_NOSTORE)             // (timing experiment: ... without its stores)
            if (v[r].re.x == 12345.678f)
#endif
            st_ext_f4(base + (long long)(j + r * m) * stride, c2_to_f4(v[r]) * scale);
    }
};

}  // namespace bbt
