// Access-pattern ceilings for the overlap-save passes (measurement aid, not product code).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench tools/membench.hip && /tmp/membench
// Pattern "tile C": a workgroup copies a tile of C columns x 256 rows of a [256][4096] float4 matrix
// (the column passes' shape: runs of C*16 bytes, 64 KiB apart); "rows": a workgroup copies one
// contiguous 64 KiB row (the row pass's shape).  16 float4 per thread in flight, like the FFT kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N1 = 256, N2 = 4096;

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int per = nblocks >> 3;
    return (bid & 7) * per + (bid >> 3);
}

template <int C, bool REMAP>
__global__ __launch_bounds__(C * 16) void k_tile(const float4* __restrict__ in, float4* __restrict__ out,
                                                 int tiles_per_block) {
    int bid = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int blk = bid / tiles_per_block, tile = bid % tiles_per_block;
    const int c = threadIdx.x % C, r0 = threadIdx.x / C;       // r0 < 16
    const long long base = (long long)blk * N1 * N2 + tile * C + c;
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = in[base + (long long)(r0 + 16 * j) * N2];
#pragma unroll
    for (int j = 0; j < 16; ++j) out[base + (long long)(r0 + 16 * j) * N2] = v[j];
}

template <bool INPLACE>
__global__ __launch_bounds__(256) void k_rows(const float4* __restrict__ in, float4* __restrict__ out) {
    const long long base = (long long)blockIdx.x * N2 + threadIdx.x;
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = in[base + 256 * j];
#pragma unroll
    for (int j = 0; j < 16; ++j) out[base + 256 * j] = v[j];
}

template <typename F>
static void run(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.2f us/launch  %6.2f TB/s (read+write)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

// producer -> consumer through the caches: one kernel writes a buffer, the next reads it
__global__ __launch_bounds__(256) void k_fill(float4* __restrict__ out, float v) {
    const long long base = (long long)blockIdx.x * N2 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; ++j) out[base + 256 * j] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_sum(const float4* __restrict__ in, float* __restrict__ sink) {
    const long long base = (long long)blockIdx.x * N2 + threadIdx.x;
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = in[base + 256 * j];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
    if (s == 123.456f) sink[0] = s;      // never true: keeps the loads
}

static void producer_consumer() {
    printf("write a buffer, then read it back with the next kernel (MiB: write TB/s, read TB/s)\n");
    float* sink;
    CHECK(hipMalloc(&sink, 4));
    for (int mib : {32, 64, 128, 192, 256, 384, 512, 1024}) {
        const size_t elems = (size_t)mib * 65536;            // float4 per MiB
        float4* buf;
        CHECK(hipMalloc(&buf, elems * 16));
        const int rows = (int)(elems / N2);
        hipEvent_t e0, e1, e2;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
        float wr = 0.f, rd = 0.f;
        const int reps = 10;
        for (int i = 0; i < reps + 2; ++i) {
            CHECK(hipEventRecord(e0));
            k_fill<<<rows, 256>>>(buf, (float)i);
            CHECK(hipEventRecord(e1));
            k_sum<<<rows, 256>>>(buf, sink);
            CHECK(hipEventRecord(e2));
            CHECK(hipEventSynchronize(e2));
            float a, b;
            CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
            if (i >= 2) { wr += a; rd += b; }
        }
        const double bytes = (double)elems * 16;
        printf("  %5d MiB: write %5.2f  read %5.2f\n", mib, bytes / (wr / reps * 1e-3) / 1e12,
               bytes / (rd / reps * 1e-3) / 1e12);
        CHECK(hipFree(buf));
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 0) {
        producer_consumer();
        return 0;
    }
    const int nblk = argc > 1 ? atoi(argv[1]) : 4;              // blocks of 16 MiB per launch
    const int nbuf = argc > 2 ? atoi(argv[2]) : 1;              // rotate over this many buffer pairs (cache state)
    const size_t elems = (size_t)nblk * N1 * N2;
    float4 *in[16], *out[16];
    for (int b = 0; b < nbuf; ++b) {
        CHECK(hipMalloc(&in[b], elems * 16)); CHECK(hipMalloc(&out[b], elems * 16));
        CHECK(hipMemset(in[b], 1, elems * 16)); CHECK(hipMemset(out[b], 0, elems * 16));
    }
    const double bytes = 2.0 * elems * 16;
    int it = 0;
    printf("blocks per launch %d (%.0f MiB in + out), rotating over %d buffer pairs\n", nblk, bytes / 1048576, nbuf);
    run("rows (64 KiB contiguous)", [&] { int b = it++ % nbuf;
        k_rows<false><<<nblk * N1, 256>>>(in[b], out[b]); }, bytes);
    run("rows in place", [&] { int b = it++ % nbuf;
        k_rows<true><<<nblk * N1, 256>>>(out[b], out[b]); }, bytes);
    run("tile 16 cols (256 B runs)", [&] { int b = it++ % nbuf;
        k_tile<16, false><<<nblk * N2 / 16, 256>>>(in[b], out[b], N2 / 16); }, bytes);
    run("tile 16 cols, xcd remap", [&] { int b = it++ % nbuf;
        k_tile<16, true><<<nblk * N2 / 16, 256>>>(in[b], out[b], N2 / 16); }, bytes);
    run("tile 32 cols (512 B runs)", [&] { int b = it++ % nbuf;
        k_tile<32, false><<<nblk * N2 / 32, 512>>>(in[b], out[b], N2 / 32); }, bytes);
    run("tile 32 cols, xcd remap", [&] { int b = it++ % nbuf;
        k_tile<32, true><<<nblk * N2 / 32, 512>>>(in[b], out[b], N2 / 32); }, bytes);
    run("tile 64 cols (1 KiB runs)", [&] { int b = it++ % nbuf;
        k_tile<64, false><<<nblk * N2 / 64, 1024>>>(in[b], out[b], N2 / 64); }, bytes);
    run("tile 64 cols, xcd remap", [&] { int b = it++ % nbuf;
        k_tile<64, true><<<nblk * N2 / 64, 1024>>>(in[b], out[b], N2 / 64); }, bytes);
    return 0;
}
