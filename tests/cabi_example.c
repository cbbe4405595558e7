/* A plain C host program on include/bbt_hip.h: what a non-Python caller of the
 * path (or the reference's maintainer writing a C extension) would link
 * against libbbt_hip.so.  Built with gcc by tests/test_cabi.py (link check on
 * the CPU) and run on the GPU by tests/test_gpu_parity.py.
 *
 * It dedisperses two blocks of a unit impulse with an all-pass "chirp" that is a
 * pure delay of 3 samples (H[k] = exp(-2 pi i 3 k / N)), so the output must be
 * the impulse moved by 3 samples -- ifft(fft(x) * H)[valid], the operation of
 * Disperse.task (dispersion.py:135-139) -- and channelizes a constant stream
 * (Channelize.task, channelize.py:73-74: bin 0 = n, all other bins 0). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bbt_hip.h"

#define CHECK(call)                                                      \
    do {                                                                 \
        if ((call) != 0) {                                               \
            fprintf(stderr, "%s failed: %s\n", #call, bbt_last_error()); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(void) {
    enum { N = 4096, S = 2, PAD = 16, VALID = N - PAD, NBLK = 2, NCH = 8 };
    int count = 0;
    CHECK(bbt_device_count(&count));
    if (count < 1) {
        fprintf(stderr, "no GPU\n");
        return 2;
    }
    CHECK(bbt_set_device(0));
    printf("libbbt_hip version %d\n", bbt_version());

    /* response: delay by 3 samples, one column shared by both streams */
    float* resp = (float*)malloc(sizeof(float) * 2 * N);
    for (int k = 0; k < N; ++k) {
        const double a = -2.0 * M_PI * 3.0 * (double)(k < N / 2 ? k : k - N) / N;
        resp[2 * k] = (float)cos(a);
        resp[2 * k + 1] = (float)sin(a);
    }
    bbt_osm_plan* plan = NULL;
    CHECK(bbt_osm_plan_create(&plan, N, S, 1, resp, 0, NULL));

    /* input: (n_in, S) complex64, impulses at samples 100 (stream 0) and 5000 (stream 1) */
    const long long n_in = (long long)(NBLK - 1) * VALID + N, n_out = (long long)NBLK * VALID;
    float* x = (float*)calloc((size_t)n_in * S * 2, sizeof(float));
    x[(100 * S + 0) * 2] = 1.0f;
    x[(5000 * S + 1) * 2 + 1] = 2.0f; /* 2i */
    void *dx = NULL, *dy = NULL;
    CHECK(bbt_malloc(&dx, (size_t)n_in * S * 8));
    CHECK(bbt_malloc(&dy, (size_t)n_out * S * 8));
    CHECK(bbt_memcpy_h2d(dx, x, (size_t)n_in * S * 8, NULL));
    /* block b reads [b * VALID, b * VALID + N), keeps samples [PAD, N) -> output [b * VALID, ...) */
    CHECK(bbt_osm_execute_regular(plan, dx, dy, NBLK, 0, 0, VALID, PAD, NULL));
    float* y = (float*)malloc((size_t)n_out * S * 8);
    CHECK(bbt_memcpy_d2h(y, dy, (size_t)n_out * S * 8, NULL));
    CHECK(bbt_stream_sync(NULL));
    /* output sample j is block sample j + PAD: the impulse at input 100 delayed by 3 lands at j = 100 + 3 - PAD */
    double err = 0.0;
    for (long long j = 0; j < n_out; ++j)
        for (int s = 0; s < S; ++s) {
            double re = 0.0, im = 0.0;
            if (s == 0 && j == 100 + 3 - PAD) re = 1.0;
            if (s == 1 && j == 5000 + 3 - PAD) im = 2.0;
            const double dr = y[(j * S + s) * 2] - re, di = y[(j * S + s) * 2 + 1] - im;
            if (fabs(dr) > err) err = fabs(dr);
            if (fabs(di) > err) err = fabs(di);
        }
    printf("overlap-save delay filter: max |error| %.2e\n", err);
    if (err > 2e-6) return 3;

    /* channelizer: constant 1 + 0i -> bin 0 = NCH */
    bbt_chan_plan* chan = NULL;
    CHECK(bbt_chan_plan_create(&chan, NCH, S, -1));
    for (long long i = 0; i < 4 * NCH * S; ++i) {
        x[2 * i] = 1.0f;
        x[2 * i + 1] = 0.0f;
    }
    CHECK(bbt_memcpy_h2d(dx, x, (size_t)4 * NCH * S * 8, NULL));
    CHECK(bbt_chan_execute(chan, dx, dy, 4, NULL));
    CHECK(bbt_memcpy_d2h(y, dy, (size_t)4 * NCH * S * 8, NULL));
    CHECK(bbt_stream_sync(NULL));
    err = 0.0;
    for (int i = 0; i < 4 * NCH * S; ++i) {
        const int chn = (i / S) % NCH;
        const double want = chn == 0 ? (double)NCH : 0.0;
        if (fabs(y[2 * i] - want) > err) err = fabs(y[2 * i] - want);
        if (fabs(y[2 * i + 1]) > err) err = fabs(y[2 * i + 1]);
    }
    printf("channelizer of a constant: max |error| %.2e\n", err);
    if (err > 1e-5) return 4;

    /* argument errors are reported, not thrown */
    if (bbt_osm_execute_regular(plan, dx, dy, 1, 0, 0, 2 * N, 0, NULL) == 0) return 5;
    printf("expected error: %s\n", bbt_last_error());

    CHECK(bbt_chan_plan_destroy(chan));
    CHECK(bbt_osm_plan_destroy(plan));
    CHECK(bbt_free(dx));
    CHECK(bbt_free(dy));
    free(x);
    free(y);
    free(resp);
    printf("C ABI example OK\n");
    return 0;
}
