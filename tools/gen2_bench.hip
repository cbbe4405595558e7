// Stand-alone A/B of the generic-length kernels: LDS Stockham engine (gen_kernels.hpp) against the
// register-resident engine (gen2_kernels.hpp), each checked against a float64 DFT on the host.
// Dev tool (quick to build; the library takes 90 s):
//
// The new engine's kernels are compiled at run time for the length asked for (rtc.hpp), as the
// library does.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I baseband-tasks_amd/csrc tools/gen2_bench.hip -o build/gen2_bench -ldl
//   build/gen2_bench row  <N1> <N2> [blocks=3] [reps=20]
//   build/gen2_bench chan <n>  [spectra=4096] [reps=20]
//   build/gen2_bench col  <N1> <N2> [blocks=3] [reps=20] [ct=8]
#include <hip/hip_runtime.h>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <chrono>
#include "gen_kernels.hpp"
#include "gen2_host.hpp"
#include "rtc.hpp"

using namespace bbt;
typedef std::complex<double> cd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T> static T* upload(const std::vector<T>& h) {
    T* d;
    CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static cf croot(long long m, long long n) { cf w; g2_root(m, n, &w.x, &w.y); return w; }
static cf* upload_tables(const G2Plan& g) {
    std::vector<float> t = g2_tables(g);
    float* d = upload(t);
    return reinterpret_cast<cf*>(d);
}
static hipFunction_t build(const std::string& src, const char* name) {
    RtcModule* m;
    std::string log;
    const auto t0 = std::chrono::steady_clock::now();
    if (rtc_module(src, &m, &log)) { fprintf(stderr, "RTC failed:\n%s\n%s\n", src.c_str(), log.c_str()); exit(1); }
    hipFunction_t f;
    if (rtc_function(m, name, &f, &log)) { fprintf(stderr, "%s\n", log.c_str()); exit(1); }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int regs = 0, lds = 0, scratch = 0;
    hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, f);
    hipFuncGetAttribute(&lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, f);
    hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f);
    printf("rtc %s: %.2f s  regs %d  lds %d  scratch %d\n", name, dt, regs, lds, scratch);
    return f;
}
template <class... A> static void launch(hipFunction_t f, dim3 grid, dim3 block, A... a) {
    void* args[] = {(void*)&a...};
    CK(hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, 0, args, nullptr));
}

static std::vector<cd> dft(const std::vector<cd>& x, int sign) {
    const int n = (int)x.size();
    std::vector<cd> w(n), y(n);
    for (int i = 0; i < n; ++i) w[i] = std::polar(1.0, sign * 2.0 * M_PI * i / n);
    for (int k = 0; k < n; ++k) {
        cd acc = 0;
        long long idx = 0;
        for (int i = 0; i < n; ++i) {
            acc += x[i] * w[idx];
            idx += k;
            if (idx >= n) idx -= n;
        }
        y[k] = acc;
    }
    return y;
}

// the old engine's stage list: fewest stages with radices <= 12, large first
static bool old_factor(int n, GenGeo* g) {
    std::vector<int> best;
    static const int rad[] = {12, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    std::map<int, std::vector<int>> b;
    b[1] = {};
    for (int d = 1; d <= n; ++d) {
        if (n % d || !b.count(d)) continue;
        for (int r : rad) {
            long long e = (long long)d * r;
            if (e > n || n % e) continue;
            auto c = b[d];
            c.push_back(r);
            std::sort(c.begin(), c.end(), std::greater<int>());
            if (!b.count((int)e) || c.size() < b[(int)e].size() ||
                (c.size() == b[(int)e].size() && c[0] < b[(int)e][0]))
                b[(int)e] = c;
        }
    }
    if (!b.count(n)) return false;
    *g = GenGeo{};
    g->n = n;
    for (int r : b[n]) g->fac[g->nfac++] = r;
    return true;
}
static std::vector<cf> old_tables(GenGeo* g) {
    int ns = 1, total = 0;
    for (int s = 0; s < g->nfac; ++s) {
        g->woff[s] = total;
        if (s > 0) total += (g->fac[s] - 1) * ns;
        ns *= g->fac[s];
    }
    std::vector<cf> h((size_t)std::max(total, 1));
    ns = 1;
    for (int s = 0; s < g->nfac; ++s) {
        if (s > 0)
            for (int r = 1; r < g->fac[s]; ++r)
                for (int k = 0; k < ns; ++k)
                    h[(size_t)g->woff[s] + (size_t)(r - 1) * ns + k] = croot((long long)r * k, (long long)ns * g->fac[s]);
        ns *= g->fac[s];
    }
    return h;
}
static int old_threads(int elements) {
    int t = ((elements + BBT_GEN_EPT - 1) / BBT_GEN_EPT + 63) / 64 * 64;
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}
static void print_geo(const char* what, const G2Plan& g) {
    printf("%s n %d =", what, g.n);
    for (int s = 0; s < g.nfac; ++s) printf(" %d", g.fac[s]);
    printf("  threads/transform %d  ct %d  block %d  slots %d  pitches", g.tj, g.ct, g.threads(), g.slots);
    for (int s = 0; s + 1 < g.nfac; ++s) printf(" %d", g.pitch[s]);
    printf("  lds %.1f KiB\n", g.lds_elems * 8 / 1024.0);
}

template <class F> static double time_us(F&& launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3 / reps;
}

static double rel_err(const std::vector<cd>& want, const float* got4, int stream) {
    // got4: internal format re_A re_B im_A im_B per element
    double num = 0, den = 0;
    for (size_t i = 0; i < want.size(); ++i) {
        cd g(got4[4 * i + stream], got4[4 * i + 2 + stream]);
        num += std::norm(g - want[i]);
        den += std::norm(want[i]);
    }
    return std::sqrt(num / den);
}

static int run_row(int N1, int N2, int blocks, int reps, bool with_old) {
    const long long N = (long long)N1 * N2;
    G2Plan g;
    if (!g2_plan(N2, 1, &g, g2_pmax(BBT_G2_KIND_ROW))) { printf("cannot factor %d\n", N2); return 1; }
    const G2Plan gr = g2_reversed(g);
    print_geo("row fwd", g);
    print_geo("row inv", gr);
    cf* wn = upload_tables(g);
    cf* wnr = upload_tables(gr);
    const std::string src = "#include \"gen2_kernels.hpp\"\n" + g2_trait_source("GA", g) + g2_trait_source("GB", gr) +
                            std::string("BBT_G2_KERNEL_ROW(k_row, GA, GB, ") + (g.threads() >= 448 ? "4" : "0") + ")\n";
    hipFunction_t k_row = build(src, "k_row");
    // big twiddle W_N^m = hi[m >> 12] lo[m & 4095]
    std::vector<cf> lo(4096), hi((size_t)((N + 4095) / 4096));
    for (int i = 0; i < 4096; ++i) lo[i] = croot(i, N);
    for (size_t j = 0; j < hi.size(); ++j) hi[j] = croot((long long)j * 4096, N);
    cf *tlo = upload(lo), *thi = upload(hi);
    auto make_tws = [&](int fac0) {
        std::vector<cf> t((size_t)N1 * fac0);
        const int m = N2 / fac0;
        for (int k1 = 0; k1 < N1; ++k1)
            for (int r = 0; r < fac0; ++r) t[(size_t)k1 * fac0 + r] = croot((long long)k1 * m * r, N);
        return t;
    };
    cf* tws = upload(make_tws(g.fac[0]));
    std::mt19937 rng(1);
    std::normal_distribution<float> nd;
    std::vector<float> work((size_t)blocks * N * 4);
    for (auto& x : work) x = nd(rng);
    std::vector<cf> resp((size_t)N);
    for (auto& h : resp) {
        const double a = 2 * M_PI * (rng() / 4294967296.0);
        h.x = (float)(std::cos(a) / N2);
        h.y = (float)(std::sin(a) / N2);
    }
    float* d_work = upload(work);
    float* d_work2 = upload(work);
    cf* d_resp = upload(resp);
    std::vector<int> ridx = {0, 0};
    int* d_ridx = upload(ridx);
    const dim3 grid(N1, blocks), block(g.threads());
    launch(k_row, grid, block, (float2*)d_work, N1, N2, (const cf*)d_resp, (const int*)d_ridx, 1, (const cf*)wn, (const cf*)wnr, (const cf*)tlo, (const cf*)thi, (const cf*)tws);
    CK(hipDeviceSynchronize());
    std::vector<float> got(work.size());
    CK(hipMemcpy(got.data(), d_work, got.size() * 4, hipMemcpyDeviceToHost));
    // check rows k1 in {0, 1, N1 - 1} of block 0 and one row of the last block
    double worst = 0;
    for (auto bk : std::vector<std::pair<int, int>>{{0, 0}, {0, 1}, {0, N1 - 1}, {blocks - 1, N1 / 2}}) {
        const int b = bk.first, k1 = bk.second;
        const float* src = work.data() + ((size_t)b * N1 + k1) * N2 * 4;
        for (int stream = 0; stream < 2; ++stream) {
            std::vector<cd> x(N2);
            for (int i = 0; i < N2; ++i)
                x[i] = cd(src[4 * i + stream], src[4 * i + 2 + stream]) * std::polar(1.0, -2 * M_PI * (double)((long long)k1 * i % N) / N);
            auto X = dft(x, -1);
            for (int i = 0; i < N2; ++i) X[i] *= cd(resp[(size_t)k1 * N2 + i].x, resp[(size_t)k1 * N2 + i].y);
            auto y = dft(X, +1);
            for (int i = 0; i < N2; ++i) y[i] *= std::polar(1.0, +2 * M_PI * (double)((long long)k1 * i % N) / N);
            const double e = rel_err(y, got.data() + ((size_t)b * N1 + k1) * N2 * 4, stream);
            worst = std::max(worst, e);
        }
    }
    printf("g2 row  rel-L2 vs float64: %.3e %s\n", worst, worst < 1e-6 ? "ok" : "FAILED");
    const double t2 = time_us([&] { launch(k_row, grid, block, (float2*)d_work, N1, N2, (const cf*)d_resp, (const int*)d_ridx, 1, (const cf*)wn, (const cf*)wnr, (const cf*)tlo, (const cf*)thi, (const cf*)tws); }, reps);
    printf("g2 row  %8.1f us per launch of %d blocks  (%.2f us per block, %.1f ps per point)\n", t2, blocks, t2 / blocks, t2 * 1e6 / (blocks * (double)N));
    if (with_old) {
        GenGeo og, ogr;
        if (!old_factor(N2, &og)) return 1;
        cf* own = upload(old_tables(&og));
        ogr = og;
        for (int s = 0; s < og.nfac; ++s) ogr.fac[s] = og.fac[og.nfac - 1 - s];
        cf* ownr = upload(old_tables(&ogr));
        cf* otws = upload(make_tws(og.fac[0]));
        const size_t olds = (size_t)N2 * 16;
        CK(hipFuncSetAttribute((const void*)k_gen_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)olds));
        const dim3 oblock(old_threads(N2));
        hipLaunchKernelGGL(k_gen_row, grid, oblock, olds, 0, (float2*)d_work2, N1, d_resp, d_ridx, 1, og, own, ogr, ownr, tlo, thi, otws);
        CK(hipDeviceSynchronize());
        std::vector<float> got2(work.size());
        CK(hipMemcpy(got2.data(), d_work2, got2.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (size_t i = 0; i < got.size(); ++i) { num += (double)(got[i] - got2[i]) * (got[i] - got2[i]); den += (double)got2[i] * got2[i]; }
        printf("old row stages"); for (int s = 0; s < og.nfac; ++s) printf(" %d", og.fac[s]);
        printf("  block %d  lds %.1f KiB   g2 vs old rel-L2 %.3e\n", oblock.x, olds / 1024.0, std::sqrt(num / den));
        const double t1 = time_us([&] { hipLaunchKernelGGL(k_gen_row, grid, oblock, olds, 0, (float2*)d_work2, N1, d_resp, d_ridx, 1, og, own, ogr, ownr, tlo, thi, otws); }, reps);
        printf("old row %8.1f us per launch  (%.2f us per block)   speed-up %.2f\n", t1, t1 / blocks, t1 / t2);
    }
    return worst < 1e-6 ? 0 : 1;
}

static int run_chan(int n, int nspec, int reps, bool with_old) {
    G2Plan g;
    if (!g2_plan(n, 1, &g, g2_pmax(BBT_G2_KIND_CHAN))) { printf("cannot factor %d\n", n); return 1; }
    print_geo("chan", g);
    cf* wn = upload_tables(g);
    const std::string src = "#include \"gen2_kernels.hpp\"\n" + g2_trait_source("GA", g) + std::string("BBT_G2_KERNEL_FFT_ROWS(k_chan, GA, -1, ") + (g.threads() >= 448 ? "4" : "0") + ")\n";
    hipFunction_t k_chan = build(src, "k_chan");
    std::mt19937 rng(2);
    std::normal_distribution<float> nd;
    std::vector<float> x((size_t)nspec * n * 4);      // external format: re_A im_A re_B im_B
    for (auto& v : x) v = nd(rng);
    float* d_in = upload(x);
    float* d_out = upload(x);
    const dim3 grid(nspec), block(g.threads());
    launch(k_chan, grid, block, (const float2*)d_in, (float2*)d_out, 2, 1, (long long)nspec, 1.0f, (const cf*)wn);
    CK(hipDeviceSynchronize());
    std::vector<float> got(x.size());
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i : {0, nspec - 1})
        for (int stream = 0; stream < 2; ++stream) {
            std::vector<cd> a(n);
            for (int k = 0; k < n; ++k) a[k] = cd(x[((size_t)i * n + k) * 4 + 2 * stream], x[((size_t)i * n + k) * 4 + 2 * stream + 1]);
            auto A = dft(a, -1);
            double num = 0, den = 0;
            for (int k = 0; k < n; ++k) {
                cd gk(got[((size_t)i * n + k) * 4 + 2 * stream], got[((size_t)i * n + k) * 4 + 2 * stream + 1]);
                num += std::norm(gk - A[k]);
                den += std::norm(A[k]);
            }
            worst = std::max(worst, std::sqrt(num / den));
        }
    printf("g2 chan rel-L2 vs float64: %.3e %s\n", worst, worst < 1e-6 ? "ok" : "FAILED");
    const double t2 = time_us([&] { launch(k_chan, grid, block, (const float2*)d_in, (float2*)d_out, 2, 1, (long long)nspec, 1.0f, (const cf*)wn); }, reps);
    printf("g2 chan %8.1f us  %.1f Gsamples/s\n", t2, (double)nspec * n / t2 * 1e-3);
    if (with_old) {
        GenGeo og;
        if (!old_factor(n, &og)) return 1;
        cf* own = upload(old_tables(&og));
        const size_t olds = (size_t)n * 16;
        CK(hipFuncSetAttribute((const void*)k_gen_fft_rows<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)olds));
        const dim3 oblock(old_threads(n));
        const double t1 = time_us([&] { hipLaunchKernelGGL((k_gen_fft_rows<-1>), grid, oblock, olds, 0, (const float2*)d_in, (float2*)d_out, 2, 1, 1.0f, og, own); }, reps);
        printf("old chan %7.1f us  %.1f Gsamples/s   speed-up %.2f\n", t1, (double)nspec * n / t1 * 1e-3, t1 / t2);
    }
    return worst < 1e-6 ? 0 : 1;
}

static int run_col(int N1, int N2, int blocks, int reps, int ct, bool with_old) {
    const long long N = (long long)N1 * N2;
    G2Plan g;
    if (!g2_plan(N1, ct, &g, g2_pmax(BBT_G2_KIND_COL))) { printf("cannot factor %d\n", N1); return 1; }
    print_geo("col", g);
    cf* wn = upload_tables(g);
    const std::string src = "#include \"gen2_kernels.hpp\"\n" + g2_trait_source("GA", g) +
                            std::string("BBT_G2_KERNEL_COL(k_first, GA, true, 0)\nBBT_G2_KERNEL_COL(k_last, GA, false, 0)\n");
    hipFunction_t k_first = build(src, "k_first"), k_last = build(src, "k_last");
    std::mt19937 rng(3);
    std::normal_distribution<float> nd;
    std::vector<float> x((size_t)(blocks * N) * 4);
    for (auto& v : x) v = nd(rng);
    float* d_in = upload(x);
    float* d_work = upload(x);
    float* d_out = upload(x);
    OsmChunk ch = {};
    ch.nblk = blocks;
    ch.reg_count = blocks;
    ch.reg_hop = N;
    ch.b[0].in_off = 0;
    ch.b[0].out_off = 0;
    ch.b[0].valid_start = 0;
    ch.b[0].valid_count = (int)N;
    const int tiles = (N2 + ct - 1) / ct;
    const dim3 grid(tiles * blocks), block(g.threads());
    launch(k_first, grid, block, (const float2*)d_in, (float2*)d_out, (float2*)d_work, ch, 2, N2, N2, (const cf*)wn);
    CK(hipDeviceSynchronize());
    std::vector<float> got(x.size());
    CK(hipMemcpy(got.data(), d_work, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (auto bc : std::vector<std::pair<int, int>>{{0, 0}, {0, 1}, {blocks - 1, N2 - 1}}) {
        const int b = bc.first, n2 = bc.second;
        for (int stream = 0; stream < 2; ++stream) {
            std::vector<cd> a(N1);
            for (int n1 = 0; n1 < N1; ++n1) {
                const size_t e = ((size_t)b * N + (size_t)n1 * N2 + n2) * 4;
                a[n1] = cd(x[e + 2 * stream], x[e + 2 * stream + 1]);
            }
            auto A = dft(a, -1);
            double num = 0, den = 0;
            for (int k1 = 0; k1 < N1; ++k1) {
                const size_t e = ((size_t)b * N + (size_t)k1 * N2 + n2) * 4;
                cd gk(got[e + stream], got[e + 2 + stream]);
                num += std::norm(gk - A[k1]);
                den += std::norm(A[k1]);
            }
            worst = std::max(worst, std::sqrt(num / den));
        }
    }
    printf("g2 col<first> rel-L2 vs float64: %.3e %s\n", worst, worst < 1e-6 ? "ok" : "FAILED");
    // last pass: inverse over k1 of the work buffer -> stream order; check one column
    launch(k_last, grid, block, (const float2*)d_in, (float2*)d_out, (float2*)d_work, ch, 2, N2, N2, (const cf*)wn);
    CK(hipDeviceSynchronize());
    std::vector<float> back(x.size());
    CK(hipMemcpy(back.data(), d_out, back.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < x.size(); ++i) { const double d = back[i] / (double)N1 - x[i]; num += d * d; den += (double)x[i] * x[i]; }
    const double rt = std::sqrt(num / den);
    printf("g2 col first + last round trip rel-L2: %.3e %s\n", rt, rt < 1e-6 ? "ok" : "FAILED");
    const double tf = time_us([&] { launch(k_first, grid, block, (const float2*)d_in, (float2*)d_out, (float2*)d_work, ch, 2, N2, N2, (const cf*)wn); }, reps);
    const double tl = time_us([&] { launch(k_last, grid, block, (const float2*)d_in, (float2*)d_out, (float2*)d_work, ch, 2, N2, N2, (const cf*)wn); }, reps);
    printf("g2 col first %8.1f us  last %8.1f us per launch of %d blocks\n", tf, tl, blocks);
    if (with_old) {
        GenGeo og;
        if (!old_factor(N1, &og)) return 1;
        cf* own = upload(old_tables(&og));
        int oct = 1;
        while (oct < 8 && 2 * oct * N1 <= BBT_GEN_MAX_LEN) oct *= 2;
        const size_t olds = (size_t)N1 * oct * 16;
        CK(hipFuncSetAttribute((const void*)k_gen_col<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)olds));
        CK(hipFuncSetAttribute((const void*)k_gen_col<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)olds));
        const dim3 ogrid((N2 + oct - 1) / oct, blocks), oblock(old_threads(N1 * oct));
        const double of = time_us([&] { hipLaunchKernelGGL((k_gen_col<true>), ogrid, oblock, olds, 0, (const float2*)d_in, (float2*)d_out, (float2*)d_work, ch, 2, N2, oct, og, own); }, reps);
        const double ol = time_us([&] { hipLaunchKernelGGL((k_gen_col<false>), ogrid, oblock, olds, 0, (const float2*)d_in, (float2*)d_out, (float2*)d_work, ch, 2, N2, oct, og, own); }, reps);
        printf("old col first %7.1f us  last %8.1f us (ct %d, block %d)   speed-up %.2f / %.2f\n", of, ol, oct, oblock.x, of / tf, ol / tl);
    }
    return (worst < 1e-6 && rt < 1e-6) ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: see the top of tools/gen2_bench.hip\n"); return 2; }
    const std::string mode = argv[1];
    const bool with_old = !getenv("G2_NO_OLD");
    if (mode == "row") return run_row(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 3, argc > 5 ? atoi(argv[5]) : 20, with_old);
    if (mode == "chan") return run_chan(atoi(argv[2]), argc > 3 ? atoi(argv[3]) : 4096, argc > 4 ? atoi(argv[4]) : 20, with_old);
    if (mode == "col") return run_col(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 3, argc > 5 ? atoi(argv[5]) : 20, argc > 6 ? atoi(argv[6]) : 8, with_old);
    return 2;
}
