// libbbt_hip.so -- host side of the C ABI declared in include/bbt_hip.h.
// Plans, twiddle tables, launch geometry; the kernels are in bbt_kernels.hpp.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/bbt_hip.h"
#include "bbt_kernels.hpp"
#include "gen_kernels.hpp"
#include "big_kernels.hpp"
#include "gen2_host.hpp"
#include "rtc.hpp"

using namespace bbt;

#define BBT_VERSION 152

// ---------------------------------------------------------------------------
// errors
static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                      \
    } while (0)

#define ARG_TRY(cond, ...) \
    do {                   \
        if (!(cond)) return fail(__VA_ARGS__); \
    } while (0)

static bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// ---------------------------------------------------------------------------
// twiddle tables, per device
struct FftTables {
    cf* tw0 = nullptr;  // [16][T]   W_N^{tau c0}
    cf* tw1 = nullptr;  // [16][R2]  W_T^{b1 c1}
};
static std::mutex g_tab_mutex;
static std::map<std::pair<int, int>, FftTables> g_tables;  // (device, N)
static std::map<int, cf*> g_wroot;                         // device -> W_4096^m, then W_65536^i (i < 256)

static int upload(cf** dst, const std::vector<cf>& h) {
    HIP_TRY(hipMalloc((void**)dst, h.size() * sizeof(cf)));
    HIP_TRY(hipMemcpy(*dst, h.data(), h.size() * sizeof(cf), hipMemcpyHostToDevice));
    return 0;
}

static cf unit_root(long long k, long long n) {
    // exp(-2 pi i k / n), evaluated in double with exact octant reduction
    k %= n;
    const double a = -2.0 * M_PI * (double)k / (double)n;
    return make_float2((float)cos(a), (float)sin(a));
}

static int get_tables(int n, FftTables* out) {
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    auto it = g_tables.find({dev, n});
    if (it != g_tables.end()) {
        *out = it->second;
        return 0;
    }
    const int r2 = n / 256, t = n / 16;
    std::vector<cf> h0(16 * t), h1(16 * r2);
    for (int c = 0; c < 16; ++c)
        for (int tau = 0; tau < t; ++tau) h0[c * t + tau] = unit_root((long long)tau * c, n);
    for (int c = 0; c < 16; ++c)
        for (int b = 0; b < r2; ++b) h1[c * r2 + b] = unit_root((long long)b * c, t);
    FftTables ft;
    if (upload(&ft.tw0, h0)) return 1;
    if (upload(&ft.tw1, h1)) return 1;
    g_tables[{dev, n}] = ft;
    *out = ft;
    return 0;
}

// twiddles of the 8192- / 16384-point transforms (fft_big.hpp: BigGeo<N>::TW_*), cached per device
static std::map<std::pair<int, int>, cf*> g_big_tables;
static int get_big_table(int n, cf** out) {
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    auto it = g_big_tables.find({dev, n});
    if (it != g_big_tables.end()) {
        *out = it->second;
        return 0;
    }
    const int t = n / 16, m = t / 16, l = m / 16;
    std::vector<cf> h;
    h.reserve(4 * (size_t)(t + m + l));
    for (int c = 1; c <= 8; c *= 2)
        for (int i = 0; i < t; ++i) h.push_back(unit_root((long long)i * c, n));
    for (int c = 1; c <= 8; c *= 2)
        for (int i = 0; i < m; ++i) h.push_back(unit_root((long long)i * c, t));
    for (int c = 1; c <= 8; c *= 2)
        for (int i = 0; i < l; ++i) h.push_back(unit_root((long long)i * c, m));
    cf* d;
    if (upload(&d, h)) return 1;
    g_big_tables[{dev, n}] = d;
    *out = d;
    return 0;
}

static int get_wroot(cf** out) {
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    auto it = g_wroot.find(dev);
    if (it != g_wroot.end()) {
        *out = it->second;
        return 0;
    }
    std::vector<cf> h(4096 + 256);
    for (int m = 0; m < 4096; ++m) h[m] = unit_root(m, 4096);
    for (int m = 0; m < 256; ++m) h[4096 + m] = unit_root(m, 65536);
    cf* d;
    if (upload(&d, h)) return 1;
    g_wroot[dev] = d;
    *out = d;
    return 0;
}

static bool fft_len_ok(int64_t n) { return is_pow2(n) && n >= 256 && n <= 4096; }

// ---- lengths 2^a 3^b 5^c 7^d (fft_generic.hpp) ------------------------------
static bool factor_7smooth(int64_t n, GenGeo* g) {
    // Stages of the LDS Stockham transform: radices from {2..10, 12, 14, 15, 16} (the composite
    // ones are small Cooley-Tukey transforms on registers, fft_generic.hpp), as few as possible --
    // every stage is a round trip of the whole tile through LDS with two barriers -- and among
    // the shortest lists the one with the smallest largest radix (registers).
    if (n < 1 || n > BBT_GEN_MAX_LEN) return false;
    static const int all[] = {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    const int* radices = all;
    int nrad = 13;
    while (nrad > 1 && radices[0] > BBT_GEN_MAXR) {   // (the list is in descending order)
        ++radices;
        --nrad;
    }
    g->n = (int)n;
    g->nfac = 0;
    if (n == 1) return true;
    // dynamic programme over the divisors of n: best[d] = (stages, largest radix) to reach d
    std::map<int64_t, std::pair<int, int>> best;
    std::map<int64_t, int> step;
    best[1] = {0, 0};
    std::vector<int64_t> divisors;
    for (int64_t d = 1; d <= n; ++d)
        if (n % d == 0) divisors.push_back(d);
    for (int64_t d : divisors) {
        auto it = best.find(d);
        if (it == best.end()) continue;
        for (int i = 0; i < nrad; ++i) {
            const int r = radices[i];
            const int64_t e = d * r;
            if (n % e) continue;
            const std::pair<int, int> cand = {it->second.first + 1, std::max(it->second.second, r)};
            auto jt = best.find(e);
            if (jt == best.end() || cand < jt->second) {
                best[e] = cand;
                step[e] = r;
            }
        }
    }
    if (!best.count(n) || best[n].first > BBT_GEN_MAX_FACTORS) return false;
    std::vector<int> fac;
    for (int64_t d = n; d > 1; d /= step[d]) fac.push_back(step[d]);
    std::sort(fac.begin(), fac.end(), std::greater<int>());      // (large radices first: fewer twiddles)
    for (int r : fac) g->fac[g->nfac++] = r;
    return true;
}
static int rtc_mode() {                     // 0 off, 1 on, 2 required
    const char* e = getenv("BBT_RTC");
    if (!e || !*e) return 1;
    if (!strcmp(e, "0")) return 0;
    if (!strcmp(e, "require")) return 2;
    return 1;
}
static bool is_7smooth(int64_t n) {
    if (n < 1) return false;
    for (int r : {2, 3, 5, 7})
        while (n % r == 0) n /= r;
    return n == 1;
}
// N = N1 * N2 with N1 <= N2 <= BBT_GEN_MAX_LEN: the largest N1 up to 512 (the column passes
// then hold 8 columns of N1 points in their LDS tile: 128-byte runs), else as balanced as possible.
static bool split_7smooth(int64_t n, int* n1, int* n2) {
    if (const char* env = getenv("BBT_GEN_N1")) {            // (dev: force the split)
        const int64_t d = atoll(env);
        if (d > 1 && n % d == 0 && n / d <= BBT_GEN_MAX_LEN && d <= BBT_GEN_MAX_LEN) {
            *n1 = (int)d;
            *n2 = (int)(n / d);
            return true;
        }
    }
    // (plans on the run-time specialised kernels: the split rule measured for them; the column
    // tile of the general kernels holds n1 * 8 <= 8192 elements, so n1 <= 1024 keeps the fall-back)
    if (rtc_mode() && g2_choose_split(n, 8, 1024, BBT_GEN_MAX_LEN, n1, n2)) return true;
    const int64_t prefer = 512;
    int64_t best = 0, wide = 0;
    for (int64_t d = 1; d * d <= n; ++d)
        if (n % d == 0 && n / d <= BBT_GEN_MAX_LEN) {
            best = d;
            if (d <= prefer) wide = d;
        }
    if (wide) best = wide;
    if (!best) return false;
    *n1 = (int)best;
    *n2 = (int)(n / best);
    return true;
}
static int gen_threads(int elements) {          // elements <= BBT_GEN_EPT * threads, whole waves
    int t = ((elements + BBT_GEN_EPT - 1) / BBT_GEN_EPT + 63) / 64 * 64;
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}
static std::map<std::pair<int, int>, cf*> g_gen_tables;    // (device, n) -> stage twiddles; -n: stages reversed
static int get_gen_table(GenGeo* g, cf** out, bool reversed = false);
// The same stages in reversed order: what the inverse of a convolution runs (fft_generic.hpp,
// gen_conv_open), with its own stage tables.
static int get_reversed(const GenGeo& g, GenGeo* gr, cf** out) {
    *gr = g;
    for (int s = 0; s < g.nfac; ++s) gr->fac[s] = g.fac[g.nfac - 1 - s];
    return get_gen_table(gr, out, true);
}
static int make_big_twiddle(int64_t n, cf** lo, cf** hi);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a
// kernel: set it once per (device, kernel) and check the result.
static int ensure_dyn_lds(const void* func, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({dev, func});
    if (it != done.end() && it->second >= bytes) return 0;
    HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[{dev, func}] = bytes;
    return 0;
}

// Stage twiddles of the LDS Stockham transform of g->n points, stage after stage: stage s (radix
// R, Ns = product of the earlier radices) has its W_{Ns R}^{r k} at [g->woff[s] + (r - 1) Ns + k],
// r = 1 .. R - 1, k < Ns -- contiguous in k, which is what neighbouring lanes differ in, so a
// stage's twiddle loads are coalesced (gathered from one W_n^k table they touched up to 64 cache
// lines per wave instruction).  The first stage (Ns = 1) has none.  Fills g->woff.
static int get_gen_table(GenGeo* g, cf** out, bool reversed) {
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    const int n = g->n;
    int ns = 1, total = 0;
    for (int s = 0; s < g->nfac; ++s) {
        g->woff[s] = total;
        if (s > 0) total += (g->fac[s] - 1) * ns;
        ns *= g->fac[s];
    }
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    const int key = reversed ? -n : n;
    auto it = g_gen_tables.find({dev, key});
    if (it != g_gen_tables.end()) {
        *out = it->second;
        return 0;
    }
    std::vector<cf> h((size_t)std::max(total, 1));
    ns = 1;
    for (int s = 0; s < g->nfac; ++s) {
        const int r_s = g->fac[s];
        if (s > 0)
            for (int r = 1; r < r_s; ++r)
                for (int k = 0; k < ns; ++k)
                    h[(size_t)g->woff[s] + (size_t)(r - 1) * ns + k] = unit_root((long long)r * k, (long long)ns * r_s);
        ns *= r_s;
    }
    cf* d;
    if (upload(&d, h)) return 1;
    g_gen_tables[{dev, key}] = d;
    *out = d;
    return 0;
}
// W_N^m = hi[m >> 12] * lo[m & 4095]  (per plan: depends on N)
static int make_big_twiddle(int64_t n, cf** lo, cf** hi) {
    std::vector<cf> l(4096), h((size_t)((n + 4095) / 4096));
    for (int i = 0; i < 4096; ++i) l[i] = unit_root(i, n);
    for (size_t j = 0; j < h.size(); ++j) h[j] = unit_root((long long)j * 4096, n);
    if (upload(lo, l) || upload(hi, h)) return 1;
    return 0;
}

// ---- the register-resident engine for those lengths, specialised at plan time ---------------
// (fft_gen2.hpp, gen2_kernels.hpp; compiled by rtc.hpp).  BBT_RTC=0: the LDS Stockham engine
// above runs every such length (as in rounds 2-4); BBT_RTC=require: a failing compilation is an
// error instead of a warning and a fall back to it.
static std::mutex g_g2_mutex;
static std::map<std::pair<int, std::string>, cf*> g_g2_tables;    // (device, stage list) -> tables
static int64_t g_rtc_modules = 0;
static double g_rtc_seconds = 0;
static int get_g2_table(const G2Plan& g, cf** out) {
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    std::string key = std::to_string(g.n);
    for (int s = 0; s < g.nfac; ++s) key += "," + std::to_string(g.fac[s]);
    std::lock_guard<std::mutex> lock(g_g2_mutex);
    auto it = g_g2_tables.find({dev, key});
    if (it != g_g2_tables.end()) {
        *out = it->second;
        return 0;
    }
    const std::vector<float> t = g2_tables(g);
    cf* d;
    HIP_TRY(hipMalloc((void**)&d, t.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    g_g2_tables[{dev, key}] = d;
    *out = d;
    return 0;
}
// source -> entry points.  0 = ok; 1 = not available (the caller falls back), with the reason in
// g_err; in `require` mode that is the error of the call.
static int g2_build(const std::string& source, const std::vector<const char*>& names, hipFunction_t* fns) {
    RtcModule* m;
    std::string log;
    const auto t0 = std::chrono::steady_clock::now();
    if (rtc_module(source, &m, &log)) return fail("run-time compilation of a generic-length kernel failed: %s", log.c_str());
    for (size_t i = 0; i < names.size(); ++i)
        if (rtc_function(m, names[i], &fns[i], &log)) return fail("%s", log.c_str());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
        std::lock_guard<std::mutex> lock(g_g2_mutex);
        if (dt > 0.02) {                    // (a module that was compiled, not found in the process's cache)
            g_rtc_modules += 1;
            g_rtc_seconds += dt;
        }
    }
    if (getenv("BBT_RTC_VERBOSE")) fprintf(stderr, "bbt: rtc %.2f s\n%s", dt, source.c_str());
    return 0;
}
static void g2_warn_once(const char* what) {
    static std::once_flag once;
    std::call_once(once, [&] {
        fprintf(stderr, "baseband_tasks_amd: %s -- %s; lengths that are not powers of two run on the slower "
                        "general kernels (BBT_RTC=require makes this an error)\n", what, g_err.c_str());
    });
}
template <class... A>
static int g2_launch(hipFunction_t f, dim3 grid, dim3 block, hipStream_t st, A... a) {
    void* args[] = {(void*)&a...};
    HIP_TRY(hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, st, args, nullptr));
    return 0;
}

struct DevicePool {
    std::mutex mu;
    struct Idle {
        void* ptr;
        hipStream_t freed_on;   // pool stream when the block was freed
    };
    std::map<int, std::multimap<size_t, Idle>> free_blocks;         // device -> size -> block
    std::map<void*, std::pair<size_t, int>> live;                   // block -> (size, device)
    size_t cached = 0;
    // The stream on which blocks of the pool are used (bbt_pool_set_stream).
    // Reuse of a freed block is ordered by that stream; a block freed under
    // another pool stream is handed out only after the device has drained.
    hipStream_t stream = nullptr;
    static size_t limit() {                      // cached (idle) bytes kept at most; BBT_POOL_MAX_GB
        static const size_t lim = [] {
            const char* e = getenv("BBT_POOL_MAX_GB");
            return (size_t)(e ? atof(e) : 96.0) << 30;
        }();
        return lim;
    }
    static bool enabled() {
        static const bool on = [] { const char* e = getenv("BBT_POOL"); return !(e && atoi(e) == 0); }();
        return on;
    }
};
static DevicePool g_pool;
extern "C" int bbt_pool_trim(void);

// ---------------------------------------------------------------------------
extern "C" {

const char* bbt_last_error(void) { return g_err.c_str(); }
int bbt_version(void) { return BBT_VERSION; }
int bbt_rtc_info(int* mode, int64_t* modules, double* seconds) {
    std::lock_guard<std::mutex> lock(g_g2_mutex);
    if (mode) *mode = rtc_mode();
    if (modules) *modules = g_rtc_modules;
    if (seconds) *seconds = g_rtc_seconds;
    return 0;
}

int bbt_device_count(int* count) {
    ARG_TRY(count, "bbt_device_count: null argument");
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}
int bbt_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return 0;
}
int bbt_get_device(int* device) {
    ARG_TRY(device, "bbt_get_device: null argument");
    HIP_TRY(hipGetDevice(device));
    return 0;
}
int bbt_device_name(char* buf, int buflen) {
    ARG_TRY(buf && buflen > 0, "bbt_device_name: bad buffer");
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return 0;
}

// Device memory comes from a caching pool: hipMalloc / hipFree of the GB-sized
// stream buffers a reader allocates per call cost milliseconds and hipFree
// synchronises the device, which would serialise consecutive calls.  Freed
// blocks are kept (per device, by size) and handed out again for requests of
// up to 1/8 less; reuse is ordered by the caller's stream, as every consumer of
// a block is enqueued on it before the block is freed (plans join their
// internal streams before returning).  BBT_POOL=0 disables caching.
int bbt_malloc(void** dev_ptr, size_t nbytes) {
    ARG_TRY(dev_ptr, "bbt_malloc: null argument");
    if (!nbytes) nbytes = 1;
    const size_t gran = nbytes >= (1u << 20) ? (size_t)2 << 20 : 512;
    const size_t want = (nbytes + gran - 1) / gran * gran;
    int dev;
    HIP_TRY(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        auto& free_blocks = g_pool.free_blocks[dev];
        auto it = free_blocks.lower_bound(want);
        if (it != free_blocks.end() && it->first - want <= want / 8) {
            const DevicePool::Idle idle = it->second;
            *dev_ptr = idle.ptr;
            g_pool.cached -= it->first;
            g_pool.live[idle.ptr] = {it->first, dev};
            free_blocks.erase(it);
            if (idle.freed_on != g_pool.stream) {
                // freed under another stream: its last users are not ordered
                // before work on the current pool stream -- drain the device
                HIP_TRY(hipDeviceSynchronize());
            }
            return 0;
        }
    }
    hipError_t e = hipMalloc(dev_ptr, want);
    if (e != hipSuccess) {                      // give cached blocks back and retry once
        (void)hipGetLastError();
        if (bbt_pool_trim()) return 1;
        e = hipMalloc(dev_ptr, want);
    }
    if (e != hipSuccess) {
        *dev_ptr = nullptr;
        return fail("bbt_malloc: hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lock(g_pool.mu);
    g_pool.live[*dev_ptr] = {want, dev};
    return 0;
}
int bbt_free(void* dev_ptr) {
    if (!dev_ptr) return 0;
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        auto it = g_pool.live.find(dev_ptr);
        if (it == g_pool.live.end()) return fail("bbt_free: pointer was not allocated by bbt_malloc");
        const size_t size = it->second.first;
        const int dev = it->second.second;
        g_pool.live.erase(it);
        if (g_pool.enabled() && g_pool.cached + size <= g_pool.limit()) {
            g_pool.free_blocks[dev].emplace(size, DevicePool::Idle{dev_ptr, g_pool.stream});
            g_pool.cached += size;
            return 0;
        }
    }
    HIP_TRY(hipFree(dev_ptr));
    return 0;
}
int bbt_pool_trim(void) {
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        for (auto& per_dev : g_pool.free_blocks) {
            for (auto& b : per_dev.second) blocks.push_back(b.second.ptr);
            per_dev.second.clear();
        }
        g_pool.cached = 0;
    }
    for (void* b : blocks) HIP_TRY(hipFree(b));
    return 0;
}
int bbt_pool_set_stream(bbt_stream stream) {
    // A change of the pool stream drains the device once: blocks are tagged with the pool
    // stream at the time they are freed, and a block whose last kernels were queued on the
    // previous stream may be freed (by a garbage collector, at any time) only after the switch
    // -- it would then carry the new stream's tag and be handed out unordered.  After the
    // drain every idle and every live block is safe under the new stream.
    bool changed;
    {
        std::lock_guard<std::mutex> lock(g_pool.mu);
        changed = g_pool.stream != (hipStream_t)stream;
    }
    if (changed) HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_pool.mu);
    g_pool.stream = (hipStream_t)stream;
    if (changed)
        for (auto& per_dev : g_pool.free_blocks)
            for (auto& b : per_dev.second) b.second.freed_on = g_pool.stream;
    return 0;
}
int bbt_pool_info(int64_t* cached_bytes, int64_t* live_bytes) {
    std::lock_guard<std::mutex> lock(g_pool.mu);
    if (cached_bytes) *cached_bytes = (int64_t)g_pool.cached;
    if (live_bytes) {
        size_t n = 0;
        for (auto& l : g_pool.live) n += l.second.first;
        *live_bytes = (int64_t)n;
    }
    return 0;
}
int bbt_host_alloc(void** host_ptr, size_t nbytes) {
    ARG_TRY(host_ptr, "bbt_host_alloc: null argument");
    HIP_TRY(hipHostMalloc(host_ptr, nbytes ? nbytes : 1, hipHostMallocDefault));
    return 0;
}
int bbt_host_free(void* host_ptr) {
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return 0;
}
int bbt_host_register(void* host_ptr, size_t nbytes) {
    ARG_TRY(host_ptr && nbytes, "bbt_host_register: null or empty range");
    HIP_TRY(hipHostRegister(host_ptr, nbytes, hipHostRegisterDefault));
    return 0;
}
int bbt_host_unregister(void* host_ptr) {
    if (host_ptr) HIP_TRY(hipHostUnregister(host_ptr));
    return 0;
}
int bbt_memset(void* dev_ptr, int value, size_t nbytes, bbt_stream stream) {
    HIP_TRY(hipMemsetAsync(dev_ptr, value, nbytes, (hipStream_t)stream));
    return 0;
}
int bbt_memcpy_h2d(void* dst, const void* src, size_t n, bbt_stream s) {
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
int bbt_memcpy_d2h(void* dst, const void* src, size_t n, bbt_stream s) {
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)s));
    return 0;
}
int bbt_memcpy_d2d(void* dst, const void* src, size_t n, bbt_stream s) {
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
}
int bbt_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                 size_t height, int kind, bbt_stream s) {
    ARG_TRY(kind >= 0 && kind <= 2, "bbt_memcpy2d: kind must be 0 (h2d), 1 (d2h) or 2 (d2d)");
    const hipMemcpyKind k = kind == 0   ? hipMemcpyHostToDevice
                            : kind == 1 ? hipMemcpyDeviceToHost
                                        : hipMemcpyDeviceToDevice;
    if (width == 0 || height == 0) return 0;
    if (kind == 2) {
        // on the device: our own copy kernel when everything is a multiple of 4 bytes
        const uintptr_t all = (uintptr_t)dst | (uintptr_t)src | dpitch | spitch | width;
        const int eb = all % 16 == 0 ? 16 : (all % 8 == 0 ? 8 : (all % 4 == 0 ? 4 : 0));
        const long long wpr = eb ? (long long)(width / eb) : 0;
        int lg_le = 0;
        while ((1ll << lg_le) < wpr && lg_le < 8) ++lg_le;
        const long long rows_per_block = (256 >> lg_le) * 4;
        const long long gx = ((long long)height + rows_per_block - 1) / rows_per_block;
        const long long gy = (wpr + (1 << lg_le) - 1) >> lg_le;
        if (eb && wpr < (1ll << 31) && gx < (1ll << 31) && gy <= 65535) {
            const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
            hipStream_t st = (hipStream_t)s;
            if (eb == 16)
                hipLaunchKernelGGL((k_copy2d<float4>), grid, block, 0, st, (const float4*)src, (float4*)dst,
                                   (long long)height, (int)wpr, (int)wpr, (long long)(spitch / 16),
                                   (long long)(dpitch / 16), lg_le);
            else if (eb == 8)
                hipLaunchKernelGGL((k_copy2d<float2>), grid, block, 0, st, (const float2*)src, (float2*)dst,
                                   (long long)height, (int)wpr, (int)wpr, (long long)(spitch / 8),
                                   (long long)(dpitch / 8), lg_le);
            else
                hipLaunchKernelGGL((k_copy2d<float>), grid, block, 0, st, (const float*)src, (float*)dst,
                                   (long long)height, (int)wpr, (int)wpr, (long long)(spitch / 4),
                                   (long long)(dpitch / 4), lg_le);
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, k, (hipStream_t)s));
    return 0;
}
int bbt_pad_streams(const void* in_dev, void* out_dev, int64_t n_rows, int n_in, int n_out,
                    int elem_bytes, bbt_stream s) {
    ARG_TRY(in_dev && out_dev, "bbt_pad_streams: null argument");
    ARG_TRY(n_rows >= 0 && n_in >= 1 && n_out >= n_in, "bbt_pad_streams: bad sizes");
    ARG_TRY(elem_bytes == 4 || elem_bytes == 8, "bbt_pad_streams: elem_bytes must be 4 or 8");
    if (n_rows == 0) return 0;
    int lg_le = 0;
    while ((1 << lg_le) < n_out && lg_le < 8) ++lg_le;
    const long long rows_per_block = (256 >> lg_le) * 4;
    const long long gx = (n_rows + rows_per_block - 1) / rows_per_block;
    const long long gy = ((long long)n_out + (1 << lg_le) - 1) >> lg_le;
    ARG_TRY(gx < (1ll << 31) && gy <= 65535, "bbt_pad_streams: too many elements for one call");
    const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
    if (elem_bytes == 8)
        hipLaunchKernelGGL((k_copy2d<float2>), grid, block, 0, (hipStream_t)s, (const float2*)in_dev,
                           (float2*)out_dev, (long long)n_rows, n_out, n_in, (long long)n_in,
                           (long long)n_out, lg_le);
    else
        hipLaunchKernelGGL((k_copy2d<float>), grid, block, 0, (hipStream_t)s, (const float*)in_dev,
                           (float*)out_dev, (long long)n_rows, n_out, n_in, (long long)n_in,
                           (long long)n_out, lg_le);
    HIP_TRY(hipGetLastError());
    return 0;
}
int bbt_stream_create(bbt_stream* stream) {
    ARG_TRY(stream, "bbt_stream_create: null argument");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (bbt_stream)s;
    return 0;
}
int bbt_stream_destroy(bbt_stream stream) {
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
int bbt_stream_sync(bbt_stream stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int bbt_device_sync(void) {
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}
int bbt_event_create(bbt_event* ev) {
    ARG_TRY(ev, "bbt_event_create: null argument");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = (bbt_event)e;
    return 0;
}
int bbt_event_create_ordering(bbt_event* ev) {
    ARG_TRY(ev, "bbt_event_create_ordering: null argument");
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    *ev = (bbt_event)e;
    return 0;
}
int bbt_event_destroy(bbt_event ev) {
    if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
int bbt_event_record(bbt_event ev, bbt_stream stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
int bbt_event_sync(bbt_event ev) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)ev));
    return 0;
}
int bbt_event_query(bbt_event ev, int* done) {
    ARG_TRY(ev && done, "bbt_event_query: null argument");
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        *done = 0;
        return 0;
    }
    HIP_TRY(e);
    *done = 1;
    return 0;
}
int bbt_stream_wait_event(bbt_stream stream, bbt_event ev) {
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return 0;
}
int bbt_event_elapsed_ms(bbt_event start, bbt_event stop, float* ms) {
    ARG_TRY(ms, "bbt_event_elapsed_ms: null argument");
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// overlap-save spectral multiply
// short-response convolution in the time domain (entry points further down)
struct bbt_fir_plan {
    int n_tap = 0, S = 0, npair = 0;
    int n_chunks = 0, pitch = 0, tap_pitch = 0;
    bool cplx = false;
    float2* tre = nullptr;
    float2* tim = nullptr;
};
static constexpr int BBT_FIR_R = 8;

#define BBT_MAX_LANES 8
// Events that only order streams of this device among each other (fork / join of the lanes, end
// of a call) and the per-pass timing events: no system-scope fence.  A default HIP event writes
// the caches back and invalidates them when it is recorded -- at every call boundary that emptied
// the Infinity Cache under the work buffers and cost both lanes 0.2-0.6 ms (round 4, measured:
// the lanes stood still after the last kernel of a call even with no cross-stream wait queued).
#define BBT_EV_ORDER (hipEventDisableTiming | hipEventDisableSystemFence)
struct bbt_osm_plan {
    // One call at a time per plan: the lanes' work buffers, the seam buffer,
    // the fork/join events and the timing vectors belong to the running call.
    std::mutex mu;
    int device = 0;
    int64_t n = 0;
    int S = 0, npair = 0, C = 0;
    // one stream (S == 1): pairs are made of two consecutive blocks (see SinglePair);
    // npair == 1 and the work buffers hold (chunk + 1) / 2 pairs of blocks
    bool single = false;
    int n1 = 1, n2 = 0;
    int outer = 1;  // 256 for three-level transforms (N > 2^20): N = outer * n1 * n2
    int chunk = 1;
    int cap = 1;                // blocks a lane's work buffer holds (>= chunk): regular runs go in launches of up to that many
    cf* resp = nullptr;        // [C][N1][N2], scaled 1/N
    int* resp_index = nullptr;  // [S]
    float2* work = nullptr;     // [chunk][npair][N1][N2] float4 (lane 0)
    size_t work_bytes = 0;      // per lane
    // Chunks alternate between `lanes` internal streams, each with its own
    // work buffer, so kernels of different chunks overlap: the bandwidth-bound
    // column passes of one chunk run beside the latency-bound row pass of
    // another and fill each other's launch tails.
    int lanes = 1;
    float2* lane_work[BBT_MAX_LANES] = {};
    hipStream_t lane_stream[BBT_MAX_LANES] = {};
    hipEvent_t ev_fork = nullptr, ev_join[BBT_MAX_LANES] = {};
    hipEvent_t ev_done = nullptr;   // end of the previous execute call (on whatever stream it ran)
    bool ev_done_set = false;
    // Deferred join (bbt_osm_plan_defer): a call that was given a completion event does not order
    // the caller's stream after its lanes.  The lanes then run on into the next call -- no drain
    // at the call boundary -- and what has to come after them (the seam pass of the fused
    // channelizer, the completion event) is queued on `tail_stream`.  Chunks keep alternating
    // between the lanes across calls (`lane_cursor`); the seam buffer has two turns so that the
    // seam pass of call i may still read its slots while the lanes of call i + 1 fill the others.
    hipStream_t tail_stream = nullptr;
    hipEvent_t defer_ev = nullptr;          // handed in for the NEXT execute call only
    bool last_forked_deferred = false;      // the previous call left its lanes unjoined
    int lane_cursor = 0;
    hipEvent_t ev_tail[2] = {nullptr, nullptr};   // seam pass of the last deferred call on turn i
    bool ev_tail_set[2] = {false, false};
    int seam_turn = 0;
    FftTables tab2;  // for N2
    FftTables tab1;  // for N1 == 256
    // four-step twiddles of two-level plans as tables (owned): W_{16 N1}^{k1 j} [N1][16] and
    // W_N^{k1 tau} [N1][N2 / 16]
    cf* tw4row = nullptr;
    cf* tw4base = nullptr;
    // three-level plans with the fused channelizer: the outer four-step factors of the row pass,
    // tw4o [outer][n2 / 16] = W_N^{tau k1o}, tw4u [outer][n1][16] = W_{16 n1}^{k1 j} W_65536^{k1o j}
    cf* tw4o = nullptr;
    cf* tw4u = nullptr;
    // the same twiddles applied by the 256-point column passes instead (col_twiddles):
    // twa [16][n2] = W_N^{tau n2}, twg [4][n2] = W_N^{16 n2 2^i}; BBT_OSM_TW_COL=0/1
    bool tw_col = false;
    cf* twa = nullptr;
    cf* twg = nullptr;
    cf* wroot = nullptr;
    cf* big_tw = nullptr;       // one-kernel plans of 8192 / 16384 samples (big_kernels.hpp): the transform's table (shared)
    // block lengths that are not powers of two (gen_kernels.hpp): N = n1 * n2,
    // n1 == 1 for N <= 8192
    bool generic = false;
    GenGeo g1 = {}, g2 = {}, g2r = {};      // g2r: the stages of g2 reversed (inverse of the row transform)
    cf* wn2r = nullptr;
    cf* wn1 = nullptr;          // W_{n1}^k (shared table)
    cf* wn2 = nullptr;          // W_{n2}^k (shared table)
    cf* tlo = nullptr;          // W_N^i, i < 4096        (owned)
    cf* thi = nullptr;          // W_N^{4096 j}           (owned)
    cf* tws = nullptr;          // W_N^{k1 m r}, [n1][g2.fac[0]], m = n2 / g2.fac[0]   (owned)
    int gen_ct = 1;             // columns per tile of the column passes
    // the same plan on the run-time specialised engine (gen2_kernels.hpp; null functions: not in use)
    bool rtc = false;
    G2Plan q1, q2, q2r;         // column transform (ct columns), row transform and its reversal
    cf* qw1 = nullptr;          // their stage tables (shared)
    cf* qw2 = nullptr;
    cf* qw2r = nullptr;
    hipFunction_t k2_small = nullptr, k2_first = nullptr, k2_row = nullptr, k2_last = nullptr;
    int n2p = 0;                // pitch of a row of the work buffer: n2 rounded up to whole 128-byte lines
    // pair-planar hand-over (bbt_osm_plan_set_layout; OsmChunk::in_plane / out_plane)
    long long in_plane = 0, out_plane = 0;
    // fused channelizer
    float2* seam = nullptr;     // [blocks][2][npair][n_chan] float4 (the turn in use)
    float2* seam_buf[2] = {nullptr, nullptr};
    size_t seam_bytes[2] = {0, 0};

    // timing
    bool timing = false;
    bool timing_isolated = false;   // mode 2: single lane, passes do not overlap
    std::vector<hipEvent_t> ev;  // 4 per chunk launch: t0, tA, tB, tC
    std::vector<hipEvent_t> ev_free;   // recycled events
    int timing_stride = 4;          // events on every timing_stride-th chunk of a lane
    int64_t pass_launches[3] = {0, 0, 0};
    int64_t pass_blocks[3] = {0, 0, 0};      // blocks the timed launches of each pass covered
    std::vector<int> ev_nblk;                // lanes schedule: blocks of each timed chunk launch
    double acc_ms[3] = {0, 0, 0};
    int64_t launches = 0;
};

template <int N2, int NCH>
static void launch_rowpass_t(bbt_osm_plan* p, float2* work, const OsmChunk& ch, hipStream_t st,
                             int y0 = 0, int ny = -1) {
    // two-level plans: a flat grid, the workgroups that share a response row on one XCD (k_osm_rowpass)
    const bool flat = p->outer == 1 && ch.nblk * p->npair > 1 &&
                      (long long)p->n1 * ch.nblk * p->npair < (1ll << 31);
    // (three-level plans may launch a range [y0, y0 + ny) of the outer rows)
    const int rows = ny >= 0 ? ny : ch.nblk * p->npair * p->outer;
    hipLaunchKernelGGL((k_osm_rowpass<N2, NCH>),
                       flat ? dim3(p->n1 * ch.nblk * p->npair, 1) : dim3(p->n1, rows),
                       dim3(N2 / 16), 0,
                       st, work, p->n1, p->resp, p->resp_index, p->npair, p->tab2.tw0, p->tab2.tw1,
                       p->wroot, ch, p->outer, y0, p->tw4row, p->tw4base,
                       p->tw_col ? (NCH ? 1 : 3) : 0, p->tw4o, p->tw4u);
}

// (row length, channels) -> instantiation; nch == 0 is the plain row pass.
static int launch_rowpass(bbt_osm_plan* p, float2* work, const OsmChunk& ch, int nch,
                          hipStream_t st, int y0 = 0, int ny = -1) {
#define BBT_RP(N2_, NCH_)                                       \
    if (p->n2 == N2_ && nch == NCH_) {                          \
        launch_rowpass_t<N2_, NCH_>(p, work, ch, st, y0, ny);   \
        return 0;                                               \
    }
    BBT_RP(256, 0) BBT_RP(512, 0) BBT_RP(1024, 0) BBT_RP(2048, 0) BBT_RP(4096, 0)
    BBT_RP(256, 2) BBT_RP(256, 4) BBT_RP(256, 8) BBT_RP(512, 2) BBT_RP(512, 4) BBT_RP(512, 8)
    BBT_RP(1024, 2) BBT_RP(1024, 4) BBT_RP(1024, 8) BBT_RP(2048, 2) BBT_RP(2048, 4) BBT_RP(2048, 8)
    BBT_RP(4096, 2) BBT_RP(4096, 4) BBT_RP(4096, 8)
    BBT_RP(256, 16) BBT_RP(256, 32) BBT_RP(256, 64) BBT_RP(256, 128)
    BBT_RP(512, 16) BBT_RP(512, 32) BBT_RP(512, 64) BBT_RP(512, 128)
    BBT_RP(1024, 16) BBT_RP(1024, 32) BBT_RP(1024, 64) BBT_RP(1024, 128)
    BBT_RP(2048, 16) BBT_RP(2048, 32) BBT_RP(2048, 64) BBT_RP(2048, 128)
    BBT_RP(4096, 16) BBT_RP(4096, 32) BBT_RP(4096, 64) BBT_RP(4096, 128)
    BBT_RP(256, 256)
    BBT_RP(512, 256) BBT_RP(512, 512)
    BBT_RP(1024, 256) BBT_RP(1024, 512) BBT_RP(1024, 1024)
    BBT_RP(2048, 256) BBT_RP(2048, 512) BBT_RP(2048, 1024) BBT_RP(2048, 2048)
    BBT_RP(4096, 256) BBT_RP(4096, 512) BBT_RP(4096, 1024) BBT_RP(4096, 2048) BBT_RP(4096, 4096)
#undef BBT_RP
    return fail("osm: no row pass for row length %d with %d channels", p->n2, nch);
}

// pairs per workgroup of the one-kernel plans at most (round 4's rule; BBT_SMALL_CAP overrides: dev)
#define BBT_SMALL_CAP(N_) ((N_) <= 512 ? 8 : ((N_) <= 2048 ? 4 : 2))
template <int N>
static int launch_small(bbt_osm_plan* p, const float2* in, float2* out, const OsmChunk& ch,
                        hipStream_t st) {
    // lanes over groups of pairs when there are many (see k_osm_small): the largest of 8, 4, 2 pairs
    // that divides the pair count and fits a workgroup (1024 threads; the interleaved exchange buffer
    // is dynamic LDS, up to 136 KiB)
    constexpr size_t lds1 = FftGeo<N>::LDS_ELEMS * sizeof(v2);
    const int nblk = ch.reg_count ? ch.reg_count : ch.nblk;
    if (p->single) {
        hipLaunchKernelGGL((k_osm_small<N, 1, true>), dim3((nblk + 1) / 2), dim3(N / 16), lds1, st, in,
                           out, ch, 1, p->resp, p->resp_index, p->tab2.tw0, p->tab2.tw1);
        return 0;
    }
    constexpr int CAP = BBT_SMALL_CAP(N);
    const int cap = getenv("BBT_SMALL_CAP") ? atoi(getenv("BBT_SMALL_CAP")) : CAP;         // (dev)
#define BBT_SMALL_PP(PP_)                                                                                    \
    if ((PP_ * N / 16 <= 1024 && lds1 * PP_ <= 160 * 1024) && PP_ <= cap && p->npair % PP_ == 0) {          \
        constexpr int Q = (PP_ * N / 16 <= 1024 && lds1 * PP_ <= 160 * 1024) ? PP_ : 1;                      \
        if (ensure_dyn_lds((const void*)k_osm_small<N, Q>, lds1 * Q)) return 1;                              \
        hipLaunchKernelGGL((k_osm_small<N, Q>), dim3(nblk * (p->npair / Q)), dim3(Q * N / 16), lds1 * Q,   \
                           st, in, out, ch, p->S, p->resp, p->resp_index, p->tab2.tw0, p->tab2.tw1);        \
        return 0;                                                                                            \
    }
    BBT_SMALL_PP(8) BBT_SMALL_PP(4) BBT_SMALL_PP(2)
#undef BBT_SMALL_PP
    hipLaunchKernelGGL((k_osm_small<N, 1>), dim3(nblk * p->npair), dim3(N / 16), lds1, st, in, out,
                       ch, p->S, p->resp, p->resp_index, p->tab2.tw0, p->tab2.tw1);
    return 0;
}

template <int N>
static int launch_small_big(bbt_osm_plan* p, const float2* in, float2* out, const OsmChunk& ch, hipStream_t st) {
    constexpr size_t lds = BigGeo<N>::LDS_ELEMS * sizeof(v2);
    const int nblk = ch.reg_count ? ch.reg_count : ch.nblk;
    if (p->single) {
        if (ensure_dyn_lds((const void*)k_osm_small_big<N, true>, lds)) return 1;
        hipLaunchKernelGGL((k_osm_small_big<N, true>), dim3((nblk + 1) / 2), dim3(N / 16), lds, st, in, out, ch,
                           1, p->resp, p->resp_index, p->big_tw);
        return 0;
    }
    if (ensure_dyn_lds((const void*)k_osm_small_big<N>, lds)) return 1;
    hipLaunchKernelGGL((k_osm_small_big<N>), dim3(nblk * p->npair), dim3(N / 16), lds, st, in, out, ch, p->S,
                       p->resp, p->resp_index, p->big_tw);
    return 0;
}

static int osm_flush_timing(bbt_osm_plan* p) {
    for (size_t i = 0; i + 3 < p->ev.size(); i += 4) {
        HIP_TRY(hipEventSynchronize(p->ev[i + 3]));
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, p->ev[i + k], p->ev[i + k + 1]));
            p->acc_ms[k] += ms;
            p->pass_launches[k] += 1;
            p->pass_blocks[k] += p->ev_nblk[i / 4];
        }
        p->launches += 1;
    }
    for (auto e : p->ev) p->ev_free.push_back(e);
    p->ev.clear();
    p->ev_nblk.clear();
    return 0;
}

template <bool FIRST, bool SPEC, int N1>
static int launch_col_long(bbt_osm_plan* p, const float2* in, float2* out, float2* work,
                           const OsmChunk& ch, int row_len, const SpecOut& so, hipStream_t st) {
    constexpr int F = 8;
    constexpr size_t lds = FftGeo<N1>::LDS_ELEMS * sizeof(v2) * F;
    if (p->single) {
        if (ensure_dyn_lds((const void*)k_osm_col256<FIRST, SPEC, F, false, 1, true, N1>, lds)) return 1;
        hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, F, false, 1, true, N1>),
                           dim3(row_len / F, (ch.nblk + 1) / 2), dim3(F * (N1 / 16)), lds, st, in, out, work,
                           ch, 1, row_len, p->tab1.tw0, so, p->tab1.tw1);
    } else {
        if (ensure_dyn_lds((const void*)k_osm_col256<FIRST, SPEC, F, false, 1, false, N1>, lds)) return 1;
        hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, F, false, 1, false, N1>),
                           dim3(row_len / F * p->npair, ch.nblk), dim3(F * (N1 / 16)), lds, st, in, out, work,
                           ch, p->S, row_len, p->tab1.tw0, so, p->tab1.tw1);
    }
    return 0;
}

template <bool FIRST, bool SPEC>
static int launch_col256(bbt_osm_plan* p, const float2* in, float2* out, float2* work,
                          const OsmChunk& ch, int row_len, const SpecOut& so_arg, hipStream_t st) {
    constexpr size_t lds1 = FftGeo<256>::LDS_ELEMS * sizeof(v2);     // per column of a tile
    SpecOut so = so_arg;
    // (the row pass then leaves the four-step twiddles to this pass: the forward ones always, the
    // inverse ones unless the fused channelizer's transform stands between them and this pass)
    if (p->tw_col && p->outer == 1 && (FIRST || !SPEC)) {
        so.twa = p->twa;
        so.twg = p->twg;
    }
    if (p->n1 == 512 || p->n1 == 1024) {
        // 512- / 1024-point columns (2^21- / 2^22-sample blocks): 8 columns per workgroup, 128-byte runs
        if (so.det) return fail("osm: fused detection needs 256-point columns");
        return p->n1 == 512 ? launch_col_long<FIRST, SPEC, 512>(p, in, out, work, ch, row_len, so, st)
                            : launch_col_long<FIRST, SPEC, 1024>(p, in, out, work, ch, row_len, so, st);
    }
    if (p->single) {
        if (so.det) return fail("osm: one-stream plans have no fused detection");
        hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, 16, false, 1, true>),
                           dim3(row_len / 16, (ch.nblk + 1) / 2), dim3(256), 16 * lds1, st, in, out, work,
                           ch, 1, row_len, p->tab1.tw0, so);
        return 0;
    }
    if constexpr (SPEC && !FIRST) {
        if (so.det) {
            hipLaunchKernelGGL((k_osm_col256<false, true, 16, true>),
                               dim3(row_len / 16 * p->npair, ch.nblk), dim3(256), 16 * lds1, st, in, out,
                               work, ch, p->S, row_len, p->tab1.tw0, so);
            return 0;
        }
    }
    if (FIRST && ch.in_plane) {
        // pair-planar input: every pair is a two-stream array, the plain 16-column tiles read
        // 256-byte runs of it (the arrangements below are for interleaved rows)
        hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, 16>), dim3(row_len / 16 * p->npair, ch.nblk),
                           dim3(256), 16 * lds1, st, in, out, work, ch, p->S, row_len, p->tab1.tw0, so);
        return 0;
    }
    // Many streams: the lanes of a row go over groups of stream pairs so that the stream side
    // moves whole lines.  Measured on MI355X (DESIGN, "column-pass tiles"):
    //   pairs in eights   first pass 8 pairs x 8 columns (64 lanes, 1024 threads: 1 KiB of a row of
    //                     the stream and 128-byte runs of work), last pass 8 pairs x 4 columns (512
    //                     threads, no spills): config 4's share +4.2 %, 16 / 32 streams +5-7 %
    //   pairs in fours    first pass from 8 pairs on 4 pairs x 16 columns (1024 threads; 16 streams
    //                     +3 %, 8 streams -5 %), last pass 4 pairs x 4 columns (8 streams +11 %)
    if (p->npair % 8 == 0) {
        constexpr int F = FIRST ? 64 : 32;
        if (ensure_dyn_lds((const void*)k_osm_col256<FIRST, SPEC, F, false, 8>, F * lds1)) return 1;
        hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, F, false, 8>),
                           dim3(row_len / (F / 8) * (p->npair / 8), ch.nblk), dim3(F * 16), F * lds1, st,
                           in, out, work, ch, p->S, row_len, p->tab1.tw0, so);
        return 0;
    }
    if constexpr (FIRST) {
        if (p->npair % 4 == 0 && p->npair >= 8) {
            if (ensure_dyn_lds((const void*)k_osm_col256<true, SPEC, 64, false, 4>, 64 * lds1)) return 1;
            hipLaunchKernelGGL((k_osm_col256<true, SPEC, 64, false, 4>),
                               dim3(row_len / 16 * (p->npair / 4), ch.nblk), dim3(1024), 64 * lds1, st, in,
                               out, work, ch, p->S, row_len, p->tab1.tw0, so);
            return 0;
        }
    } else if (p->npair % 4 == 0) {
        hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, 16, false, 4>),
                           dim3(row_len / 4 * (p->npair / 4), ch.nblk), dim3(256), 16 * lds1, st, in,
                           out, work, ch, p->S, row_len, p->tab1.tw0, so);
        return 0;
    }
    // Plain 16-column tiles.  The first pass (116 VGPRs, 34 KiB of LDS) fits four workgroups per
    // CU, 464 registers per lane of a SIMD: no row-pass wave (162) of the other lane can join
    // them.  Asking for 9 KiB more of (unused) LDS leaves three (348 + 162 = 510 of 512): config 2
    // +2.4 %, headline +0.3 %, four pairs -0.5 % -- hence for one pair only.
    const int col_pad = (FIRST && p->npair == 1) ? 9216 : 0;
    if (col_pad && ensure_dyn_lds((const void*)k_osm_col256<FIRST, SPEC, 16>, 16 * lds1 + col_pad)) return 1;
    hipLaunchKernelGGL((k_osm_col256<FIRST, SPEC, 16>), dim3(row_len / 16 * p->npair, ch.nblk),
                       dim3(256), 16 * lds1 + col_pad, st, in, out, work, ch, p->S, row_len, p->tab1.tw0, so);
    return 0;
}

static int osm_launch_pass(bbt_osm_plan* p, int pass, const float2* in, float2* out, const OsmChunk& ch,
                           const SpecOut& so, float2* work, hipStream_t st) {
    const int nch = so.n_chan;
    OsmChunk pairs_view;               // one stream: the work buffers hold pairs of blocks
    if (p->single) {
        pairs_view = ch;
        pairs_view.nblk = (ch.nblk + 1) / 2;
    }
    const OsmChunk& chw = p->single ? pairs_view : ch;       // what the row pass counts
    const dim3 g16(p->n2 / 256 * p->npair, chw.nblk);        // (the same count with 8 pairs x 32 columns per workgroup)
    const bool pp8 = p->n1 == 16 && !p->single && p->npair % 8 == 0;
    if (pass == 0) {
        if (p->n1 == 16 && p->single)
            hipLaunchKernelGGL((k_osm_col16<true, false, true>), g16, dim3(256), 0, st, in, out, work, ch,
                               1, p->n2, so);
        else if (pp8)
            hipLaunchKernelGGL((k_osm_col16<true, false, false, 8>), g16, dim3(256), 0, st, in, out, work, ch,
                               p->S, p->n2, so);
        else if (p->n1 == 16)
            hipLaunchKernelGGL((k_osm_col16<true, false>), g16, dim3(256), 0, st, in, out, work, ch,
                               p->S, p->n2, so);
        else if (launch_col256<true, false>(p, in, out, work, ch, p->n2, so, st))
            return 1;
    } else if (pass == 1) {
        if (launch_rowpass(p, work, chw, nch, st)) return 1;
    } else if (p->n1 == 16) {
        if (p->single && nch)
            hipLaunchKernelGGL((k_osm_col16<false, true, true>), g16, dim3(256), 0, st, in, out,
                               work, ch, 1, p->n2, so);
        else if (p->single)
            hipLaunchKernelGGL((k_osm_col16<false, false, true>), g16, dim3(256), 0, st, in, out,
                               work, ch, 1, p->n2, so);
        else if (nch && pp8)
            hipLaunchKernelGGL((k_osm_col16<false, true, false, 8>), g16, dim3(256), 0, st, in, out,
                               work, ch, p->S, p->n2, so);
        else if (pp8)
            hipLaunchKernelGGL((k_osm_col16<false, false, false, 8>), g16, dim3(256), 0, st, in, out,
                               work, ch, p->S, p->n2, so);
        else if (nch)
            hipLaunchKernelGGL((k_osm_col16<false, true>), g16, dim3(256), 0, st, in, out,
                               work, ch, p->S, p->n2, so);
        else
            hipLaunchKernelGGL((k_osm_col16<false, false>), g16, dim3(256), 0, st, in, out,
                               work, ch, p->S, p->n2, so);
    } else {
        if (nch ? launch_col256<false, true>(p, in, out, work, ch, p->n2, so, st)
                : launch_col256<false, false>(p, in, out, work, ch, p->n2, so, st))
            return 1;
    }
    return 0;
}

static int osm_run_chunk(bbt_osm_plan* p, const float2* in, float2* out, const OsmChunk& ch_given,
                         const SpecOut& so, float2* work, hipStream_t st, bool sample = true) {
    const int nch = so.n_chan;   // 0: plain overlap-save output
    OsmChunk ch_arg = ch_given;
    ch_arg.in_plane = p->in_plane;
    ch_arg.out_plane = p->out_plane;
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    const bool timed = p->timing && sample;       // (events on this chunk's kernels)
    if (timed) {
        for (int i = 0; i < 4; ++i) {
            if (!p->ev_free.empty()) {
                e[i] = p->ev_free.back();
                p->ev_free.pop_back();
            } else {
                HIP_TRY(hipEventCreateWithFlags(&e[i], hipEventDisableSystemFence));
            }
        }
        HIP_TRY(hipEventRecord(e[0], st));
    }
    const OsmChunk& ch = ch_arg;
    OsmChunk pairs_view;               // one stream: the work buffers hold pairs of blocks
    if (p->single) {
        pairs_view = ch;
        pairs_view.nblk = (ch.nblk + 1) / 2;
    }
    const OsmChunk& chw = p->single ? pairs_view : ch;       // what the row / middle passes count
    if (p->generic && p->rtc) {
        if (nch) return fail("osm: the fused channelizer needs a power-of-two block length");
        if (p->n1 == 1) {
            if (g2_launch(p->k2_small, dim3(ch.nblk * p->npair), dim3(p->q2.threads()), st, (const float2*)in, out, ch,
                          p->S, (const cf*)p->resp, (const int*)p->resp_index, (const cf*)p->qw2, (const cf*)p->qw2r))
                return 1;
            if (timed) {
                HIP_TRY(hipEventRecord(e[1], st));
                HIP_TRY(hipEventRecord(e[2], st));
            }
        } else {
            const int ct = p->gen_ct, tiles = (p->n2 + ct - 1) / ct;
            const dim3 gcol((unsigned)tiles * p->npair * ch.nblk), bcol(p->q1.threads());
            if (g2_launch(p->k2_first, gcol, bcol, st, (const float2*)in, out, work, ch, p->S, p->n2, p->n2p,
                          (const cf*)p->qw1))
                return 1;
            if (timed) HIP_TRY(hipEventRecord(e[1], st));
            if (g2_launch(p->k2_row, dim3((p->n1 + p->q2.ct - 1) / p->q2.ct, ch.nblk * p->npair), dim3(p->q2.threads()), st, work, p->n1, p->n2p,
                          (const cf*)p->resp, (const int*)p->resp_index, p->npair, (const cf*)p->qw2,
                          (const cf*)p->qw2r, (const cf*)p->tlo, (const cf*)p->thi, (const cf*)p->tws))
                return 1;
            if (timed) HIP_TRY(hipEventRecord(e[2], st));
            if (g2_launch(p->k2_last, gcol, bcol, st, (const float2*)in, out, work, ch, p->S, p->n2, p->n2p,
                          (const cf*)p->qw1))
                return 1;
        }
    } else if (p->generic) {
        if (nch) return fail("osm: the fused channelizer needs a power-of-two block length");
        if (p->n1 == 1) {
            const size_t lds = (size_t)p->n2 * sizeof(f4);
            if (ensure_dyn_lds((const void*)k_gen_osm_small, lds)) return 1;
            hipLaunchKernelGGL(k_gen_osm_small, dim3(ch.nblk * p->npair), dim3(gen_threads(p->n2)), lds,
                               st, in, out, ch, p->S, p->resp, p->resp_index, p->g2, p->wn2, p->g2r, p->wn2r);
            if (timed) {
                HIP_TRY(hipEventRecord(e[1], st));
                HIP_TRY(hipEventRecord(e[2], st));
            }
        } else {
            const int ct = p->gen_ct, tiles = (p->n2 + ct - 1) / ct;
            const size_t lds_c = (size_t)p->n1 * ct * sizeof(f4), lds_r = (size_t)p->n2 * sizeof(f4);
            if (ensure_dyn_lds((const void*)k_gen_col<true>, lds_c) ||
                ensure_dyn_lds((const void*)k_gen_col<false>, lds_c) ||
                ensure_dyn_lds((const void*)k_gen_row, lds_r))
                return 1;
            const dim3 gcol(tiles * p->npair, ch.nblk), bcol(gen_threads(p->n1 * ct));
            hipLaunchKernelGGL((k_gen_col<true>), gcol, bcol, lds_c, st, in, out, work, ch, p->S, p->n2,
                               ct, p->g1, p->wn1);
            if (timed) HIP_TRY(hipEventRecord(e[1], st));
            hipLaunchKernelGGL(k_gen_row, dim3(p->n1, ch.nblk * p->npair), dim3(gen_threads(p->n2)),
                               lds_r, st, work, p->n1, p->resp, p->resp_index, p->npair, p->g2, p->wn2, p->g2r, p->wn2r,
                               p->tlo, p->thi, p->tws);
            if (timed) HIP_TRY(hipEventRecord(e[2], st));
            hipLaunchKernelGGL((k_gen_col<false>), gcol, bcol, lds_c, st, in, out, work, ch, p->S, p->n2,
                               ct, p->g1, p->wn1);
        }
    } else if (p->n1 == 1) {
        int rc = 0;
        switch (p->n2) {
            case 256: rc = launch_small<256>(p, in, out, ch, st); break;
            case 512: rc = launch_small<512>(p, in, out, ch, st); break;
            case 1024: rc = launch_small<1024>(p, in, out, ch, st); break;
            case 2048: rc = launch_small<2048>(p, in, out, ch, st); break;
            case 4096: rc = launch_small<4096>(p, in, out, ch, st); break;
            case 8192: rc = launch_small_big<8192>(p, in, out, ch, st); break;
            case 16384: rc = launch_small_big<16384>(p, in, out, ch, st); break;
            default: return fail("osm: unsupported n_fft %lld", (long long)p->n);
        }
        if (rc) return rc;
        if (timed) {
            HIP_TRY(hipEventRecord(e[1], st));
            HIP_TRY(hipEventRecord(e[2], st));
        }
    } else if (p->outer > 1) {
        // three levels: outer 256-point column pass over rows of M = 16 * n2, then
        // the two-level machinery in place on every outer row
        // The three middle passes (outer twiddle + 16-point step, row pass, and back) act on
        // every outer row by itself, in place: they go over the work buffer in pieces of
        // BBT_OSM_MID_MIB (default 128 MiB), so that a piece can still be in the Infinity Cache
        // when the next pass comes for it.  Measured on MI355X (config 4's share, two runs each):
        // whole buffer 2.67 / 2.69, pieces of 32 MiB 2.57, 64 MiB 2.70 / 2.71, 128 MiB 2.75 / 2.75,
        // 256 MiB 2.70 / 2.76 G complete samples/s -- +3 %, not the 1.5x fewer HBM bytes would give:
        // the other lane's outer column passes stream 4 GiB through the cache meanwhile (making
        // those accesses non-temporal cost 7 %, one lane 8 %).
        const int m_len = 16 * p->n2;
        const int rows_all = chw.nblk * p->npair * 256;
        static const long long mid_bytes = [] { const char* e = getenv("BBT_OSM_MID_MIB"); return (long long)(e ? atoi(e) : 128) << 20; }();
        const long long row_bytes = 16ll * p->n2 * 16;
        int step = mid_bytes > 0 ? (int)std::max<long long>(1, mid_bytes / row_bytes) : rows_all;
        if (p->timing) step = rows_all;            // (per-pass events: the passes one after the other)
        if (launch_col256<true, false>(p, in, out, work, ch, m_len, so, st)) return 1;
        for (int y0 = 0; y0 < rows_all; y0 += step) {
            const int ny = std::min(step, rows_all - y0);
            const dim3 gmid(p->n2 / 256, ny);
            hipLaunchKernelGGL((k_osm_mid16<true>), gmid, dim3(256), 0, st, work, p->n2, (int)p->n,
                               p->wroot, 0, y0);
            if (timed) HIP_TRY(hipEventRecord(e[1], st));
            if (launch_rowpass(p, work, chw, nch, st, y0, ny)) return 1;
            if (timed) HIP_TRY(hipEventRecord(e[2], st));
            hipLaunchKernelGGL((k_osm_mid16<false>), gmid, dim3(256), 0, st, work, p->n2, (int)p->n,
                               p->wroot, nch ? 1 : 0, y0);
        }
        if (nch ? launch_col256<false, true>(p, in, out, work, ch, m_len, so, st)
                : launch_col256<false, false>(p, in, out, work, ch, m_len, so, st))
            return 1;
    } else {
        if (osm_launch_pass(p, 0, in, out, ch, so, work, st)) return 1;
        if (timed) HIP_TRY(hipEventRecord(e[1], st));
        if (osm_launch_pass(p, 1, in, out, ch, so, work, st)) return 1;
        if (timed) HIP_TRY(hipEventRecord(e[2], st));
        if (osm_launch_pass(p, 2, in, out, ch, so, work, st)) return 1;
    }
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(e[3], st));
        for (int i = 0; i < 4; ++i) p->ev.push_back(e[i]);
        p->ev_nblk.push_back(ch_arg.nblk);
        if (p->ev.size() >= 65536) return osm_flush_timing(p);   // (a flush waits for the events)
    }
    return 0;
}

// One execute call at a time per plan -- on the host (the mutex) and on the
// device: the work, staging and seam buffers belong to the running call, so a
// call queued on another stream than the previous one first waits for that
// call's last kernel.
//
// Deferred calls (bbt_osm_plan_defer) are the exception: a forking call that was
// given a completion event leaves the caller's stream unordered with respect to
// its lanes, and the next forking deferred call does not wait for it either --
// the lanes are in-order queues, a lane's work buffer is only ever touched from
// that lane, and everything else (the seam buffer) has two turns.  `fin` is the
// stream on which such a call ends: whatever must follow the lanes goes there.
// The completion event handed in by bbt_osm_plan_defer belongs to the NEXT execute call on the
// plan, whatever becomes of that call: every entry point takes it first thing, and if the call
// returns before anything was queued (an argument error, nothing to do) the event is recorded on
// the caller's stream as it stands.
struct DeferTake {
    hipEvent_t ev = nullptr;
    hipStream_t st;
    DeferTake(bbt_osm_plan* p, hipStream_t stream) : st(stream) {
        if (!p) return;
        std::lock_guard<std::mutex> lock(p->mu);
        ev = p->defer_ev;
        p->defer_ev = nullptr;
    }
    hipEvent_t hand_over() {
        hipEvent_t e = ev;
        ev = nullptr;
        return e;
    }
    void give_back(bbt_osm_plan* p) {        // (for an entry point that ends in another one)
        std::lock_guard<std::mutex> lock(p->mu);
        p->defer_ev = hand_over();
    }
    ~DeferTake() {
        if (ev) (void)hipEventRecord(ev, st);
    }
};

struct PlanCall {
    bbt_osm_plan* p;
    hipStream_t st;
    std::lock_guard<std::mutex> lock;
    hipEvent_t defer;           // the caller's completion event, or null
    bool fork;                  // chunks go to the lane streams
    bool deferred;              // ... and `st` is not joined after them
    hipStream_t fin;            // where the call ends
    hipStream_t run;            // where chunks that are not forked run: `st`, or -- a deferred call of a
                                // plan without lanes (one kernel per chunk) -- the tail stream, so that
                                // what the caller queues on `st` next (the upstream task's kernels for
                                // the following run) overlaps it
    PlanCall(bbt_osm_plan* plan, DeferTake& take, int64_t n_blocks)
        : p(plan), st(take.st), lock(plan->mu), defer(take.hand_over()) {
        const int64_t n_chunks = (n_blocks + p->chunk - 1) / p->chunk;
        const bool usual = !(p->timing && p->timing_isolated);
        const bool lanes_ok = p->lanes > 1 && usual && n_chunks > 0;
        const bool aside = p->lanes == 1 && usual;
        deferred = defer && p->tail_stream && (lanes_ok || aside);
        fork = lanes_ok && (n_chunks > 1 || deferred);
        fin = deferred ? p->tail_stream : st;
        run = deferred && !fork ? p->tail_stream : st;
        // (two deferred calls in a row: nothing of the earlier one is touched outside the lanes'
        // -- the tail stream's -- own order, see above)
        if (p->ev_done_set && !(deferred && p->last_forked_deferred))
            (void)hipStreamWaitEvent(st, p->ev_done, 0);
        if (run != st) {                   // the tail stream takes over from the caller's
            (void)hipEventRecord(p->ev_fork, st);
            (void)hipStreamWaitEvent(run, p->ev_fork, 0);
        }
    }
    ~PlanCall() {
        if (defer) (void)hipEventRecord(defer, fin);
        if (hipEventRecord(p->ev_done, fin) == hipSuccess) p->ev_done_set = true;
        p->last_forked_deferred = deferred;
    }
};

// Run all chunks, alternating lanes.  Returns with `call.st` ordered after every lane, or -- a
// deferred call -- with `call.fin` (the plan's tail stream) ordered after them instead.
template <class FillChunk>
static int osm_run_all(bbt_osm_plan* p, const float2* in, float2* out, int64_t n_blocks,
                       const SpecOut& so, const PlanCall& call, FillChunk fill) {
    hipStream_t st = call.st;
    const bool fork = call.fork;
    if (fork) {
        HIP_TRY(hipEventRecord(p->ev_fork, st));
        for (int l = 0; l < p->lanes; ++l) HIP_TRY(hipStreamWaitEvent(p->lane_stream[l], p->ev_fork, 0));
    }
    // (lanes keep alternating across deferred calls; a joined call starts on lane 0 as ever)
    const int l0 = call.deferred ? p->lane_cursor : 0;
    // regular runs (the fused channelizer's per-block shift and seam slot follow from the hop: osm_block)
    const bool regular_ok = p->cap > p->chunk;
    const int mask = so.n_chan ? so.n_chan - 1 : 0;
    auto follows = [mask](const OsmBlock& a, const OsmBlock& b, long long d) {
        return b.in_off - a.in_off == d && b.out_off - a.out_off == d && b.valid_start == a.valid_start &&
               b.valid_count == a.valid_count && !a.flat && !b.flat &&
               b.shift == (int)(((long long)a.shift - d) & mask);      // (index: the fills count blocks up)
    };
    int64_t c = 0;
    auto launch = [&](const OsmChunk& ch) {
        const int l = fork ? (int)((l0 + c) % p->lanes) : 0;
        const bool sample = !fork || (c / p->lanes) % p->timing_stride == 0;
        ++c;
        return osm_run_chunk(p, in, out, ch, so, p->lane_work[l], fork ? p->lane_stream[l] : call.run, sample);
    };
    for (int64_t b0 = 0; b0 < n_blocks;) {
        OsmChunk ch = {};                  // (fields a caller's fill does not set stay 0)
        fill(ch.b[0], b0);
        int64_t run = 1;                   // blocks from b0 on that step regularly
        long long hop = 0;
        if (regular_ok && b0 + p->chunk < n_blocks) {
            OsmBlock nx = {};
            fill(nx, b0 + 1);
            hop = nx.in_off - ch.b[0].in_off;
            if (hop > 0 && follows(ch.b[0], nx, hop))
                for (run = 2; b0 + run < n_blocks; ++run) {
                    fill(nx, b0 + run);
                    if (!follows(ch.b[0], nx, hop * run)) break;
                }
        }
        if (run > p->chunk) {
            // launches of equal size, as few as the work buffer allows -- one per lane at least when
            // each still gets more than a chunk of descriptors would hold
            int64_t parts = (run + p->cap - 1) / p->cap;
            if (fork && parts < p->lanes && run >= (int64_t)p->lanes * 2 * p->chunk) parts = p->lanes;
            const int64_t per = (run + parts - 1) / parts;
            const OsmBlock first = ch.b[0];
            for (int64_t r0 = 0; r0 < run; r0 += per) {
                OsmChunk rc = {};
                rc.b[0] = first;
                rc.b[0].in_off += r0 * hop;
                rc.b[0].out_off += r0 * hop;
                rc.b[0].shift = (int)(((long long)first.shift - r0 * hop) & mask);
                rc.b[0].index = first.index + (int)r0;
                rc.reg_count = rc.nblk = (int)std::min(per, run - r0);
                rc.reg_hop = hop;
                rc.reg_mask = mask;
                if (launch(rc)) return 1;
            }
            b0 += run;
            continue;
        }
        ch.nblk = (int)((n_blocks - b0 < p->chunk) ? (n_blocks - b0) : p->chunk);
        for (int i = 1; i < ch.nblk; ++i) fill(ch.b[i], b0 + i);
        if (launch(ch)) return 1;
        b0 += ch.nblk;
    }
    if (fork) {
        if (call.deferred) p->lane_cursor = (int)((l0 + c) % p->lanes);
        for (int l = 0; l < p->lanes; ++l) {
            HIP_TRY(hipEventRecord(p->ev_join[l], p->lane_stream[l]));
            HIP_TRY(hipStreamWaitEvent(call.fin, p->ev_join[l], 0));
        }
    }
    return 0;
}

static int osm_check_blocks(const bbt_osm_plan* p, const char* who, int64_t n_blocks,
                            const int64_t* in_off, const int64_t* out_off,
                            const int32_t* valid_start, const int32_t* valid_count) {
    ARG_TRY(n_blocks >= 0, "%s: n_blocks=%lld < 0", who, (long long)n_blocks);
    ARG_TRY(n_blocks == 0 || (in_off && out_off && valid_start && valid_count),
            "%s: null descriptor array", who);
    for (int64_t b = 0; b < n_blocks; ++b) {
        ARG_TRY(in_off[b] >= 0 && out_off[b] >= 0, "%s: negative offset in block %lld", who,
                (long long)b);
        ARG_TRY(valid_start[b] >= 0 && valid_count[b] >= 0 &&
                    (int64_t)valid_start[b] + valid_count[b] <= p->n,
                "%s: block %lld keeps [%d, %d) outside [0, %lld)", who, (long long)b,
                valid_start[b], valid_start[b] + valid_count[b], (long long)p->n);
    }
    return 0;
}

template <int NCH>
static void launch_seam_fix(bbt_osm_plan* p, float2* out, const std::vector<SeamJob>& jobs,
                            const FftTables& tab, const SpecOut& so, hipStream_t st) {
    for (size_t j0 = 0; j0 < jobs.size(); j0 += BBT_SEAM_JOBS_PER_LAUNCH) {
        SeamJobs batch;
        const size_t n = std::min(jobs.size() - j0, (size_t)BBT_SEAM_JOBS_PER_LAUNCH);
        for (size_t i = 0; i < n; ++i) batch.j[i] = jobs[j0 + i];
        if (p->single)
            hipLaunchKernelGGL((k_seam_fix<NCH, true>), dim3((unsigned)n, 1), dim3(NCH / 16), 0, st, p->seam,
                               out, batch, 1, 1, tab.tw0, tab.tw1, so);
        else
            hipLaunchKernelGGL((k_seam_fix<NCH>), dim3((unsigned)n, p->npair), dim3(NCH / 16), 0, st,
                               p->seam, out, batch, p->S, p->npair, tab.tw0, tab.tw1, so);
    }
}

extern "C" {

int bbt_osm_plan_create(bbt_osm_plan** plan, int64_t n_fft, int n_stream, int n_resp,
                        const void* resp, int resp_on_device, const int32_t* resp_index) {
    ARG_TRY(plan && resp, "bbt_osm_plan_create: null argument");
    *plan = nullptr;
    const bool fast = is_pow2(n_fft) && n_fft >= 256 && n_fft <= (1 << 24);
    int gn1 = 1, gn2 = 0;
    ARG_TRY(fast || (is_7smooth(n_fft) && n_fft >= 2 &&
                     (n_fft <= BBT_GEN_MAX_LEN || split_7smooth(n_fft, &gn1, &gn2))),
            "bbt_osm_plan_create: n_fft=%lld must be a power of two in [256, 2^24] or a product of "
            "2, 3, 5, 7 that is <= 8192 or splits into two such factors",
            (long long)n_fft);
    const bool single = n_stream == 1 && fast;
    ARG_TRY(single || (n_stream >= 2 && n_stream % 2 == 0 && n_stream <= 65535 * 2),
            "bbt_osm_plan_create: n_stream=%d must be even and >= 2 (or 1 with a power-of-two block "
            "length)", n_stream);
    ARG_TRY(n_resp >= 1, "bbt_osm_plan_create: n_resp=%d must be >= 1", n_resp);
    std::vector<int> idx(single ? 2 : n_stream, 0);
    if (resp_index)
        for (int s = 0; s < n_stream; ++s) {
            ARG_TRY(resp_index[s] >= 0 && resp_index[s] < n_resp,
                    "bbt_osm_plan_create: resp_index[%d]=%d out of range", s, resp_index[s]);
            idx[s] = resp_index[s];
        }
    if (single) idx[1] = idx[0];            // both halves of a pair of blocks are the one stream
    bbt_osm_plan* p = new bbt_osm_plan;
    auto bail = [&](int) {
        bbt_osm_plan_destroy(p);
        return 1;
    };
    if (hipGetDevice(&p->device) != hipSuccess) return bail(fail("hipGetDevice failed"));
    p->n = n_fft;
    p->S = n_stream;
    p->npair = single ? 1 : n_stream / 2;
    p->single = single;
    p->C = n_resp;
    if (!fast) {
        p->generic = true;
        if (n_fft <= BBT_GEN_MAX_LEN) {
            gn1 = 1;
            gn2 = (int)n_fft;
        }
        p->n1 = gn1;
    } else if (n_fft <= 4096 ||
               ((n_fft == 8192 || n_fft == 16384) && (single || (n_stream / 2) % 8 != 0) && !getenv("BBT_OSM_NO_BIG"))) {
        // (8192 / 16384 samples: one workgroup of 512 / 1024 threads, big_kernels.hpp -- one stream pair
        // per workgroup, 16 bytes of every complete sample: with pairs in eights the two-level plan, whose
        // 16-point column passes then move whole lines, is as fast at 8192 and faster at 16384 samples:
        // 16 / 128 / 2048 streams 113 / 109 / 84 against 84 / 88 / 58 G stream-samples/s)
        p->n1 = 1;
    } else if (n_fft <= (1 << 16)) {
        p->n1 = 16;
    } else if (n_fft <= (1 << 20)) {
        p->n1 = 256;
    } else if (n_fft == (1 << 21)) {
        // 512 x 4096: still two levels -- three streaming passes where 256 x 16 x 512 takes five.
        // (A 512-point column pass holds 8 columns in 48 KiB of exchange area, three workgroups
        // per CU; the row pass is the 4096-point one of the 2^20 blocks.)
        p->n1 = 512;
    } else if (n_fft == (1 << 22) && (single || (n_stream / 2) % 8 != 0)) {
        // (stream pairs in eights: three levels, whose 256-point column passes take 8 pairs per workgroup --
        // 16 / 128 streams 59.9 / 66.4 against 54.3 / 59.7 G stream-samples/s; at 2^21 two levels stay ahead)
        p->n1 = 1024;            // 1024 x 4096, likewise (8 columns: 80 KiB, two workgroups per CU)
    } else {
        p->outer = 256;          // three levels, 256 x 16 x N2
        p->n1 = 16;
        if (n_stream / 2 > 255)
            return bail(fail("bbt_osm_plan_create: blocks longer than 2^20 support at most 510 streams"));
    }
    p->n2 = (int)(n_fft / p->n1 / p->outer);
    if (p->generic) {
        if (!factor_7smooth(p->n2, &p->g2) || (p->n1 > 1 && !factor_7smooth(p->n1, &p->g1)))
            return bail(fail("bbt_osm_plan_create: cannot factor %d x %d", p->n1, p->n2));
        if (get_gen_table(&p->g2, &p->wn2) || get_reversed(p->g2, &p->g2r, &p->wn2r)) return bail(1);
        // columns per tile: a power of two (gen_stage), as many as fit the LDS tile up to 8
        // (128-byte runs of the stream and of the work buffer; measured: 8 columns 18.9, 16
        // columns 18.1, 4 columns 17.3 Gsamples/s for the 1 666 980-sample block)
        p->gen_ct = 1;
        if (p->n1 > 1) {
            // (short blocks on the compiled kernels: as many columns as fill a wave, gen2_host.hpp g2_col_ct)
            const int ct_cap = getenv("BBT_GEN_CT") ? atoi(getenv("BBT_GEN_CT"))         // (dev)
                               : (rtc_mode() && n_fft <= (1 << 17)) ? g2_col_ct(p->n1, g2_pmax(BBT_G2_KIND_COL)) : 8;
            while (p->gen_ct < ct_cap && 2 * p->gen_ct * p->n1 <= 2 * BBT_GEN_MAX_LEN) p->gen_ct *= 2;
        }
        // The kernels specialised on this length (fft_gen2.hpp), compiled now; if that is not
        // possible the plan runs on the general ones.
        if (rtc_mode()) {
            // (short rows: several neighbouring rows per workgroup, gen2_host.hpp g2_row_ct)
            const int row_ct = p->n1 > 1 ? g2_row_ct(p->n2, g2_pmax(BBT_G2_KIND_ROW)) : 1;
            bool ok = g2_plan(p->n2, row_ct, &p->q2, g2_pmax(BBT_G2_KIND_ROW));
            if (ok) p->q2r = g2_reversed(p->q2);
            if (ok && p->n1 > 1) {
                // (a column tile is one workgroup: at most 1024 threads and 64 KiB of exchange area)
                int ct = p->gen_ct;
                while ((ok = g2_plan(p->n1, ct, &p->q1, g2_pmax(BBT_G2_KIND_COL))) && ct > 1 &&
                       (p->q1.threads() > 1024 || p->q1.lds_elems * 8 > 64 * 1024))
                    ct /= 2;
                p->gen_ct = ct;
                ok = ok && p->q1.threads() <= 1024 && p->q1.lds_elems * 8 <= 64 * 1024;
            }
            ok = ok && p->q2.threads() <= 1024 && std::max(p->q2.lds_elems, p->q2r.lds_elems) * 8 <= 96 * 1024;
            if (!ok) fail("no stage list within a workgroup for %d x %d", p->n1, p->n2);
            if (ok) {
                std::string src = "#include \"gen2_kernels.hpp\"\n" + g2_trait_source("GA", p->q2) +
                                  g2_trait_source("GB", p->q2r);
                // (workgroups of 7 waves or more: at most 128 registers, two of them per CU)
                const std::string w2 = p->q2.threads() >= 448 ? "4" : "0";
                if (p->n1 == 1) {
                    src += "BBT_G2_KERNEL_OSM_SMALL(k_small, GA, GB, " + w2 + ")\n";
                    ok = !g2_build(src, {"k_small"}, &p->k2_small);
                } else {
                    const std::string w1 = p->q1.threads() >= 448 ? "4" : "0";
                    src += g2_trait_source("GC", p->q1) + "BBT_G2_KERNEL_ROW(k_row, GA, GB, " + w2 + ")\n"
                           "BBT_G2_KERNEL_COL(k_first, GC, true, " + w1 + ")\nBBT_G2_KERNEL_COL(k_last, GC, false, " + w1 + ")\n";
                    hipFunction_t f[3];
                    ok = !g2_build(src, {"k_row", "k_first", "k_last"}, f);
                    if (ok) {
                        p->k2_row = f[0];
                        p->k2_first = f[1];
                        p->k2_last = f[2];
                    }
                }
            }
            if (ok) ok = !(get_g2_table(p->q2, &p->qw2) || get_g2_table(p->q2r, &p->qw2r) ||
                           (p->n1 > 1 && get_g2_table(p->q1, &p->qw1)));
            if (!ok && rtc_mode() == 2) return bail(1);
            if (!ok) g2_warn_once("bbt_osm_plan_create");
            p->rtc = ok;
        }
        if (p->n1 > 1) {
            if (get_gen_table(&p->g1, &p->wn1) || make_big_twiddle(n_fft, &p->tlo, &p->thi))
                return bail(1);
            {   // the uniform factors of the row kernel's four-step twiddles (GenRowSrc): the first
                // forward and the last inverse stage have radix r0
                const int r0 = p->rtc ? p->q2.fac[0] : p->g2.fac[0];
                const long long m = p->n2 / r0;
                std::vector<cf> t((size_t)p->n1 * r0);
                for (int k1 = 0; k1 < p->n1; ++k1)
                    for (int r = 0; r < r0; ++r)
                        t[(size_t)k1 * r0 + r] = unit_root((long long)k1 * m % n_fft * r, n_fft);
                if (upload(&p->tws, t)) return bail(1);
            }
        }
    } else {
        if (p->n2 > 4096) {
            if (get_big_table(p->n2, &p->big_tw)) return bail(1);
        } else if (get_tables(p->n2, &p->tab2)) {
            return bail(1);
        }
        if ((p->n1 == 256 || p->outer == 256) && get_tables(256, &p->tab1)) return bail(1);
        if ((p->n1 == 512 || p->n1 == 1024) && get_tables(p->n1, &p->tab1)) return bail(1);
        if (get_wroot(&p->wroot)) return bail(1);
        if ((p->n1 == 16 || p->n1 == 256 || p->n1 == 512 || p->n1 == 1024) && (p->outer == 1 || p->outer == 256)) {
            // (three-level plans: these are the twiddles of the inner transform of n1 * n2 points)
            const int t = p->n2 / 16;
            const long long inner = (long long)p->n1 * p->n2;
            std::vector<cf> row((size_t)p->n1 * 16), base((size_t)p->n1 * t);
            for (int k1 = 0; k1 < p->n1; ++k1) {
                for (int j = 0; j < 16; ++j) row[(size_t)k1 * 16 + j] = unit_root((long long)k1 * j, 16ll * p->n1);
                for (int tau = 0; tau < t; ++tau)
                    base[(size_t)k1 * t + tau] = unit_root((long long)k1 * tau, inner);
            }
            if (upload(&p->tw4row, row) || upload(&p->tw4base, base)) return bail(1);
            if (p->outer > 1) {
                std::vector<cf> o((size_t)p->outer * t), uu((size_t)p->outer * p->n1 * 16);
                for (int k1o = 0; k1o < p->outer; ++k1o) {
                    for (int tau = 0; tau < t; ++tau) o[(size_t)k1o * t + tau] = unit_root((long long)k1o * tau, n_fft);
                    for (int k1 = 0; k1 < p->n1; ++k1)
                        for (int j = 0; j < 16; ++j) {
                            // W_{16 n1}^{k1 j} W_65536^{k1o j} as one angle over 65536 * 16 n1 / gcd ...: in double
                            const double a = -2.0 * M_PI * ((double)((long long)k1 * j % (16ll * p->n1)) / (16.0 * p->n1) +
                                                            (double)((long long)k1o * j % 65536) / 65536.0);
                            uu[((size_t)k1o * p->n1 + k1) * 16 + j] = make_float2((float)cos(a), (float)sin(a));
                        }
                }
                if (upload(&p->tw4o, o) || upload(&p->tw4u, uu)) return bail(1);
            }
            // twiddles in the column passes: only with 256- / 512-point columns (T1 = n1 / 16 threads
            // per column, thread tau holds rows tau + T1 j: factors a g^j, a = W_N^{tau n2}, g = W_N^{T1 n2})
            if (p->n1 >= 256 && p->outer == 1) {
                const int t1 = p->n1 / 16;
                std::vector<cf> a((size_t)t1 * p->n2), g((size_t)4 * p->n2);
                for (int n2 = 0; n2 < p->n2; ++n2) {
                    for (int i = 0; i < 4; ++i) g[(size_t)i * p->n2 + n2] = unit_root(((long long)t1 << i) * n2, n_fft);
                    for (int tau = 0; tau < t1; ++tau) a[(size_t)tau * p->n2 + n2] = unit_root((long long)tau * n2, n_fft);
                }
                if (upload(&p->twa, a) || upload(&p->twg, g)) return bail(1);
                p->tw_col = true;
            }
        }
    }

    // response: upload (if needed), permute to [C][N1][N2], scale by 1/N
    const size_t rbytes = (size_t)n_resp * n_fft * sizeof(cf);
    cf* nat = nullptr;
    if (resp_on_device) {
        nat = (cf*)resp;
    } else {
        if (hipMalloc((void**)&nat, rbytes) != hipSuccess ||
            hipMemcpy(nat, resp, rbytes, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail("bbt_osm_plan_create: uploading the response failed"));
    }
    if (hipMalloc((void**)&p->resp, rbytes) != hipSuccess)
        return bail(fail("bbt_osm_plan_create: hipMalloc(response) failed"));
    // a device-resident response may still be being written on the caller's
    // (non-blocking) stream, which the null stream used below does not wait for
    if (resp_on_device && hipDeviceSynchronize() != hipSuccess)
        return bail(fail("bbt_osm_plan_create: hipDeviceSynchronize failed"));
    hipLaunchKernelGGL(k_permute_resp, dim3((unsigned)((n_fft + 255) / 256), n_resp), dim3(256), 0, 0,
                       nat, p->resp, p->outer, p->n1, (long long)p->n2, 1.0f / (float)n_fft);
    hipError_t e = hipDeviceSynchronize();
    if (!resp_on_device) hipFree(nat);
    if (e != hipSuccess)
        return bail(fail("bbt_osm_plan_create: response permutation failed: %s",
                         hipGetErrorString(e)));
    if (hipMalloc((void**)&p->resp_index, idx.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(p->resp_index, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice) !=
            hipSuccess)
        return bail(fail("bbt_osm_plan_create: resp_index upload failed"));

    // workspace: `lanes` work buffers of `chunk` blocks each, 192 MiB in total.
    // Measured on MI355X (2-pol 2^20 blocks): 2 x 6 blocks is best; a launch of
    // 6 blocks is 1536 row-pass workgroups = exactly two rounds of the 768
    // resident ones, while 2 x 12 (384 MiB) falls out of the 256 MiB Infinity
    // Cache and loses 6 % although every pass alone is faster.
    int lanes = 2;
    if (const char* env = getenv("BBT_OSM_LANES")) lanes = atoi(env);
    lanes = lanes < 1 ? 1 : (lanes > BBT_MAX_LANES ? BBT_MAX_LANES : lanes);
    if (const char* env = getenv("BBT_OSM_TIMING_STRIDE")) p->timing_stride = std::max(1, atoi(env));
    // (plans on the specialised generic kernels pad the rows of the work buffer to whole lines)
    p->n2p = p->rtc ? (p->n2 + 7) / 8 * 8 : p->n2;
    const size_t per_block = p->rtc ? (size_t)p->npair * p->n1 * p->n2p * 16 : (size_t)p->npair * n_fft * 16;
    int chunk = (int)((192u << 20) / per_block / lanes);
    if (const char* env = getenv("BBT_OSM_CHUNK")) chunk = atoi(env);
    if (chunk < 1) chunk = 1;
    if (chunk > BBT_MAX_CHUNK) chunk = BBT_MAX_CHUNK;
    if (p->outer > 1) {                       // grid.y = blocks * pairs * 256 must fit
        while (chunk > 1 && (long long)chunk * p->npair * p->outer > 65535) --chunk;
    }
    while (chunk > 1 && (long long)chunk * p->npair > 65535) --chunk;      // grid.y of the row pass
    if ((double)chunk * per_block * lanes > 16.0 * (1u << 30)) lanes = 1;   // (config 4: 2 x 2 GiB)
    // Short blocks: sixteen descriptors are a small launch (16 x 8192 samples: 32 workgroups per
    // pass), so runs of REGULAR blocks -- equal steps of input and output, the same kept range:
    // every block of a padded task but a re-aligned last one -- go as one descriptor and a hop
    // (OsmChunk::reg_count), as many blocks as the work buffer (the same 192 MiB) holds.
    int cap = (int)std::min<size_t>((192u << 20) / per_block / lanes, 4096);
    while (cap > 1 && (long long)cap * p->npair * p->outer > 65535) --cap;
    if (cap < chunk || getenv("BBT_OSM_CHUNK")) cap = chunk;
    if (single) {
        // `chunk` pairs of blocks fit the work buffer: a chunk of descriptors holds twice as many blocks
        const int pairs = chunk > BBT_MAX_CHUNK / 2 ? BBT_MAX_CHUNK / 2 : chunk;
        cap = std::max(cap, pairs);
        p->work_bytes = per_block * cap;
        chunk = 2 * pairs;
        cap = 2 * cap;
    } else {
        p->work_bytes = per_block * cap;
    }
    if (p->n1 == 1 && !getenv("BBT_OSM_CHUNK")) {
        // one kernel, no work buffer: nothing bounds a launch but the descriptor array
        chunk = BBT_MAX_CHUNK;
        while (chunk > 1 && (long long)chunk * p->npair >= (1ll << 31)) --chunk;
    }
    if (p->n1 == 1) cap = (int)std::min<long long>(1 << 20, ((1ll << 31) - 1) / p->npair);   // nothing to hold
    p->chunk = chunk;
    p->cap = cap;
    p->lanes = lanes;
    if (p->n1 == 1) p->work_bytes = 0;        // one kernel, no work buffer
    if (p->n1 > 1) {
        for (int l = 0; l < p->lanes; ++l) {
            if (hipMalloc((void**)&p->lane_work[l], p->work_bytes) != hipSuccess)
                return bail(fail("bbt_osm_plan_create: hipMalloc(workspace %zu bytes) failed",
                                 p->work_bytes));
            if (p->lanes > 1 &&
                (hipStreamCreateWithFlags(&p->lane_stream[l], hipStreamNonBlocking) != hipSuccess ||
                 hipEventCreateWithFlags(&p->ev_join[l], BBT_EV_ORDER) != hipSuccess))
                return bail(fail("bbt_osm_plan_create: creating the lane streams failed"));
        }
        p->work = p->lane_work[0];
        if (p->lanes > 1 &&
            hipEventCreateWithFlags(&p->ev_fork, BBT_EV_ORDER) != hipSuccess)
            return bail(fail("bbt_osm_plan_create: creating the fork event failed"));
        if (p->lanes > 1) {
            if (hipStreamCreateWithFlags(&p->tail_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&p->ev_tail[0], BBT_EV_ORDER) != hipSuccess ||
                hipEventCreateWithFlags(&p->ev_tail[1], BBT_EV_ORDER) != hipSuccess)
                return bail(fail("bbt_osm_plan_create: creating the tail stream failed"));
        }
    } else {
        p->lanes = 1;
    }
    // (BBT_ASIDE=0, a development switch: one-kernel plans have no stream of their own and run
    // every call on the caller's)
    const char* aside_env = getenv("BBT_ASIDE");
    if (p->lanes == 1 && !(aside_env && !strcmp(aside_env, "0"))) {
        // (no lanes: a deferred call runs on the tail stream, beside what the caller queues next)
        if (hipStreamCreateWithFlags(&p->tail_stream, hipStreamNonBlocking) != hipSuccess ||
            (!p->ev_fork && hipEventCreateWithFlags(&p->ev_fork, BBT_EV_ORDER) != hipSuccess) ||
            hipEventCreateWithFlags(&p->ev_tail[0], BBT_EV_ORDER) != hipSuccess ||
            hipEventCreateWithFlags(&p->ev_tail[1], BBT_EV_ORDER) != hipSuccess)
            return bail(fail("bbt_osm_plan_create: creating the tail stream failed"));
    }
    if (hipEventCreateWithFlags(&p->ev_done, BBT_EV_ORDER) != hipSuccess)
        return bail(fail("bbt_osm_plan_create: creating the completion event failed"));
    *plan = p;
    return 0;
}

int bbt_osm_plan_destroy(bbt_osm_plan* p) {
    if (!p) return 0;
    // (a deferred call may still be running on the plan's own streams -- its caller has only
    // QUEUED the wait for it: nothing the kernels read goes before they are done)
    for (int l = 0; l < BBT_MAX_LANES; ++l)
        if (p->lane_stream[l]) hipStreamSynchronize(p->lane_stream[l]);
    if (p->tail_stream) hipStreamSynchronize(p->tail_stream);
    for (auto e : p->ev) hipEventDestroy(e);
    for (auto e : p->ev_free) hipEventDestroy(e);
    if (p->resp) hipFree(p->resp);
    if (p->resp_index) hipFree(p->resp_index);
    if (p->tw4row) hipFree(p->tw4row);
    if (p->tw4base) hipFree(p->tw4base);
    if (p->tw4o) hipFree(p->tw4o);
    if (p->tw4u) hipFree(p->tw4u);
    if (p->twa) hipFree(p->twa);
    if (p->twg) hipFree(p->twg);
    if (p->tlo) hipFree(p->tlo);
    if (p->thi) hipFree(p->thi);
    if (p->tws) hipFree(p->tws);
    for (int l = 0; l < BBT_MAX_LANES; ++l) {
        if (p->lane_stream[l]) {
            hipStreamSynchronize(p->lane_stream[l]);
            hipStreamDestroy(p->lane_stream[l]);
        }
        if (p->ev_join[l]) hipEventDestroy(p->ev_join[l]);
        if (p->lane_work[l]) hipFree(p->lane_work[l]);
    }
    if (p->tail_stream) {
        hipStreamSynchronize(p->tail_stream);
        hipStreamDestroy(p->tail_stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (p->ev_tail[i]) hipEventDestroy(p->ev_tail[i]);
        if (p->seam_buf[i]) hipFree(p->seam_buf[i]);
    }
    if (p->ev_fork) hipEventDestroy(p->ev_fork);
    if (p->ev_done) hipEventDestroy(p->ev_done);
    delete p;
    return 0;
}

int bbt_osm_plan_info(const bbt_osm_plan* p, int64_t* workspace_bytes, int* chunk_blocks, int* n1,
                      int* n2) {
    ARG_TRY(p, "bbt_osm_plan_info: null plan");
    if (workspace_bytes) *workspace_bytes = (int64_t)p->work_bytes * p->lanes;
    if (chunk_blocks) *chunk_blocks = p->chunk;
    if (n1) *n1 = p->n1;
    if (n2) *n2 = p->n2;
    return 0;
}

int bbt_osm_plan_defer(bbt_osm_plan* p, bbt_event done) {
    ARG_TRY(p, "bbt_osm_plan_defer: null plan");
    std::lock_guard<std::mutex> lock(p->mu);
    p->defer_ev = (hipEvent_t)done;
    return 0;
}

int bbt_osm_plan_fusable(const bbt_osm_plan* p, int n_chan) {
    // can bbt_osm_execute_channelized take Channelize(n_chan) into the row pass?
    if (!p || p->generic || (p->n1 == 1 && p->outer == 1)) return 0;
    if (p->single)              // one stream: blocks side by side; 256 channels and up, no detection
        return fft_len_ok(n_chan) && n_chan <= p->n2 && p->n2 % n_chan == 0;
    if (n_chan == 16 || n_chan == 32 || n_chan == 64 || n_chan == 128)       // few channels: after an
        return p->n1 >= 256 || p->outer == 256;                             // exchange in the row pass
    if (n_chan == 2 || n_chan == 4 || n_chan == 8)       // very few: lane butterflies in the row pass,
        return p->outer == 1 && (p->n1 == 16 || p->n1 == 256);           // two-level plans (2^13 ... 2^20 samples)
    return fft_len_ok(n_chan) && n_chan <= p->n2 && p->n2 % n_chan == 0;
}

int bbt_osm_execute(bbt_osm_plan* p, const void* in_dev, void* out_dev, int64_t n_blocks,
                    const int64_t* in_off, const int64_t* out_off, const int32_t* valid_start,
                    const int32_t* valid_count, bbt_stream stream) {
    DeferTake take(p, (hipStream_t)stream);
    ARG_TRY(p && in_dev && out_dev, "bbt_osm_execute: null argument");
    if (osm_check_blocks(p, "bbt_osm_execute", n_blocks, in_off, out_off, valid_start, valid_count))
        return 1;
    SpecOut so = {};
    PlanCall call(p, take, n_blocks);
    return osm_run_all(p, (const float2*)in_dev, (float2*)out_dev, n_blocks, so, call,
                       [&](OsmBlock& blk, int64_t b) {
                           blk.in_off = in_off[b];
                           blk.out_off = out_off[b];
                           blk.valid_start = valid_start[b];
                           blk.valid_count = valid_count[b];
                           blk.shift = 0;
                           blk.index = (int)b;
                       });
}

int bbt_osm_execute_flat(bbt_osm_plan* p, const void* in_dev, void* out_dev, int64_t n_blocks,
                         const int64_t* in_off, const int64_t* out_elem_off, const int32_t* valid_start,
                         int32_t first_elem, const int32_t* valid_elems, bbt_stream stream) {
    const char* who = "bbt_osm_execute_flat";
    DeferTake take(p, (hipStream_t)stream);
    ARG_TRY(p && in_dev && out_dev, "%s: null argument", who);
    ARG_TRY(!p->generic && !p->single && p->n1 == 1 && p->outer == 1,
            "%s: only for power-of-two blocks of at most 4096 samples with an even stream count", who);
    ARG_TRY(n_blocks >= 0 && (n_blocks == 0 || (in_off && out_elem_off && valid_start && valid_elems)),
            "%s: bad descriptors", who);
    ARG_TRY(first_elem >= 0 && first_elem < p->S && first_elem % 2 == 0,
            "%s: first_elem=%d must be an even element of a row of %d", who, first_elem, p->S);
    for (int64_t b = 0; b < n_blocks; ++b)
        ARG_TRY(in_off[b] >= 0 && out_elem_off[b] >= 0 && out_elem_off[b] % 2 == 0 && valid_start[b] >= 0 &&
                    valid_elems[b] >= 0 && valid_elems[b] % 2 == 0 &&
                    (int64_t)valid_start[b] * p->S + first_elem + valid_elems[b] <= p->n * p->S,
                "%s: block %lld keeps elements outside the block", who, (long long)b);
    SpecOut so = {};
    PlanCall call(p, take, n_blocks);
    return osm_run_all(p, (const float2*)in_dev, (float2*)out_dev, n_blocks, so, call,
                       [&](OsmBlock& blk, int64_t b) {
                           blk.in_off = in_off[b];
                           blk.out_off = out_elem_off[b];
                           blk.valid_start = valid_start[b];
                           blk.valid_count = valid_elems[b];
                           blk.flat = 1;
                           blk.flat_sub = first_elem;
                       });
}

int bbt_osm_plan_set_layout(bbt_osm_plan* p, int64_t in_plane, int64_t out_plane) {
    ARG_TRY(p, "bbt_osm_plan_set_layout: null plan");
    ARG_TRY(in_plane >= 0 && out_plane >= 0, "bbt_osm_plan_set_layout: negative plane length");
    const bool pairs = !p->generic && !p->single && p->outer == 1 && p->S % 2 == 0;
    ARG_TRY(in_plane == 0 || (pairs && p->n1 == 256),
            "bbt_osm_plan_set_layout: pair-planar input needs a two-level plan with 256-point columns "
            "(blocks of 2^17 to 2^20 samples) of an even number of streams");
    ARG_TRY(out_plane == 0 || (pairs && p->n1 == 1),
            "bbt_osm_plan_set_layout: pair-planar output needs a one-kernel plan (blocks of at most "
            "4096 samples) of an even number of streams");
    std::lock_guard<std::mutex> lock(p->mu);
    p->in_plane = in_plane;
    p->out_plane = out_plane;
    return 0;
}

// Channelize(overlap-save task) as one call; with det_step > 0 the spectra are
// detected and integrated instead of stored (out_dev = float32 bins).
static int osm_channelized(bbt_osm_plan* p, const char* who, const void* in_dev, void* out_dev,
                           int64_t n_blocks, const int64_t* in_off, const int64_t* out_off,
                           const int32_t* valid_start, const int32_t* valid_count, int n_chan,
                           int64_t first_spectrum, int64_t n_spectra, int det_step, int det_mode,
                           float det_scale, hipStream_t st) {
    DeferTake take(p, st);
    ARG_TRY(p && in_dev && out_dev, "%s: null argument", who);
    ARG_TRY(!p->generic, "%s: the fused channelizer needs a power-of-two block length (got %lld)",
            who, (long long)p->n);
    ARG_TRY(p->n1 > 1 || p->outer > 1, "%s: block length %lld is too short to fuse", who,
            (long long)p->n);
    ARG_TRY(bbt_osm_plan_fusable(p, n_chan),
            "%s: n_chan=%d must be a power of two in [256, %d] (or 16..128 for blocks with 256 "
            "columns or of three levels, 2..8 for two-level blocks of up to 2^20 samples)", who, n_chan, p->n2);
    const bool small = n_chan < 256;
    ARG_TRY(!(small && det_step > 0), "%s: fused detection needs n_chan >= 256", who);
    ARG_TRY(!(p->single && det_step > 0), "%s: one-stream plans have no fused detection", who);
    ARG_TRY(first_spectrum >= 0 && n_spectra >= 0, "%s: bad spectrum range", who);
    if (osm_check_blocks(p, who, n_blocks, in_off, out_off, valid_start, valid_count)) return 1;
    for (int64_t b = 0; b < n_blocks; ++b)
        ARG_TRY(valid_count[b] >= n_chan, "%s: block %lld keeps %d samples < n_chan", who,
                (long long)b, valid_count[b]);
    if (n_blocks == 0 || n_spectra == 0) return 0;
    PlanCall call(p, take, n_blocks);
    FftTables tabc;
    GenGeo gsmall = {};
    cf* wsmall = nullptr;
    if (small ? (!factor_7smooth(n_chan, &gsmall) || get_gen_table(&gsmall, &wsmall))
              : get_tables(n_chan, &tabc))
        return 1;
    // seam slots and jobs.  The slots are filled by the lanes and read by the seam pass at the
    // end of the call; a deferred call takes the turn the call before it did not use, after the
    // seam pass of the last call on that turn (two calls back, long done: no stall).
    const int turn = call.deferred ? (p->seam_turn ^= 1) : p->seam_turn;
    if (p->ev_tail_set[turn]) {
        HIP_TRY(hipStreamWaitEvent(st, p->ev_tail[turn], 0));
        if (!call.deferred) p->ev_tail_set[turn] = false;
    }
    const size_t need = (size_t)n_blocks * 2 * p->npair * n_chan * 16;
    if (need > p->seam_bytes[turn]) {
        if (p->seam_buf[turn]) HIP_TRY(hipFree(p->seam_buf[turn]));      // (waits for the device)
        p->seam_buf[turn] = nullptr;
        p->seam_bytes[turn] = 0;
        HIP_TRY(hipMalloc((void**)&p->seam_buf[turn], need));
        p->seam_bytes[turn] = need;
    }
    p->seam = p->seam_buf[turn];
    std::vector<SeamJob> jobs;
    for (int64_t b = 0; b + 1 < n_blocks; ++b) {
        const int64_t seam_pos = out_off[b] + valid_count[b];
        if (out_off[b + 1] != seam_pos || seam_pos % n_chan == 0) continue;   // not adjacent / aligned
        const int64_t s = seam_pos / n_chan;
        if (s < first_spectrum || s >= first_spectrum + n_spectra) continue;
        SeamJob j;
        j.spectrum = s - first_spectrum;
        j.first_block = (int)b;
        j.split = (int)(seam_pos - s * n_chan);
        jobs.push_back(j);
    }
    SpecOut so = {};
    so.seam = p->seam;
    so.s_base = first_spectrum;
    so.n_out = n_spectra;
    so.n_chan = n_chan;
    so.lg_chan = 0;
    while ((1 << so.lg_chan) < n_chan) ++so.lg_chan;
    so.n_fft = (int)p->n;
    so.small_l = small ? n_chan / 16 : 0;             // (0 below 16 channels)
    so.tiny_lg = n_chan < 16 ? so.lg_chan : 0;
    so.small_row = p->n2;
    if (det_step > 0) {
        so.det = (float*)out_dev;
        so.det_step = det_step;
        so.det_mode = det_mode;
        so.det_scale = det_scale;
    }
    if (osm_run_all(p, (const float2*)in_dev, (float2*)out_dev, n_blocks, so, call,
                    [&](OsmBlock& blk, int64_t b) {
                        blk.in_off = in_off[b];
                        blk.out_off = out_off[b];
                        blk.valid_start = valid_start[b];
                        blk.valid_count = valid_count[b];
                        // circular shift that puts spectrum boundaries on multiples of n_chan
                        int64_t o = ((int64_t)valid_start[b] - out_off[b]) % n_chan;
                        if (o < 0) o += n_chan;
                        blk.shift = (int)o;
                        blk.index = (int)b;
                    }))
        return 1;
    hipStream_t fin = call.fin;        // (a deferred call: the plan's tail stream, after the lanes)
    if (!jobs.empty() && small) {
        const size_t lds = (size_t)2 * n_chan * sizeof(f4);
        for (size_t j0 = 0; j0 < jobs.size(); j0 += BBT_SEAM_JOBS_PER_LAUNCH) {
            SeamJobs batch;
            const size_t n = std::min(jobs.size() - j0, (size_t)BBT_SEAM_JOBS_PER_LAUNCH);
            for (size_t i = 0; i < n; ++i) batch.j[i] = jobs[j0 + i];
            hipLaunchKernelGGL(k_seam_fix_gen, dim3((unsigned)n, p->npair), dim3(gen_threads(2 * n_chan)),
                               lds, fin, p->seam, (float2*)out_dev, batch, p->S, p->npair, gsmall, wsmall, so);
        }
        HIP_TRY(hipGetLastError());
    } else if (!jobs.empty()) {
        switch (n_chan) {
            case 256: launch_seam_fix<256>(p, (float2*)out_dev, jobs, tabc, so, fin); break;
            case 512: launch_seam_fix<512>(p, (float2*)out_dev, jobs, tabc, so, fin); break;
            case 1024: launch_seam_fix<1024>(p, (float2*)out_dev, jobs, tabc, so, fin); break;
            case 2048: launch_seam_fix<2048>(p, (float2*)out_dev, jobs, tabc, so, fin); break;
            case 4096: launch_seam_fix<4096>(p, (float2*)out_dev, jobs, tabc, so, fin); break;
        }
        HIP_TRY(hipGetLastError());
    }
    if (call.deferred) {
        HIP_TRY(hipEventRecord(p->ev_tail[turn], fin));
        p->ev_tail_set[turn] = true;
    }
    return 0;
}

int bbt_osm_execute_channelized(bbt_osm_plan* p, const void* in_dev, void* out_dev,
                                int64_t n_blocks, const int64_t* in_off, const int64_t* out_off,
                                const int32_t* valid_start, const int32_t* valid_count, int n_chan,
                                int64_t first_spectrum, int64_t n_spectra, bbt_stream stream) {
    return osm_channelized(p, "bbt_osm_execute_channelized", in_dev, out_dev, n_blocks, in_off,
                           out_off, valid_start, valid_count, n_chan, first_spectrum, n_spectra, 0,
                           0, 0.f, (hipStream_t)stream);
}

int bbt_osm_detect_bins_max(const bbt_osm_plan* p, int n_chan, int step) {
    // integration bins one column-pass workgroup touches: its 256 rows hold
    // every (n2 / n_chan)-th of 256 * n2 / n_chan consecutive spectra
    if (!p || n_chan <= 0 || step <= 0) return -1;
    const int64_t span = 255ll * (p->n2 / n_chan);
    return (int)(span / step) + 2;
}

int bbt_osm_execute_channelized_detect(bbt_osm_plan* p, const void* in_dev, void* out_dev,
                                       int64_t n_blocks, const int64_t* in_off,
                                       const int64_t* out_off, const int32_t* valid_start,
                                       const int32_t* valid_count, int n_chan,
                                       int64_t first_spectrum, int64_t n_bins, int step, int mode,
                                       int average, bbt_stream stream) {
    const char* who = "bbt_osm_execute_channelized_detect";
    DeferTake take(p, (hipStream_t)stream);
    ARG_TRY(p && out_dev, "%s: null argument", who);
    ARG_TRY(p->n1 == 256 && p->outer == 1,
            "%s: fused detection needs a two-level transform with 256 columns (block length 2^16..2^20)",
            who);
    ARG_TRY(step >= 1 && n_bins >= 0 && (mode == 0 || mode == 1), "%s: bad step/bins/mode", who);
    ARG_TRY(n_bins * (int64_t)step < (1ll << 40), "%s: too many spectra", who);
    // (step 1 -- Square / Power without integration -- stores every power itself: no bins, no zeroing)
    ARG_TRY(step == 1 || bbt_osm_detect_bins_max(p, n_chan, step) <= BBT_DET_MAX_BINS,
            "%s: step=%d is too short for %d channels on rows of %d (a workgroup would touch more "
            "than %d bins)", who, step, n_chan, p->n2, BBT_DET_MAX_BINS);
    const size_t out_bytes = (size_t)n_bins * n_chan * p->npair * (mode ? 4 : 2) * sizeof(float);
    if (out_bytes && step > 1) HIP_TRY(hipMemsetAsync(out_dev, 0, out_bytes, (hipStream_t)stream));
    take.give_back(p);
    return osm_channelized(p, who, in_dev, out_dev, n_blocks, in_off, out_off, valid_start,
                           valid_count, n_chan, first_spectrum, n_bins * step, step, mode,
                           average ? 1.0f / (float)step : 1.0f, (hipStream_t)stream);
}

int bbt_osm_execute_regular(bbt_osm_plan* p, const void* in_dev, void* out_dev, int64_t n_blocks,
                            int64_t in_off0, int64_t out_off0, int64_t hop, int32_t valid_start,
                            bbt_stream stream) {
    DeferTake take(p, (hipStream_t)stream);
    ARG_TRY(p, "bbt_osm_execute_regular: null plan");
    ARG_TRY(n_blocks >= 0 && hop > 0 && hop <= p->n, "bbt_osm_execute_regular: bad hop %lld",
            (long long)hop);
    if (!p->generic && p->n1 == 1 && p->outer == 1 && n_blocks > 0) {
        // one-kernel plans: the whole run as regular descriptors, up to 2^20 blocks per launch
        ARG_TRY(in_dev && out_dev && in_off0 >= 0 && out_off0 >= 0 && valid_start >= 0 &&
                    valid_start + hop <= p->n,
                "bbt_osm_execute_regular: blocks keep [%d, %lld) outside [0, %lld)", valid_start,
                (long long)(valid_start + hop), (long long)p->n);
        SpecOut so = {};
        PlanCall call(p, take, 0);         // (one-kernel plans have no lanes)
        const int64_t per_launch = std::min<int64_t>(1 << 20, ((1ll << 31) - 1) / p->npair);
        for (int64_t b0 = 0; b0 < n_blocks; b0 += per_launch) {
            OsmChunk ch = {};
            ch.reg_count = (int)std::min(per_launch, n_blocks - b0);
            ch.reg_hop = hop;
            ch.nblk = 1;
            ch.b[0].in_off = in_off0 + b0 * hop;
            ch.b[0].out_off = out_off0 + b0 * hop;
            ch.b[0].valid_start = valid_start;
            ch.b[0].valid_count = (int)hop;
            if (osm_run_chunk(p, (const float2*)in_dev, (float2*)out_dev, ch, so, nullptr, call.run)) return 1;
        }
        return 0;
    }
    std::vector<int64_t> io(n_blocks), oo(n_blocks);
    std::vector<int32_t> vs(n_blocks, valid_start), vc(n_blocks, (int32_t)hop);
    for (int64_t b = 0; b < n_blocks; ++b) {
        io[b] = in_off0 + b * hop;
        oo[b] = out_off0 + b * hop;
    }
    take.give_back(p);
    return bbt_osm_execute(p, in_dev, out_dev, n_blocks, io.data(), oo.data(), vs.data(),
                           vc.data(), stream);
}

int bbt_osm_timing_enable(bbt_osm_plan* p, int enable) {
    ARG_TRY(p, "bbt_osm_timing_enable: null plan");
    std::lock_guard<std::mutex> lock(p->mu);
    if (osm_flush_timing(p)) return 1;
    p->timing = enable != 0;
    p->timing_isolated = enable == 2;
    p->acc_ms[0] = p->acc_ms[1] = p->acc_ms[2] = 0;
    p->launches = 0;
    p->pass_launches[0] = p->pass_launches[1] = p->pass_launches[2] = 0;
    p->pass_blocks[0] = p->pass_blocks[1] = p->pass_blocks[2] = 0;
    return 0;
}

int bbt_osm_timing_read(bbt_osm_plan* p, double ms[3], int64_t* launches) {
    ARG_TRY(p && ms, "bbt_osm_timing_read: null argument");
    std::lock_guard<std::mutex> lock(p->mu);
    if (osm_flush_timing(p)) return 1;
    for (int k = 0; k < 3; ++k) ms[k] = p->acc_ms[k];
    if (launches) *launches = p->pass_launches[0];
    return 0;
}

int bbt_osm_timing_read_passes(bbt_osm_plan* p, double ms[3], int64_t launches[3], int64_t blocks[3]) {
    ARG_TRY(p && ms && launches && blocks, "bbt_osm_timing_read_passes: null argument");
    std::lock_guard<std::mutex> lock(p->mu);
    if (osm_flush_timing(p)) return 1;
    for (int k = 0; k < 3; ++k) {
        ms[k] = p->acc_ms[k];
        launches[k] = p->pass_launches[k];
        blocks[k] = p->pass_blocks[k];
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// channelizer
struct bbt_chan_plan {
    int n = 0, S = 0, npair = 0, dir = -1;
    bool split_real = false;    // direction -2: one stream z = a + i b, output = half spectra of a and b
    FftTables tab;
    cf* wroot = nullptr;
    // 8192 / 16384 channels of stream pairs (fft_big.hpp)
    cf* big = nullptr;
    // channel counts that are not powers of two (gen_kernels.hpp)
    bool generic = false;
    GenGeo g = {};
    cf* wn = nullptr;
    int ct = 1;                 // stream pairs per workgroup tile
    // ... on the run-time specialised engine (gen2_kernels.hpp; null: not in use)
    G2Plan q;                   // q.ct columns per workgroup: cp stream pairs of q.ct / cp transforms
    int cp = 1;
    cf* qw = nullptr;
    hipFunction_t k2 = nullptr;
};

template <int N, int SIGN>
static void launch_short(const bbt_chan_plan* p, const float2* in, float2* out, int64_t n_fft,
                         float scale, hipStream_t st) {
    constexpr int R = (N <= 16) ? 1 : N / 16;
    constexpr int FPW = 256 / R;
    if constexpr (N <= 16) {
        if (p->S == 2) {                  // one pair: whole lines in and out, the transposition in LDS
            constexpr int T = 256;          // (128 for N = 16 -- four workgroups per CU instead of two -- measured no faster)
            constexpr size_t lds = T * (N + 1) * 16;
            if (ensure_dyn_lds((const void*)k_fft_tiny<N, SIGN, T>, lds) == 0) {
                hipLaunchKernelGGL((k_fft_tiny<N, SIGN, T>), dim3((unsigned)((n_fft + T - 1) / T)), dim3(T), lds, st,
                                   in, out, (long long)n_fft, scale);
                return;
            }
        }
    }
#define BBT_SHORT(PP_)                                                                         \
    {                                                                                          \
        constexpr int FPB = FPW / PP_;                                                         \
        const unsigned gx = (unsigned)((n_fft + FPB - 1) / FPB);                               \
        hipLaunchKernelGGL((k_fft_short<N, SIGN, PP_>), dim3(gx * (p->npair / PP_)), dim3(256), 0, \
                           st, in, out, (long long)n_fft, p->S, scale, p->wroot);              \
    }
    if (p->npair % 8 == 0) BBT_SHORT(8)
    else if (p->npair % 4 == 0) BBT_SHORT(4)
    else if (p->npair % 2 == 0) BBT_SHORT(2)
    else BBT_SHORT(1)
#undef BBT_SHORT
}

template <int N, int SIGN>
static void launch_rows(const bbt_chan_plan* p, const float2* in, float2* out, int64_t n_fft,
                        float scale, hipStream_t st) {
    constexpr int FPW = (N >= 1024) ? (4096 / N >= 4 ? 4 : 4096 / N) : (N == 512 ? 8 : 16);
    if (p->S == 1) {            // one stream: two consecutive transforms side by side
        const unsigned gx = (unsigned)(((n_fft + 1) / 2 + FPW - 1) / FPW);
        if (p->split_real) {    // ... of two real streams: half spectra out (forward) / in (inverse)
            hipLaunchKernelGGL((k_fft_rows<N, SIGN, FPW, true, true>), dim3(gx), dim3(FPW * N / 16), 0, st,
                               in, out, (long long)n_fft, 1, scale, p->tab.tw0, p->tab.tw1);
            return;
        }
        hipLaunchKernelGGL((k_fft_rows<N, SIGN, FPW, true>), dim3(gx), dim3(FPW * N / 16), 0, st, in, out,
                           (long long)n_fft, 1, scale, p->tab.tw0, p->tab.tw1);
        return;
    }
    // several pairs: the lanes over PP pairs first (k_fft_rows_pp) -- the largest of 8, 4, 2 that fits a
    // workgroup (1024 threads, 160 KiB of exchange area: 8 up to 2048 points, 4 at 4096) and that divides
    // the number of pairs.  Measured, Channelize(256 / 1024 / 4096) in G stream-samples/s: 16 streams
    // 174 / 150 / 165 -> 361 / 294 / 252, 2048 streams 87 / 114 / 124 -> 355 / 274 / 168 (two streams: 383 / 374 / 318)
    // with 8 / 4 / 2 pairs (round 4: as many as leave two workgroups per CU); whole 128-byte lines are worth
    // more than the second workgroup (round 5, 8 / 8 / 4 pairs): 16 streams 345 / 340 / 280, 128 streams
    // 357 / 343 / 279, 2048 streams 344 / 333 / 245.
    constexpr int CAP = N <= 2048 ? 8 : 4;
    const int cap = getenv("BBT_ROWS_CAP") ? atoi(getenv("BBT_ROWS_CAP")) : CAP;      // (dev)
#define BBT_ROWS_PP(PP_)                                                                                         \
    if ((PP_ * N / 16 <= 1024 && FftGeo<N>::LDS_ELEMS * sizeof(v2) * PP_ <= 160 * 1024) && PP_ <= cap &&        \
        p->npair % PP_ == 0 && n_fft * (p->npair / PP_) < (1ll << 31)) {                                        \
        constexpr int Q = (PP_ * N / 16 <= 1024 && FftGeo<N>::LDS_ELEMS * sizeof(v2) * PP_ <= 160 * 1024) ? PP_ : 1; \
        constexpr size_t lds = FftGeo<N>::LDS_ELEMS * sizeof(v2) * Q;                                            \
        if (ensure_dyn_lds((const void*)k_fft_rows_pp<N, SIGN, Q>, lds) == 0) {                                  \
            hipLaunchKernelGGL((k_fft_rows_pp<N, SIGN, Q>), dim3((unsigned)(n_fft * (p->npair / Q))),            \
                               dim3(Q * N / 16), lds, st, in, out, (long long)n_fft, p->S, scale, p->tab.tw0,    \
                               p->tab.tw1);                                                                      \
            return;                                                                                              \
        }                                                                                                        \
    }
    if (!p->split_real && p->S > 2 && !getenv("BBT_ROWS_NO_PP")) {
        BBT_ROWS_PP(8) BBT_ROWS_PP(4) BBT_ROWS_PP(2)
    }
#undef BBT_ROWS_PP
    const unsigned gx = (unsigned)((n_fft + FPW - 1) / FPW);
    if (p->split_real) {            // every stream z = a + i b of two real streams: half spectra out / in
        hipLaunchKernelGGL((k_fft_rows<N, SIGN, FPW, false, true>), dim3(gx * p->npair), dim3(FPW * N / 16), 0,
                           st, in, out, (long long)n_fft, p->S, scale, p->tab.tw0, p->tab.tw1);
        return;
    }
    hipLaunchKernelGGL((k_fft_rows<N, SIGN, FPW>), dim3(gx * p->npair), dim3(FPW * N / 16), 0, st, in,
                       out, (long long)n_fft, p->S, scale, p->tab.tw0, p->tab.tw1);
}

template <int SIGN>
static int chan_dispatch(const bbt_chan_plan* p, const float2* in, float2* out, int64_t n_fft,
                         float scale, hipStream_t st) {
    switch (p->n) {
        case 2: launch_short<2, SIGN>(p, in, out, n_fft, scale, st); break;
        case 4: launch_short<4, SIGN>(p, in, out, n_fft, scale, st); break;
        case 8: launch_short<8, SIGN>(p, in, out, n_fft, scale, st); break;
        case 16: launch_short<16, SIGN>(p, in, out, n_fft, scale, st); break;
        case 32: launch_short<32, SIGN>(p, in, out, n_fft, scale, st); break;
        case 64: launch_short<64, SIGN>(p, in, out, n_fft, scale, st); break;
        case 128: launch_short<128, SIGN>(p, in, out, n_fft, scale, st); break;
        case 256: launch_rows<256, SIGN>(p, in, out, n_fft, scale, st); break;
        case 512: launch_rows<512, SIGN>(p, in, out, n_fft, scale, st); break;
        case 1024: launch_rows<1024, SIGN>(p, in, out, n_fft, scale, st); break;
        case 2048: launch_rows<2048, SIGN>(p, in, out, n_fft, scale, st); break;
        case 4096: launch_rows<4096, SIGN>(p, in, out, n_fft, scale, st); break;
        default: return fail("channelize: unsupported n_chan %d", p->n);
    }
    return 0;
}

extern "C" {

}  // extern "C"

// Scratch device memory of the plans that time their candidates (chan_pick, pfb_pick): ONE buffer
// per device, allocated when the first such plan is made, grown when a plan needs more, and kept
// -- mapping and unmapping a GiB per plan preempts every queue of the process and took 0.3-2 s
// each after a few dozen plans.  1 GiB each way when the card has it to spare (well past the 256
// MiB memory-side cache: at 256 MiB each way 1536 channels ranked wrongly), at most an eighth of
// what is free; BBT_TUNE_KEEP=0 frees it after every plan.
struct TuneScratch {
    std::mutex mu;
    std::map<int, std::pair<void*, size_t>> buf;         // device -> (pointer, bytes)
};
static TuneScratch& tune_scratch() {
    static TuneScratch* t = new TuneScratch;              // (never destroyed: the runtime may be gone by then)
    return *t;
}
// A buffer of at least `least` bytes (as much as `want` if the card has it to spare); null if not to be had.
// The caller holds it until tune_release(): plans are made one at a time (the mutex is held).
static void* tune_acquire(size_t least, size_t want, size_t* got) {
    TuneScratch& t = tune_scratch();
    t.mu.lock();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        t.mu.unlock();
        return nullptr;
    }
    auto& slot = t.buf[dev];
    if (slot.second < least) {
        if (slot.first) hipFree(slot.first);
        slot = {nullptr, 0};
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        size_t bytes = std::max(least, std::min(want, free_b / 4));
        void* ptr = nullptr;
        if (bytes > free_b / 2 || hipMalloc(&ptr, bytes) != hipSuccess) {
            hipGetLastError();
            t.mu.unlock();
            return nullptr;
        }
        slot = {ptr, bytes};
    }
    *got = slot.second;
    return slot.first;
}
static void tune_release() {
    TuneScratch& t = tune_scratch();
    if (getenv("BBT_TUNE_KEEP") && !strcmp(getenv("BBT_TUNE_KEEP"), "0")) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && t.buf[dev].first) {
            hipFree(t.buf[dev].first);
            t.buf[dev] = {nullptr, 0};
        }
    }
    t.mu.unlock();
}
extern "C" int bbt_tune_scratch(int release, int64_t* bytes) {
    TuneScratch& t = tune_scratch();
    std::lock_guard<std::mutex> lock(t.mu);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    auto& slot = t.buf[dev];
    if (bytes) *bytes = (int64_t)slot.second;
    if (release && slot.first) {
        HIP_TRY(hipFree(slot.first));
        slot = {nullptr, 0};
    }
    return 0;
}

// One candidate of a generic-length channelizer plan on the compiled kernels: columns of a
// workgroup = cp neighbouring stream pairs (up to 8: 128-byte pieces of a complete sample, as
// many as divide the pair count) x consecutive transforms while they still fit ONE wave (a
// transform of 100 threads gains nothing from a neighbour: 1000 channels 160 G alone, 128 G in
// twos).  0 = ok (k2 set), 1 = not available (g_err says why).
// `wide` (many streams): as many pairs as fit a workgroup of 1024 threads and 144 KiB instead of 256
// threads -- whole 128-byte lines at one workgroup per CU, as k_fft_rows_pp.
static int chan_candidate(bbt_chan_plan* p, int n_chan, int direction, int pmax, bool wide = false) {
    G2Plan probe;
    if (!g2_plan(n_chan, 1, &probe, pmax)) return fail("no stage list for %d", n_chan);
    int cp = 1;
    while (cp < 8 && p->npair % (2 * cp) == 0 &&
           (wide ? probe.tj * 2 * cp <= 1024 && (int64_t)n_chan * 2 * cp * 8 <= 144 * 1024 : probe.tj * 2 * cp <= 256))
        cp *= 2;
    int ct = cp;
    const int ct_cap = getenv("BBT_G2_CHAN_CT") ? atoi(getenv("BBT_G2_CHAN_CT")) : 256;      // (dev)
    while (2 * ct <= ct_cap && probe.tj * 2 * ct <= 64 && (int64_t)n_chan * 2 * ct * 8 <= 64 * 1024) ct *= 2;
    G2Plan q;
    if (!g2_plan(n_chan, ct, &q, pmax) || q.threads() > 1024) return fail("no plan for %d x %d columns", n_chan, ct);
    const std::string src = "#include \"gen2_kernels.hpp\"\n" + g2_trait_source("GA", q) +
                            "BBT_G2_KERNEL_FFT_ROWS(k_rows, GA, " + (direction < 0 ? "-1" : "+1") +
                            (q.threads() >= 448 ? ", 4)\n" : ", 0)\n");
    hipFunction_t k = nullptr;
    cf* w = nullptr;
    if (g2_build(src, {"k_rows"}, &k) || get_g2_table(q, &w)) return 1;
    p->q = q;
    p->cp = cp;
    p->k2 = k;
    p->qw = w;
    return 0;
}
// Choose among the candidates (and the general kernel) by timing them; 0 = ok (p->k2 null: the
// general kernel stays), 1 = error (only in `require` mode).
static int chan_pick(bbt_chan_plan* p, int n_chan, int direction) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, int> chosen;       // (device, n, streams, direction) -> points (+ 100: wide), 0 = general
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const auto key = std::make_tuple(dev, n_chan, p->S, direction);
    const bool tune = !(getenv("BBT_G2_TUNE") && !strcmp(getenv("BBT_G2_TUNE"), "0"));
    int pick = -1;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = chosen.find(key);
        if (it != chosen.end()) pick = it->second;
    }
    if (pick < 0 && !tune) pick = g2_pmax(BBT_G2_KIND_CHAN) + (getenv("BBT_G2_CHAN_WIDE") ? 100 : 0);      // (wide: tests)
    if (pick == 0) return 0;                                           // (the general kernel won before)
    if (pick > 0) {
        if (!chan_candidate(p, n_chan, direction, pick % 100, pick >= 100)) return 0;
        if (rtc_mode() == 2) return 1;
        g2_warn_once("bbt_chan_plan_create");
        p->k2 = nullptr;
        return 0;
    }
    // time the candidates on scratch memory (tune_acquire), from HBM as in use, on the null stream (a
    // stream of its own would be one more hardware queue made and destroyed per plan)
    const size_t per = (size_t)n_chan * p->S * sizeof(cf);
    size_t got = 0;
    char* base = (char*)tune_acquire(16 * per, (size_t)2 << 30, &got);
    const int64_t ns = (int64_t)(got / (2 * per));
    void *a = base, *b = base ? base + ns * per : nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ready = base && hipMemsetAsync(a, 0, ns * per, nullptr) == hipSuccess &&
                 hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
                 hipDeviceSynchronize() == hipSuccess;
    auto time_current = [&]() -> float {
        float best = 1e30f;
        if (bbt_chan_execute(p, a, b, ns, st)) return best;           // (warm-up: code object upload, tables)
        for (int r = 0; r < 2; ++r) {
            float ms = 0.f;
            if (hipEventRecord(e0, st) != hipSuccess || bbt_chan_execute(p, a, b, ns, st) ||
                hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                return 1e30f;
            best = std::min(best, ms);
        }
        return best;
    };
    float best_ms = 1e30f;
    int any = 0;
    if (ready) {
        p->k2 = nullptr;
        best_ms = time_current();                                       // the general kernel
        pick = 0;
        bbt_chan_plan best_plan = *p;
        std::vector<G2Plan> seen;                                       // (plans already timed: stages, threads, columns)
        for (int code : {10, 16, 20, 110, 116, 120}) {
            const int pmax = code % 100;
            const bool wide = code >= 100;
            if (wide && p->npair < 4) continue;
            if (chan_candidate(p, n_chan, direction, pmax, wide)) continue;
            ++any;
            bool same = false;
            for (const G2Plan& o : seen) {
                bool eq = o.nfac == p->q.nfac && o.tj == p->q.tj && o.ct == p->q.ct;
                for (int i = 0; eq && i < o.nfac; ++i) eq = o.fac[i] == p->q.fac[i];
                same = same || eq;
            }
            if (same) continue;
            seen.push_back(p->q);
            const float ms = time_current();
            if (getenv("BBT_RTC_VERBOSE"))
                fprintf(stderr, "bbt: Channelize(%d) x %d streams, %d points per thread, %d pairs x %d transforms per workgroup: "
                                "%.3f ms (general %.3f)\n", n_chan, p->S, pmax, p->cp, p->q.ct / p->cp, ms,
                        best_plan.k2 ? -1.f : best_ms);
            if (ms < best_ms) {
                best_ms = ms;
                pick = code;
                best_plan = *p;
            }
        }
        *p = best_plan;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (base) tune_release();
    if (!ready) {                                                       // (no scratch memory: the rule, untimed)
        hipGetLastError();
        if (chan_candidate(p, n_chan, direction, g2_pmax(BBT_G2_KIND_CHAN))) {
            if (rtc_mode() == 2) return 1;
            g2_warn_once("bbt_chan_plan_create");
            p->k2 = nullptr;
        }
        return 0;
    }
    if (!any) {
        if (rtc_mode() == 2) return 1;
        g2_warn_once("bbt_chan_plan_create");
        p->k2 = nullptr;
        return 0;
    }
    std::lock_guard<std::mutex> lock(mu);
    chosen[key] = pick;
    return 0;
}

extern "C" {

int bbt_chan_plan_create(bbt_chan_plan** plan, int n_chan, int n_stream, int direction) {
    ARG_TRY(plan, "bbt_chan_plan_create: null argument");
    *plan = nullptr;
    const bool fast = is_pow2(n_chan) && n_chan >= 2 && n_chan <= 4096;
    // (8192 and 16384 channels: one workgroup of 512 / 1024 threads per transform, stream pairs, plain directions)
    const bool big = (n_chan == 8192 || n_chan == 16384) && n_stream % 2 == 0 && (direction == -1 || direction == 1);
    ARG_TRY(fast || big || (n_chan >= 2 && n_chan <= BBT_GEN_MAX_LEN && is_7smooth(n_chan)),
            "bbt_chan_plan_create: n_chan=%d must be a power of two in [2, 16384] (above 4096: stream "
            "pairs, directions -1 / +1) or a product of 2, 3, 5, 7 up to 8192", n_chan);
    const bool single = n_stream == 1 && fast && n_chan >= 256;
    ARG_TRY(single || (n_stream >= 2 && n_stream % 2 == 0 && n_stream <= 65535 * 2),
            "bbt_chan_plan_create: n_stream=%d must be even and >= 2 (or 1 for a power-of-two n_chan "
            "in [256, 4096])", n_stream);
    // direction -2: forward transform of ONE stream z = a + i b made of two real streams, writing their
    // half spectra (n_spectra, n_chan / 2 + 1, 2) instead of Z (Channelize of float32 streams in one pass)
    // (+2: the inverse -- half spectra of two real streams in, z = a + i b out)
    const bool split_real = direction == -2 || direction == 2;
    ARG_TRY(!split_real || (fast && n_chan >= 256),
            "bbt_chan_plan_create: direction -2 / +2 needs a power-of-two n_chan in [256, 4096]");
    if (split_real) direction /= 2;
    ARG_TRY(direction == -1 || direction == 1, "bbt_chan_plan_create: direction must be -1, +1, -2 or +2");
    bbt_chan_plan* p = new bbt_chan_plan;
    p->n = n_chan;
    p->S = n_stream;
    p->npair = n_stream / 2;
    p->dir = direction;
    p->split_real = split_real;
    if (big) {
        if (get_big_table(n_chan, &p->big)) {
            delete p;
            return 1;
        }
    } else if (!fast) {
        p->generic = true;
        if (!factor_7smooth(n_chan, &p->g) || get_gen_table(&p->g, &p->wn)) {
            if (g_err.empty()) fail("bbt_chan_plan_create: cannot factor n_chan=%d", n_chan);
            delete p;
            return 1;
        }
        for (int ct = 8; ct >= 1; ct /= 2)           // (a power of two: gen_stage)
            if (p->npair % ct == 0 && (int64_t)n_chan * ct <= BBT_GEN_MAX_LEN) {
                p->ct = ct;
                break;
            }
        // The kernels compiled for this length -- and which of them.  A streaming transform wants few
        // registers and many waves, but how few depends on the length (MI355X, round 5, one stream
        // pair, 1 GB and more in HBM, Gsamples/s general | specialised at 20 / 16 / 10 points per
        // thread: 1536: 150 | 141 / 142 / 171, 2187: 119 | 122 / 167 / 164, 4374: 116 | 139 / 156 / 139,
        // 6174: 96 | 132 / 107 / 113, 6561: 119 | 121 / 147 / 151, 3000: 122 | 127 / 135 / 128, 1000: 164 |
        // 144 / 158 / 160, 360: 162 | 90 / 118 / 166; short transforms fill a wave with several of them:
        // 14: 25 -> 73, 30: 44 -> 71).  So the plan is made for 10, 16 and 20 points, each candidate --
        // and the general kernel -- is timed once on scratch memory, and the fastest is kept (a few
        // hundred milliseconds per new length and stream count, remembered for the process).
        // On four stream pairs and more each is also tried `wide`: as many pairs as fit a workgroup of
        // 1024 threads (whole lines: Channelize(3000) on 16 / 128 / 2048 streams 127 / 107 / 110 -> 200 / 201 / 203 G
        // stream-samples/s).  BBT_G2_TUNE=0: no timing, 10 points; BBT_G2_CHAN=none: the general kernels.
        const char* chan_env = getenv("BBT_G2_CHAN");
        if (rtc_mode() && !(chan_env && !strcmp(chan_env, "none")) && chan_pick(p, n_chan, direction)) {
            delete p;
            return 1;
        }
    } else if ((n_chan >= 256 && get_tables(n_chan, &p->tab)) ||
               (n_chan < 256 && get_wroot(&p->wroot))) {
        delete p;
        return 1;
    }
    *plan = p;
    return 0;
}

int bbt_chan_plan_destroy(bbt_chan_plan* p) {
    delete p;
    return 0;
}

int bbt_chan_execute(bbt_chan_plan* p, const void* in_dev, void* out_dev, int64_t n_spectra,
                     bbt_stream stream) {
    ARG_TRY(p && in_dev && out_dev, "bbt_chan_execute: null argument");
    ARG_TRY(n_spectra >= 0, "bbt_chan_execute: n_spectra < 0");
    if (n_spectra == 0) return 0;
    // keep grid.x within limits: process in slabs
    const int64_t slab = (int64_t)1 << 20;
    const float2* in = (const float2*)in_dev;
    float2* out = (float2*)out_dev;
    if (p->big) {
        const float scale = p->dir < 0 ? 1.0f : 1.0f / (float)p->n;
        const int64_t per = std::max<int64_t>(1, ((int64_t)1 << 30) / p->npair);      // (grid.x < 2^31)
        for (int64_t s0 = 0; s0 < n_spectra; s0 += per) {
            const int64_t ns = std::min(per, n_spectra - s0);
            const int64_t off = s0 * p->n * p->S;
            const dim3 grid((unsigned)(ns * p->npair));
#define BBT_BIG_ROWS(N_, SIGN_)                                                                        \
    do {                                                                                               \
        constexpr size_t lds = BigGeo<N_>::LDS_ELEMS * sizeof(v2);                                     \
        if (ensure_dyn_lds((const void*)k_fft_rows_big<N_, SIGN_>, lds)) return 1;                     \
        hipLaunchKernelGGL((k_fft_rows_big<N_, SIGN_>), grid, dim3(N_ / 16), lds, (hipStream_t)stream, \
                           in + off, out + off, (long long)ns, p->S, scale, p->big);                   \
    } while (0)
            if (p->n == 8192) {
                if (p->dir < 0) BBT_BIG_ROWS(8192, -1); else BBT_BIG_ROWS(8192, +1);
            } else {
                if (p->dir < 0) BBT_BIG_ROWS(16384, -1); else BBT_BIG_ROWS(16384, +1);
            }
#undef BBT_BIG_ROWS
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    for (int64_t s0 = 0; s0 < n_spectra; s0 += slab) {
        const int64_t ns = (n_spectra - s0 < slab) ? n_spectra - s0 : slab;
        const int64_t off = s0 * p->n * p->S;
        if (p->generic && p->k2) {
            const int bt = p->q.ct / p->cp;
            const dim3 grid((unsigned)(((ns + bt - 1) / bt) * (p->npair / p->cp))), block(p->q.threads());
            if (g2_launch(p->k2, grid, block, (hipStream_t)stream, in + off, out + off, p->S, p->cp, (long long)ns,
                          p->dir < 0 ? 1.0f : 1.0f / (float)p->n, (const cf*)p->qw))
                return 1;
            continue;
        }
        if (p->generic) {
            const size_t lds = (size_t)p->n * p->ct * sizeof(f4);
            const dim3 grid((unsigned)(ns * (p->npair / p->ct))), block(gen_threads(p->n * p->ct));
            if (p->dir < 0) {
                if (ensure_dyn_lds((const void*)k_gen_fft_rows<-1>, lds)) return 1;
                hipLaunchKernelGGL((k_gen_fft_rows<-1>), grid, block, lds, (hipStream_t)stream, in + off,
                                   out + off, p->S, p->ct, 1.0f, p->g, p->wn);
            } else {
                if (ensure_dyn_lds((const void*)k_gen_fft_rows<+1>, lds)) return 1;
                hipLaunchKernelGGL((k_gen_fft_rows<+1>), grid, block, lds, (hipStream_t)stream, in + off,
                                   out + off, p->S, p->ct, 1.0f / (float)p->n, p->g, p->wn);
            }
            continue;
        }
        // (half spectra of a real pair: n/2 + 1 channels x 2 streams per spectrum on that side)
        const int64_t half_off = s0 * (p->n / 2 + 1) * 2 * p->S;
        int rc = (p->dir < 0)
                     ? chan_dispatch<-1>(p, in + off, out + (p->split_real ? half_off : off), ns, 1.0f,
                                         (hipStream_t)stream)
                     : chan_dispatch<+1>(p, in + (p->split_real ? half_off : off), out + off, ns,
                                         1.0f / (float)p->n, (hipStream_t)stream);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// polyphase filter bank
struct bbt_pfb_plan {
    int n = 0, S = 0, npair = 0, n_tap = 0;
    bool split_real = false;    // n_stream -1: one stream z = a + i b of two real streams, half spectra out
    float* taps = nullptr;
    float* taps_quads = nullptr;   // window kernels: [column group][tap quad][thread] float4
    FftTables tab;
    FftTables tab4096;   // for the sliding-window kernel
    bool window = false;
    // several stream pairs per workgroup of the sliding-window kernel (pp > 1: geometry gn, tables tabg,
    // taps_quads permuted for gn / 16 threads)
    int pp = 1, gn = 4096, gl = 0;
    FftTables tabg;
    float* taps_quads_pp = nullptr;
    // many streams: the window as a streaming pass over whole rows (k_pfb_fir_rows), then the
    // transform in place through a channelizer plan (k_fft_rows_pp)
    bbt_chan_plan* two_pass = nullptr;
};

// geometry of one pair in the several-pairs form of the window kernel: 128 threads and 2048 / N
// spectra per workgroup up to 1024 channels, 256 threads and two spectra for 2048
static constexpr int pfb_pp_geometry(int n) { return n <= 1024 ? 2048 : 4096; }

template <int N, int NTAP>
static void launch_pfb_window(const bbt_pfb_plan* p, const float2* in, float2* out, int64_t n_spec,
                              hipStream_t st) {
    constexpr int NG = 4096 / N;
    if (p->S == 1) {            // one stream: two groups of NG spectra side by side
        const unsigned gx = (unsigned)((n_spec + 2 * NG - 1) / (2 * NG));
        if (p->split_real)
            hipLaunchKernelGGL((k_pfb_window<N, NTAP, true, true>), dim3(gx), dim3(256), 0, st, in, out,
                               (long long)n_spec, 1, p->taps_quads, p->tab4096.tw0, p->tab4096.tw1);
        else
            hipLaunchKernelGGL((k_pfb_window<N, NTAP, true>), dim3(gx), dim3(256), 0, st, in, out,
                               (long long)n_spec, 1, p->taps_quads, p->tab4096.tw0, p->tab4096.tw1);
        return;
    }
    if (p->pp > 1) {            // several pairs per workgroup (many streams): pfb_pick
        constexpr int GN = pfb_pp_geometry(N);
        const unsigned gy = (unsigned)((n_spec + GN / N - 1) / (GN / N));
        if (p->pp == 4)
            hipLaunchKernelGGL((k_pfb_window<N, NTAP, false, false, 4, GN>), dim3(gy * (p->npair / 4)),
                               dim3(GN / 16 * 4), 0, st, in, out, (long long)n_spec, p->S, p->taps_quads_pp,
                               p->tabg.tw0, p->tabg.tw1, p->gl);
        if constexpr (GN == 2048) {
            if (p->pp == 8)
                hipLaunchKernelGGL((k_pfb_window<N, NTAP, false, false, 8, GN>), dim3(gy * (p->npair / 8)),
                                   dim3(GN / 16 * 8), 0, st, in, out, (long long)n_spec, p->S, p->taps_quads_pp,
                                   p->tabg.tw0, p->tabg.tw1, p->gl);
        }
        return;
    }
    const unsigned gx = (unsigned)((n_spec + NG - 1) / NG);
    if (p->split_real)          // every stream z = a + i b of two real streams: half spectra out
        hipLaunchKernelGGL((k_pfb_window<N, NTAP, false, true>), dim3(gx * p->npair), dim3(256), 0, st, in, out,
                           (long long)n_spec, p->S, p->taps_quads, p->tab4096.tw0, p->tab4096.tw1);
    else
        hipLaunchKernelGGL((k_pfb_window<N, NTAP>), dim3(gx * p->npair), dim3(256), 0, st, in, out,
                           (long long)n_spec, p->S, p->taps_quads, p->tab4096.tw0, p->tab4096.tw1);
}

// sliding-window variants exist for these (n_chan, n_tap)
static bool pfb_window_dispatch(const bbt_pfb_plan* p, const float2* in, float2* out,
                                int64_t n_spec, hipStream_t st, bool probe) {
#define BBT_PW(N_, T_)                                                  \
    if (p->n == N_ && p->n_tap == T_) {                                 \
        if (!probe) launch_pfb_window<N_, T_>(p, in, out, n_spec, st);  \
        return true;                                                    \
    }
    BBT_PW(256, 4) BBT_PW(512, 4) BBT_PW(1024, 4) BBT_PW(2048, 4)
    BBT_PW(256, 8) BBT_PW(512, 8) BBT_PW(1024, 8) BBT_PW(2048, 8)
    BBT_PW(256, 12) BBT_PW(512, 12) BBT_PW(1024, 12) BBT_PW(2048, 12)
    BBT_PW(256, 16) BBT_PW(512, 16) BBT_PW(1024, 16) BBT_PW(2048, 16)
#undef BBT_PW
    return false;
}

template <int N>
static void launch_pfb(const bbt_pfb_plan* p, const float2* in, float2* out, int64_t n_spec,
                       hipStream_t st) {
    constexpr int FPW = (N >= 1024) ? (4096 / N >= 4 ? 4 : 4096 / N) : (N == 512 ? 8 : 16);
    const unsigned gx = (unsigned)((n_spec + FPW - 1) / FPW);
    hipLaunchKernelGGL((k_pfb<N, FPW>), dim3(gx * p->npair), dim3(FPW * N / 16), 0, st, in, out,
                       (long long)n_spec, p->S, p->n_tap, p->taps, p->tab.tw0, p->tab.tw1);
}

// Which route a filter-bank plan takes.  Measured on MI355X (round 5, 1024 channels, G
// stream-samples/s at 16 / 128 / 2048 streams): one pair per workgroup (16 bytes of every complete
// sample per lane) 4 taps 140 / 99 / 82, 12 taps 107 / 54 / 54; two passes (window over whole rows,
// transform in place: 33 bytes per stream-sample instead of 16) 150 / 159 / 146 and 144 / 154 / 133;
// several pairs per workgroup, lanes over the pairs first (64- or 128-byte runs): 4 taps
// 223 / 228 / 186, 12 taps 148 / 131 / 130 -- with 128 registers and at most two workgroups per CU
// its NTAP + NG - 1 row loads for NG spectra weigh more the more taps there are.  Which of them
// and which order of workgroups wins depends on taps, channels and streams, so the candidates
// are timed once on scratch memory (as chan_pick) and the choice is remembered for the process.
// BBT_PFB_TUNE=0: the round-4 rule (two passes from 16 streams on); BBT_PFB_TWO_PASS=0 / 1,
// BBT_PFB_PP=4 / 8 (+ BBT_PFB_GL) force a route (dev).
#ifndef BBT_PFB_NI
#define BBT_PFB_NI 192                // spectra per sweep of the window pass (k_pfb_fir_rows): 96 -> 192 +2-5 %, 384 no better
#endif
static int pfb_pick(bbt_pfb_plan* p, bool pp_ok) {
    const int n_tap = p->n_tap, n_stream = p->S;
    const bool two_ok = !p->split_real && n_stream >= 2 && (n_tap == 4 || n_tap == 8 || n_tap == 12 || n_tap == 16);
    const char* env2 = getenv("BBT_PFB_TWO_PASS");
    const char* envp = getenv("BBT_PFB_PP");
    auto set_pp = [&](int pp, int gl) {
        p->pp = pp;
        p->gl = (gl > 0 && (p->npair / pp) % gl == 0) ? gl : 0;
    };
    auto need_two_pass = [&]() -> int {
        return p->two_pass ? 0 : bbt_chan_plan_create(&p->two_pass, p->n, n_stream, -1);
    };
    if (envp && pp_ok) {
        const int pp = atoi(envp);
        if ((pp == 4 || (pp == 8 && p->gn == 2048)) && p->npair % pp == 0) {
            set_pp(pp, getenv("BBT_PFB_GL") ? atoi(getenv("BBT_PFB_GL")) : 0);
            return 0;
        }
    }
    // (two passes also where no sliding-window kernel exists and every spectrum would re-read its rows:
    // 16 x 4096 on two streams 64.4 -> 84.2 G complete samples/s, 0.26 -> 0.34; with a window kernel one
    // pass wins on few streams: 8 x 2048 125 against 91, 4 x 1024 164 against 88)
    const bool rule_two = two_ok && (n_stream >= 16 || (!p->window && n_tap >= 8));
    if (env2) return (two_ok && atoi(env2) != 0) ? need_two_pass() : 0;
    const bool tune = !(getenv("BBT_PFB_TUNE") && !strcmp(getenv("BBT_PFB_TUNE"), "0"));
    if (!pp_ok || !tune || n_stream < 8) return rule_two ? need_two_pass() : 0;

    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, std::pair<int, int>> chosen;   // -> (pp: 0 two passes, 1 one pair; gl)
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const auto key = std::make_tuple(dev, p->n, n_tap, n_stream);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = chosen.find(key);
        if (it != chosen.end()) {
            if (it->second.first == 0) return need_two_pass();
            if (it->second.first > 1) set_pp(it->second.first, it->second.second);
            return 0;
        }
    }
    if (two_ok && need_two_pass()) return 1;
    bbt_chan_plan* const tp = p->two_pass;
    const size_t per = (size_t)p->n * p->S * sizeof(cf);
    const auto t_start = std::chrono::steady_clock::now();
    auto trace = [&](const char* what, int x = 0, int y = 0) {
        if (!getenv("BBT_PFB_TRACE")) return;
        fprintf(stderr, "bbt: pfb_pick %d x %d on %d streams, %8.3f ms: %s %d %d\n", n_tap, p->n, n_stream,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what, x, y);
        fflush(stderr);
    };
    trace("allocating");
    // (scratch memory: tune_acquire; on the null stream, as chan_pick)
    size_t got = 0;
    char* base = (char*)tune_acquire((2 * 16 + n_tap) * per, (size_t)2 << 30, &got);
    const int64_t ns = base ? (int64_t)((got - (n_tap - 1) * per) / (2 * per)) : 0;
    void *a = base, *b = base ? base + (ns + n_tap - 1) * per : nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool ready = base && hipMemsetAsync(a, 0, (ns + n_tap - 1) * per, nullptr) == hipSuccess &&
                       hipEventCreate(&e0) == hipSuccess &&
                       hipEventCreate(&e1) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    trace("allocated");
    struct Cand { int pp, gl; };
    std::vector<Cand> cands;
    if (two_ok) cands.push_back({0, 0});
    if (n_stream < 16) cands.push_back({1, 0});
    cands.push_back({4, 0});
    if ((p->npair / 4) % 4 == 0) cands.push_back({4, 4});
    if (p->gn == 2048 && p->npair % 8 == 0) {
        cands.push_back({8, 0});
        if (p->npair / 8 > 1) cands.push_back({8, 1});
    }
    Cand best = rule_two ? Cand{0, 0} : Cand{1, 0};
    float best_ms = 1e30f;
    for (const Cand& c : cands) {
        if (!ready) break;
        p->two_pass = c.pp == 0 ? tp : nullptr;
        p->pp = 1;
        p->gl = 0;
        if (c.pp > 1) set_pp(c.pp, c.gl);
        float ms = 1e30f;
        trace("candidate", c.pp, c.gl);
        if (!bbt_pfb_execute(p, a, b, ns, st)) {
            for (int r = 0; r < 2; ++r) {
                float t = 0.f;
                if (hipEventRecord(e0, st) != hipSuccess || bbt_pfb_execute(p, a, b, ns, st) ||
                    hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                    hipEventElapsedTime(&t, e0, e1) != hipSuccess) {
                    ms = 1e30f;
                    break;
                }
                ms = std::min(ms, t);
            }
        }
        if (getenv("BBT_RTC_VERBOSE"))
            fprintf(stderr, "bbt: PolyphaseFilterBank %d x %d on %d streams, %s (pairs %d, order %d): %.3f ms\n", n_tap, p->n,
                    n_stream, c.pp == 0 ? "two passes" : "one pass", c.pp, c.gl, ms);
        if (ms < best_ms) {
            best_ms = ms;
            best = c;
        }
    }
    trace("timed");
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (base) tune_release();
    if (!ready) hipGetLastError();
    trace("freed");
    p->two_pass = nullptr;
    p->pp = 1;
    p->gl = 0;
    if (best.pp == 0) p->two_pass = tp;
    else if (tp) bbt_chan_plan_destroy(tp);
    if (best.pp > 1) set_pp(best.pp, best.gl);
    if (ready) {
        std::lock_guard<std::mutex> lock(mu);
        chosen[key] = std::make_pair(best.pp, p->gl);
    }
    return 0;
}

extern "C" {

int bbt_pfb_plan_create(bbt_pfb_plan** plan, int n_tap, int n_chan, int n_stream,
                        const float* taps_host) {
    ARG_TRY(plan && taps_host, "bbt_pfb_plan_create: null argument");
    *plan = nullptr;
    ARG_TRY(fft_len_ok(n_chan),
            "bbt_pfb_plan_create: n_chan=%d must be a power of two in [256, 4096]", n_chan);
    ARG_TRY(n_tap >= 1 && n_tap <= 64, "bbt_pfb_plan_create: n_tap=%d must be in [1, 64]", n_tap);
    // n_stream -1: as 1, the stream being z = a + i b of two real streams; out receives their half
    // spectra (n_spectra, n_chan / 2 + 1, 2) -- the filter bank of two float32 streams in one pass
    // (a negative even n_stream -S: S such streams, in pairs)
    const bool split_real = n_stream == -1 || (n_stream < 0 && n_stream % 2 == 0);
    if (split_real) n_stream = -n_stream;
    ARG_TRY(n_stream == 1 || (n_stream >= 2 && n_stream % 2 == 0 && n_stream <= 65535 * 2),
            "bbt_pfb_plan_create: n_stream=%d must be even and >= 2", n_stream);
    bbt_pfb_plan* p = new bbt_pfb_plan;
    p->split_real = split_real;
    p->n = n_chan;
    p->S = n_stream;
    p->npair = n_stream / 2;
    p->n_tap = n_tap;
    const size_t tb = (size_t)n_tap * n_chan * sizeof(float);
    p->window = pfb_window_dispatch(p, nullptr, nullptr, 0, nullptr, true);
    if ((n_stream == 1 || split_real) && !p->window) {   // only the sliding-window kernels take one stream / split
        delete p;
        return fail("bbt_pfb_plan_create: one stream needs n_chan in 256..2048 and 4, 8, 12 or 16 taps "
                    "(got %d x %d); pad to two streams otherwise", n_tap, n_chan);
    }
    // several stream pairs per workgroup of the window kernel are possible for (pfb_pick chooses)
    const bool pp_ok = p->window && !split_real && n_chan <= 2048 && p->npair % 4 == 0;
    p->gn = pp_ok ? pfb_pp_geometry(n_chan) : 4096;
    if ((pp_ok && get_tables(p->gn, &p->tabg)) ||
        (p->window && get_tables(4096, &p->tab4096)) ||
        get_tables(n_chan, &p->tab) || hipMalloc((void**)&p->taps, tb) != hipSuccess ||
        hipMemcpy(p->taps, taps_host, tb, hipMemcpyHostToDevice) != hipSuccess) {
        if (g_err.empty()) fail("bbt_pfb_plan_create: tap upload failed");
        bbt_pfb_plan_destroy(p);
        return 1;
    }
    if (p->window) {
        // taps of column tau + T c, four at a time: quads[((c * n_tap / 4) + q) * T + tau][k] = h[4 q + k]
        // (T = 256 threads per pair; gn / 16 in the several-pairs form)
        for (int form = 0; form < (pp_ok ? 2 : 1); ++form) {
            const int T = form ? p->gn / 16 : 256, cols = n_chan / T, quads = n_tap / 4;
            std::vector<float> perm((size_t)n_tap * n_chan);
            for (int c = 0; c < cols; ++c)
                for (int q = 0; q < quads; ++q)
                    for (int tau = 0; tau < T; ++tau)
                        for (int k = 0; k < 4; ++k)
                            perm[(((size_t)c * quads + q) * T + tau) * 4 + k] =
                                taps_host[(size_t)(4 * q + k) * n_chan + tau + T * c];
            float** dst = form ? &p->taps_quads_pp : &p->taps_quads;
            if (hipMalloc((void**)dst, tb) != hipSuccess ||
                hipMemcpy(*dst, perm.data(), tb, hipMemcpyHostToDevice) != hipSuccess) {
                fail("bbt_pfb_plan_create: tap upload failed");
                bbt_pfb_plan_destroy(p);
                return 1;
            }
        }
    }
    if (pfb_pick(p, pp_ok)) {
        bbt_pfb_plan_destroy(p);
        return 1;
    }
    *plan = p;
    return 0;
}

int bbt_pfb_plan_destroy(bbt_pfb_plan* p) {
    if (!p) return 0;
    if (p->two_pass) bbt_chan_plan_destroy(p->two_pass);
    if (p->taps_quads) hipFree(p->taps_quads);
    if (p->taps_quads_pp) hipFree(p->taps_quads_pp);
    if (p->taps) hipFree(p->taps);
    delete p;
    return 0;
}

int bbt_pfb_execute(bbt_pfb_plan* p, const void* in_dev, void* out_dev, int64_t n_spectra,
                    bbt_stream stream) {
    ARG_TRY(p && in_dev && out_dev, "bbt_pfb_execute: null argument");
    ARG_TRY(n_spectra >= 0, "bbt_pfb_execute: n_spectra < 0");
    if (n_spectra == 0) return 0;
    const int64_t slab = (int64_t)1 << 20;
    const float2* in = (const float2*)in_dev;
    float2* out = (float2*)out_dev;
    for (int64_t s0 = 0; s0 < n_spectra; s0 += slab) {
        const int64_t ns = (n_spectra - s0 < slab) ? n_spectra - s0 : slab;
        const int64_t off = s0 * p->n * p->S;
        hipStream_t st = (hipStream_t)stream;
        if (p->two_pass) {
            // (one launch of each pass per slab: in pieces whose windowed rows would stay in the
            // Infinity Cache between the passes -- 48 ... 192 MiB -- it is SLOWER, 93-147 against
            // 148-159 G stream-samples/s for 16 / 128 streams: shorter launches, nothing to overlap)
            constexpr int NI = BBT_PFB_NI;
            const long long row16 = (long long)p->n * p->npair;
            const dim3 grid((unsigned)((row16 + 255) / 256), (unsigned)((ns + NI - 1) / NI));
            const float4* src = reinterpret_cast<const float4*>(in + off);
            float4* dst = reinterpret_cast<float4*>(out + off);
            switch (p->n_tap) {
#define BBT_PF(T_) case T_: hipLaunchKernelGGL((k_pfb_fir_rows<T_, NI>), grid, dim3(256), 0, st, src, dst, (long long)ns, \
                                               row16, p->npair, p->n, p->taps); break;
                BBT_PF(4) BBT_PF(8) BBT_PF(12) BBT_PF(16)
#undef BBT_PF
                default: return fail("pfb: two passes need 4, 8, 12 or 16 taps");
            }
            if (bbt_chan_execute(p->two_pass, out + off, out + off, ns, stream)) return 1;
            continue;
        }
        if (p->window) {
            // input of slab s0 starts at spectrum s0 (same offset as the output, except
            // for half spectra of a real pair: n/2 + 1 channels x 2 streams per spectrum)
            const int64_t out_off = p->split_real ? s0 * (p->n / 2 + 1) * 2 * p->S : off;
            pfb_window_dispatch(p, in + off, out + out_off, ns, st, false);
            continue;
        }
        switch (p->n) {
            case 256: launch_pfb<256>(p, in + off, out + off, ns, st); break;
            case 512: launch_pfb<512>(p, in + off, out + off, ns, st); break;
            case 1024: launch_pfb<1024>(p, in + off, out + off, ns, st); break;
            case 2048: launch_pfb<2048>(p, in + off, out + off, ns, st); break;
            case 4096: launch_pfb<4096>(p, in + off, out + off, ns, st); break;
            default: return fail("pfb: unsupported n_chan %d", p->n);
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// detection + integration
extern "C" int bbt_detect_integrate(const void* in_dev, void* out_dev, int64_t n_out, int64_t step,
                                    int64_t n_elem, int mode, int average, bbt_stream stream) {
    ARG_TRY(in_dev && out_dev, "bbt_detect_integrate: null argument");
    ARG_TRY(n_out >= 0 && step >= 1 && n_elem >= 1, "bbt_detect_integrate: bad sizes");
    ARG_TRY(mode >= 0 && mode <= 2, "bbt_detect_integrate: mode must be 0 (square), 1 (power) or 2 (sum)");
    ARG_TRY(mode != 1 || n_elem % 2 == 0,
            "bbt_detect_integrate: power needs (X, Y) pairs, n_elem=%lld is odd", (long long)n_elem);
    if (n_out == 0) return 0;
    const long long q = mode == 1 ? n_elem / 2 : n_elem;
    const long long tiles = (q + 255) / 256;
    ARG_TRY(n_out * tiles < (1ll << 31), "bbt_detect_integrate: too many output elements for one call");
    const float scale = average ? 1.0f / (float)step : 1.0f;
    const dim3 grid((unsigned)(n_out * tiles)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0)
        hipLaunchKernelGGL((k_detect_integrate<0>), grid, block, 0, st, in_dev, out_dev,
                           (long long)n_out, (long long)step, q, scale);
    else if (mode == 1)
        hipLaunchKernelGGL((k_detect_integrate<1>), grid, block, 0, st, in_dev, out_dev,
                           (long long)n_out, (long long)step, q, scale);
    else
        hipLaunchKernelGGL((k_detect_integrate<2>), grid, block, 0, st, in_dev, out_dev,
                           (long long)n_out, (long long)step, q, scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int bbt_detect_power_axis(const void* in_dev, void* out_dev, int64_t n_out, int64_t step,
                                     int outer, int inner, int average, bbt_stream stream) {
    ARG_TRY(in_dev && out_dev, "bbt_detect_power_axis: null argument");
    ARG_TRY(n_out >= 0 && step >= 1 && outer >= 1 && inner >= 1, "bbt_detect_power_axis: bad sizes");
    if (n_out == 0) return 0;
    const long long total = (long long)n_out * outer * inner;
    ARG_TRY((total + 255) / 256 < (1ll << 31), "bbt_detect_power_axis: too many output elements for one call");
    hipLaunchKernelGGL(k_power_axis, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)in_dev, (float*)out_dev, (long long)n_out, (long long)step, outer, inner,
                       average ? 1.0f / (float)step : 1.0f);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// integer sample shifts
struct bbt_shift_plan {
    int n_elem = 0, elem_bytes = 0;      // as given
    int* offset = nullptr;
    // neighbours with equal offsets merged into one wider element (used when the
    // buffers of a call are aligned to it)
    int merged_n_elem = 0, merged_bytes = 0;
    int* merged_offset = nullptr;
    // tiled form (k_shift_tiled): per group of merged elements that fill a cache line, the
    // smallest offset and the span of the offsets; 0 groups: gather only
    int n_group = 0, window = 0;
    int* group_lo = nullptr;
    int* group_span = nullptr;
};

extern "C" int bbt_shift_plan_destroy(bbt_shift_plan* p);
extern "C" int bbt_shift_plan_create(bbt_shift_plan** plan, int n_elem, int elem_bytes,
                                     const int32_t* offsets_host) {
    ARG_TRY(plan && offsets_host, "bbt_shift_plan_create: null argument");
    *plan = nullptr;
    ARG_TRY(n_elem >= 1, "bbt_shift_plan_create: n_elem=%d must be >= 1", n_elem);
    ARG_TRY(elem_bytes == 4 || elem_bytes == 8, "bbt_shift_plan_create: elem_bytes must be 4 or 8");
    for (int e = 0; e < n_elem; ++e)
        ARG_TRY(offsets_host[e] >= 0, "bbt_shift_plan_create: offset[%d]=%d is negative", e,
                offsets_host[e]);
    // merge aligned groups of 2 (4) neighbouring elements that move together, up to 16 bytes
    int m = 1;
    while (m * 2 * elem_bytes <= 16 && n_elem % (m * 2) == 0) {
        bool same = true;
        for (int e = 0; e < n_elem && same; e += m * 2)
            for (int k = 1; k < m * 2 && same; ++k) same = offsets_host[e + k] == offsets_host[e];
        if (!same) break;
        m *= 2;
    }
    std::vector<int> merged(n_elem / m);
    for (int e = 0; e < n_elem / m; ++e) merged[e] = offsets_host[e * m];
    bbt_shift_plan* p = new bbt_shift_plan;
    p->n_elem = n_elem;
    p->elem_bytes = elem_bytes;
    p->merged_n_elem = n_elem / m;
    p->merged_bytes = elem_bytes * m;
    if (hipMalloc((void**)&p->offset, (n_elem + merged.size()) * sizeof(int)) != hipSuccess ||
        hipMemcpy(p->offset, offsets_host, n_elem * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->offset + n_elem, merged.data(), merged.size() * sizeof(int),
                  hipMemcpyHostToDevice) != hipSuccess) {
        fail("bbt_shift_plan_create: uploading the offsets failed");
        if (p->offset) hipFree(p->offset);
        delete p;
        return 1;
    }
    p->merged_offset = p->offset + n_elem;
    // groups of merged elements that fill a 128-byte line: the tiled kernel stages their input
    // lines in LDS.  Window: 1024 rows (128 KiB, one workgroup per CU) when some group's offsets
    // span more than 128 rows, else 512 (two per CU); half of it is output rows.
    const int g_elems = 128 / p->merged_bytes;
    if (p->merged_bytes >= 8 && p->merged_n_elem % g_elems == 0) {
        const int ng = p->merged_n_elem / g_elems;
        std::vector<int> lo(ng), span(ng);
        int tiled = 0, widest = 0;
        for (int g = 0; g < ng; ++g) {
            int a = merged[g * g_elems], b = a;
            for (int k = 1; k < g_elems; ++k) {
                a = std::min(a, merged[g * g_elems + k]);
                b = std::max(b, merged[g * g_elems + k]);
            }
            lo[g] = a;
            span[g] = b - a;
            if (span[g] <= 512) {
                ++tiled;
                widest = std::max(widest, span[g]);
            }
        }
        if (2 * tiled >= ng) {                 // (mostly scattered offsets: the gather kernel as before)
            if (hipMalloc((void**)&p->group_lo, 2 * ng * sizeof(int)) != hipSuccess ||
                hipMemcpy(p->group_lo, lo.data(), ng * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(p->group_lo + ng, span.data(), ng * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                fail("bbt_shift_plan_create: uploading the group table failed");
                bbt_shift_plan_destroy(p);
                return 1;
            }
            p->group_span = p->group_lo + ng;
            p->n_group = ng;
            p->window = widest > 128 ? 1024 : 512;
        }
    }
    *plan = p;
    return 0;
}

extern "C" int bbt_shift_plan_destroy(bbt_shift_plan* p) {
    if (!p) return 0;
    if (p->offset) hipFree(p->offset);
    if (p->group_lo) hipFree(p->group_lo);
    delete p;
    return 0;
}

extern "C" int bbt_shift_execute(bbt_shift_plan* p, const void* in_dev, void* out_dev,
                                 int64_t n_out, bbt_stream stream) {
    ARG_TRY(p && in_dev && out_dev, "bbt_shift_execute: null argument");
    ARG_TRY(n_out >= 0, "bbt_shift_execute: n_out < 0");
    if (n_out == 0) return 0;
#ifndef BBT_SHIFT_ITER
#define BBT_SHIFT_ITER 2     // rows per thread: 1 2.83, 2 3.17, 4 3.02, 8 2.95, 16 2.29 TB/s (128 x 8-byte elements, offsets over 1000 rows)
#endif
    constexpr int ITER = BBT_SHIFT_ITER;
    const bool wide = (((uintptr_t)in_dev | (uintptr_t)out_dev) % (uintptr_t)p->merged_bytes) == 0;
    const int n_elem = wide ? p->merged_n_elem : p->n_elem;
    const int bytes = wide ? p->merged_bytes : p->elem_bytes;
    const int* offset = wide ? p->merged_offset : p->offset;
    int lg_le = 0;
    while ((1 << lg_le) < n_elem && lg_le < 8) ++lg_le;
    const long long rows_per_block = (256 >> lg_le) * ITER;
    const long long gx = (n_out + rows_per_block - 1) / rows_per_block;
    const long long gy = ((long long)n_elem + (1 << lg_le) - 1) >> lg_le;
    ARG_TRY(gx < (1ll << 31) && gy <= 65535, "bbt_shift_execute: too many elements for one call");
    const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (wide && p->n_group) {
        const int rows = p->window / 2;
        const long long tx = (n_out + rows - 1) / rows;
        ARG_TRY(tx < (1ll << 31) && p->n_group <= 65535, "bbt_shift_execute: too many elements for one call");
        const dim3 tgrid((unsigned)tx, (unsigned)p->n_group);
        const size_t lds = (size_t)p->window * 128;
        if (bytes == 16) {
            if (ensure_dyn_lds((const void*)k_shift_tiled<float4>, lds)) return 1;
            hipLaunchKernelGGL((k_shift_tiled<float4>), tgrid, block, lds, st, (const float4*)in_dev, (float4*)out_dev,
                               (long long)n_out, n_elem, offset, p->group_lo, p->group_span, p->window);
        } else {
            if (ensure_dyn_lds((const void*)k_shift_tiled<float2>, lds)) return 1;
            hipLaunchKernelGGL((k_shift_tiled<float2>), tgrid, block, lds, st, (const float2*)in_dev, (float2*)out_dev,
                               (long long)n_out, n_elem, offset, p->group_lo, p->group_span, p->window);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (bytes == 16)
        hipLaunchKernelGGL((k_shift_samples<float4, ITER>), grid, block, 0, st, (const float4*)in_dev,
                           (float4*)out_dev, (long long)n_out, n_elem, lg_le, offset);
    else if (bytes == 8)
        hipLaunchKernelGGL((k_shift_samples<float2, ITER>), grid, block, 0, st, (const float2*)in_dev,
                           (float2*)out_dev, (long long)n_out, n_elem, lg_le, offset);
    else
        hipLaunchKernelGGL((k_shift_samples<float, ITER>), grid, block, 0, st, (const float*)in_dev,
                           (float*)out_dev, (long long)n_out, n_elem, lg_le, offset);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// real-stream glue
extern "C" int bbt_real_op(const void* in_dev, void* out_dev, int op, int64_t n_total, int n_chan,
                           int n_stream, bbt_stream stream) {
    ARG_TRY(in_dev && out_dev, "bbt_real_op: null argument");
    ARG_TRY(op >= 0 && op <= 6, "bbt_real_op: op must be 0..6");
    ARG_TRY(n_total >= 0, "bbt_real_op: n_total < 0");
    ARG_TRY(op != 2 || (n_chan >= 2 && n_chan % 2 == 0 && n_stream >= 1),
            "bbt_real_op: half-to-full needs an even n_chan and n_stream >= 1");
    ARG_TRY(op < 4 || (n_chan >= 2 && n_chan % 2 == 0 && n_stream >= 2 && n_stream % 2 == 0),
            "bbt_real_op: splitting / merging two real streams needs even n_chan and n_stream");
    if (n_total == 0) return 0;
    ARG_TRY((n_total + 255) / 256 < (1ll << 31), "bbt_real_op: too many elements for one call");
    const dim3 grid((unsigned)((n_total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
        case 0: hipLaunchKernelGGL((k_real_ops<0>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
        case 1: hipLaunchKernelGGL((k_real_ops<1>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
        case 2: hipLaunchKernelGGL((k_real_ops<2>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
        case 3: hipLaunchKernelGGL((k_real_ops<3>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
        case 4: hipLaunchKernelGGL((k_real_ops<4>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
        case 6: hipLaunchKernelGGL((k_real_ops<4, true>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
        default: hipLaunchKernelGGL((k_real_ops<5>), grid, block, 0, st, in_dev, out_dev, (long long)n_total, n_chan, n_stream); break;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// per-stream complex factor
extern "C" int bbt_chirp(void* out_dev, int64_t n, int n_col, const double* freq_hz, const double* sideband,
                         const double* ref_hz, double rate_hz, double d_dm, double offset_s, bbt_stream stream) {
    ARG_TRY(out_dev && freq_hz && sideband && ref_hz, "bbt_chirp: null argument");
    ARG_TRY(n >= 1 && n_col >= 1 && n_col <= 65535 && rate_hz > 0, "bbt_chirp: bad sizes");
    std::vector<double4> h(n_col);
    for (int c = 0; c < n_col; ++c) {
        ARG_TRY(ref_hz[c] > 0, "bbt_chirp: reference frequency %g of column %d is not positive", ref_hz[c], c);
        h[c] = make_double4(freq_hz[c], sideband[c], ref_hz[c], 0.);
    }
    double4* col = nullptr;
    HIP_TRY(hipMalloc((void**)&col, sizeof(double4) * n_col));
    hipError_t e = hipMemcpyAsync(col, h.data(), sizeof(double4) * n_col, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_chirp, dim3((unsigned)((n + 255) / 256), n_col), dim3(256), 0, (hipStream_t)stream,
                           (float2*)out_dev, (long long)n, col, rate_hz, d_dm, offset_s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);     // (the host copy of `col` goes with this frame)
    hipFree(col);
    if (e != hipSuccess) return fail("bbt_chirp: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int bbt_scale_streams(const void* in_dev, void* out_dev, int64_t n_samples, int n_elem,
                                 const void* factor_dev, bbt_stream stream) {
    ARG_TRY(in_dev && out_dev && factor_dev, "bbt_scale_streams: null argument");
    ARG_TRY(n_samples >= 0 && n_elem >= 1, "bbt_scale_streams: bad sizes");
    const long long total = (long long)n_samples * n_elem;
    if (total == 0) return 0;
    ARG_TRY((total + 255) / 256 < (1ll << 31), "bbt_scale_streams: too many elements for one call");
    hipLaunchKernelGGL(k_scale_streams, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float2*)in_dev, (float2*)out_dev, total, n_elem,
                       (const float2*)factor_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// short-response convolution in the time domain

extern "C" int bbt_fir_plan_create(bbt_fir_plan** plan, int n_tap, int n_stream,
                                   const void* response_host) {
    ARG_TRY(plan && response_host, "bbt_fir_plan_create: null argument");
    *plan = nullptr;
    ARG_TRY(n_tap >= 1 && n_tap <= 1024, "bbt_fir_plan_create: n_tap=%d must be in [1, 1024]", n_tap);
    ARG_TRY(n_stream == 1 || (n_stream >= 2 && n_stream % 2 == 0),
            "bbt_fir_plan_create: n_stream=%d must be 1 or even (pad other odd counts)", n_stream);
    constexpr int R = BBT_FIR_R;
    bbt_fir_plan* p = new bbt_fir_plan;
    p->n_tap = n_tap;
    p->S = n_stream;
    p->npair = n_stream == 1 ? 1 : n_stream / 2;
    p->n_chunks = (n_tap + R - 1 + R - 1) / R;           // inputs u = r + m < n_tap + R - 1
    p->pitch = 256 + p->n_chunks;
    while (p->pitch % 16 != 2) ++p->pitch;
    p->tap_pitch = R * p->n_chunks + 2 * R;              // R - 1 zeros in front, zeros behind
    const cf* resp = (const cf*)response_host;           // (n_tap, n_stream), reference order
    std::vector<float2> re((size_t)p->npair * p->tap_pitch, make_float2(0.f, 0.f)), im(re);
    for (int sp = 0; sp < p->npair; ++sp)
        for (int m = 0; m < n_tap; ++m) {
            const cf a = resp[(size_t)(n_tap - 1 - m) * n_stream + 2 * sp];
            const cf b = n_stream == 1 ? a          // one stream: both halves of the time range
                                       : resp[(size_t)(n_tap - 1 - m) * n_stream + 2 * sp + 1];
            re[(size_t)sp * p->tap_pitch + (R - 1) + m] = make_float2(a.x, b.x);
            im[(size_t)sp * p->tap_pitch + (R - 1) + m] = make_float2(a.y, b.y);
            if (a.y != 0.f || b.y != 0.f) p->cplx = true;
        }
    if (upload(&p->tre, re) || upload(&p->tim, im)) {
        if (p->tre) hipFree(p->tre);
        delete p;
        return 1;
    }
    *plan = p;
    return 0;
}

extern "C" int bbt_fir_plan_destroy(bbt_fir_plan* p) {
    if (!p) return 0;
    if (p->tre) hipFree(p->tre);
    if (p->tim) hipFree(p->tim);
    delete p;
    return 0;
}

extern "C" int bbt_fir_execute(bbt_fir_plan* p, const void* in_dev, void* out_dev, int64_t n_out,
                               bbt_stream stream) {
    ARG_TRY(p && in_dev && out_dev, "bbt_fir_execute: null argument");
    ARG_TRY(n_out >= 0, "bbt_fir_execute: n_out=%lld is negative", (long long)n_out);
    if (n_out == 0) return 0;
    constexpr int R = BBT_FIR_R;
    const long long n_in = n_out + p->n_tap - 1;
    // (one stream: the grid covers the first half of the outputs, a thread's second filter the other half)
    const long long covered = p->S == 1 ? (n_out + 1) / 2 : n_out;
    const long long tiles = (covered + 256 * R - 1) / (256 * R);
    ARG_TRY(tiles * p->npair < (1ll << 31), "bbt_fir_execute: too large for one call");
    const size_t lds = (size_t)R * p->pitch * sizeof(float4);
    const dim3 grid((unsigned)(tiles * p->npair));
    if (p->cplx)
        hipLaunchKernelGGL((k_fir<R, true>), grid, dim3(256), lds, (hipStream_t)stream,
                           (const float2*)in_dev, (float2*)out_dev, n_in, (long long)n_out, p->S,
                           p->tre, p->tim, p->tap_pitch, p->n_chunks, p->pitch);
    else
        hipLaunchKernelGGL((k_fir<R, false>), grid, dim3(256), lds, (hipStream_t)stream,
                           (const float2*)in_dev, (float2*)out_dev, n_in, (long long)n_out, p->S,
                           p->tre, p->tim, p->tap_pitch, p->n_chunks, p->pitch);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// sampler frames -> float32 / complex64 (k_unpack)
extern "C" int bbt_unpack_masked(const void* raw_dev, void* out_dev, int64_t n_frames, int frame_bytes,
                                 int header_bytes, int bits, int samples_per_frame, int n_thread,
                                 int n_elem, int code, const void* valid_dev, bbt_stream stream);
extern "C" int bbt_unpack(const void* raw_dev, void* out_dev, int64_t n_frames, int frame_bytes,
                          int header_bytes, int bits, int samples_per_frame, int n_thread,
                          int n_elem, int code, bbt_stream stream) {
    return bbt_unpack_masked(raw_dev, out_dev, n_frames, frame_bytes, header_bytes, bits,
                             samples_per_frame, n_thread, n_elem, code, nullptr, stream);
}
extern "C" int bbt_unpack_masked(const void* raw_dev, void* out_dev, int64_t n_frames, int frame_bytes,
                                 int header_bytes, int bits, int samples_per_frame, int n_thread,
                                 int n_elem, int code, const void* valid_dev, bbt_stream stream) {
    ARG_TRY(raw_dev && out_dev, "bbt_unpack: null argument");
    ARG_TRY(n_frames >= 0 && n_thread >= 1 && n_frames % n_thread == 0,
            "bbt_unpack: %lld frames are not whole sets of %d threads", (long long)n_frames, n_thread);
    ARG_TRY(code == 0 || code == 1, "bbt_unpack: code must be 0 (VDIF levels) or 1 (two's complement)");
    ARG_TRY(code == 0 ? (bits == 1 || bits == 2 || bits == 4 || bits == 8 || bits == 16)
                      : (bits == 8 || bits == 16),
            "bbt_unpack: %d bits per component are not supported for code %d", bits, code);
    ARG_TRY(frame_bytes > header_bytes && header_bytes >= 0 && frame_bytes % 4 == 0 &&
                header_bytes % 4 == 0,
            "bbt_unpack: frame of %d bytes with a header of %d", frame_bytes, header_bytes);
    ARG_TRY(samples_per_frame >= 1 && n_elem >= 1 &&
                (int64_t)samples_per_frame * n_elem * bits <= (int64_t)(frame_bytes - header_bytes) * 8,
            "bbt_unpack: %d samples of %d components at %d bits do not fit the payload",
            samples_per_frame, n_elem, bits);
    if (n_frames == 0) return 0;
    ARG_TRY(n_frames < (1ll << 31), "bbt_unpack: too many frames for one call");
    const long long per_frame = (long long)samples_per_frame * n_elem;
    ARG_TRY(per_frame * bits < (1ll << 32), "bbt_unpack: frames of %lld components are too long",
            per_frame);
    // components decoded per thread: adjacent in payload and output, 16-byte aligned stores
    const int g = n_elem % 4 == 0 ? 4 : (n_elem % 2 == 0 ? 2 : 1);
    const long long by = (per_frame / g + 256 * BBT_UNPACK_ITER - 1) / (256 * BBT_UNPACK_ITER);
    ARG_TRY(by <= 65535, "bbt_unpack: frames of %lld components are too long", per_frame);
    const dim3 grid((unsigned)n_frames, (unsigned)by);
    int lg_e = -1;
    for (int k = 0; k < 16; ++k)
        if ((1 << k) == n_elem) lg_e = k;
#define BBT_UNPACK_B(G_, B_)                                                                       \
    hipLaunchKernelGGL((k_unpack<G_, B_>), grid, dim3(256), 0, (hipStream_t)stream,                \
                       (const unsigned char*)raw_dev, (float*)out_dev, frame_bytes, header_bytes,  \
                       bits, samples_per_frame, n_thread, n_elem, code, (const unsigned char*)valid_dev, lg_e)
#define BBT_UNPACK(G_)                                  \
    switch (bits) {                                     \
        case 1: BBT_UNPACK_B(G_, 1); break;             \
        case 2: BBT_UNPACK_B(G_, 2); break;             \
        case 4: BBT_UNPACK_B(G_, 4); break;             \
        case 8: BBT_UNPACK_B(G_, 8); break;             \
        default: BBT_UNPACK_B(G_, 0); break;            \
    }
    if (g == 4) { BBT_UNPACK(4) }
    else if (g == 2) { BBT_UNPACK(2) }
    else { BBT_UNPACK(1) }
#undef BBT_UNPACK_B
#undef BBT_UNPACK
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// multi-GPU: one process per GPU, RCCL over xGMI.  The path
// has no data-path collective; these are the two it uses around it: the
// broadcast of the response (chirp) at plan time and the optional all-gather
// of the outputs.  librccl is loaded on first use (dlopen), so the library
// itself loads on hosts without it.
struct bbt_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
};
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail("bbt_comm: cannot load librccl: %s", dlerror());
    RcclApi api;
    api.handle = h;
    api.get_unique_id = (decltype(api.get_unique_id))dlsym(h, "ncclGetUniqueId");
    api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(h, "ncclCommInitRank");
    api.comm_destroy = (decltype(api.comm_destroy))dlsym(h, "ncclCommDestroy");
    api.broadcast = (decltype(api.broadcast))dlsym(h, "ncclBroadcast");
    api.all_gather = (decltype(api.all_gather))dlsym(h, "ncclAllGather");
    api.error_string = (decltype(api.error_string))dlsym(h, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.broadcast ||
        !api.all_gather || !api.error_string)
        return fail("bbt_comm: librccl lacks a required symbol");
    g_rccl = api;
    return 0;
}
}  // namespace

#define RCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) return fail("%s failed: %s", #expr, g_rccl.error_string(r_));   \
    } while (0)

extern "C" {

int bbt_comm_unique_id(void* id_out, size_t id_bytes) {
    ARG_TRY(id_out && id_bytes >= sizeof(ncclUniqueId), "bbt_comm_unique_id: need a buffer of %zu bytes",
            sizeof(ncclUniqueId));
    if (rccl_load()) return 1;
    ncclUniqueId id;
    RCCL_TRY(g_rccl.get_unique_id(&id));
    memcpy(id_out, &id, sizeof id);
    return 0;
}

int bbt_comm_init(bbt_comm** comm, int n_ranks, int rank, const void* id, size_t id_bytes) {
    ARG_TRY(comm && id, "bbt_comm_init: null argument");
    *comm = nullptr;
    ARG_TRY(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bbt_comm_init: rank %d of %d", rank, n_ranks);
    ARG_TRY(id_bytes >= sizeof(ncclUniqueId), "bbt_comm_init: the id must hold %zu bytes",
            sizeof(ncclUniqueId));
    if (rccl_load()) return 1;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    bbt_comm* c = new bbt_comm;
    c->rank = rank;
    c->world = n_ranks;
    ncclResult_t r = g_rccl.comm_init_rank(&c->nccl, n_ranks, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank failed: %s", g_rccl.error_string(r));
    }
    *comm = c;
    return 0;
}

int bbt_comm_destroy(bbt_comm* comm) {
    if (!comm) return 0;
    if (comm->nccl) g_rccl.comm_destroy(comm->nccl);
    delete comm;
    return 0;
}

int bbt_bcast_chirp(bbt_comm* comm, void* resp_dev, int64_t n_complex, int root, bbt_stream stream) {
    ARG_TRY(comm && resp_dev, "bbt_bcast_chirp: null argument");
    ARG_TRY(n_complex >= 0 && root >= 0 && root < comm->world, "bbt_bcast_chirp: bad count or root");
    if (n_complex == 0) return 0;
    RCCL_TRY(g_rccl.broadcast(resp_dev, resp_dev, (size_t)n_complex * 2, ncclFloat32, root, comm->nccl,
                              (hipStream_t)stream));
    return 0;
}

int bbt_gather_output(bbt_comm* comm, const void* send_dev, void* recv_dev, int64_t bytes_per_rank,
                      bbt_stream stream) {
    ARG_TRY(comm && send_dev && recv_dev, "bbt_gather_output: null argument");
    ARG_TRY(bytes_per_rank >= 0, "bbt_gather_output: negative size");
    if (bytes_per_rank == 0) return 0;
    if (bytes_per_rank % 4 == 0)
        RCCL_TRY(g_rccl.all_gather(send_dev, recv_dev, (size_t)bytes_per_rank / 4, ncclFloat32,
                                   comm->nccl, (hipStream_t)stream));
    else
        RCCL_TRY(g_rccl.all_gather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclUint8, comm->nccl,
                                   (hipStream_t)stream));
    return 0;
}

}  // extern "C"
