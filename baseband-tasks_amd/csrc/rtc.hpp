// Run-time specialisation of the generic-length kernels (gen2_kernels.hpp): when a plan is made
// for a length that is not a power of two, the library writes a few lines of source -- the
// geometry traits of that length and the entry points -- and compiles them with hipRTC against
// the headers of this directory, exactly as rocFFT builds its kernels for the lengths it meets.
// The compiled code objects are kept per process (and, if BBT_RTC_CACHE names a directory, on
// disk).  hipRTC is loaded with dlopen at first use: the library itself does not link it, and a
// process that never meets such a length never loads it.
//
// Plain host C++ (HIP runtime API only).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

namespace bbt {

struct RtcApi {
    void* handle = nullptr;
    int (*create)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*compile)(void*, int, const char**) = nullptr;
    int (*log_size)(void*, size_t*) = nullptr;
    int (*get_log)(void*, char*) = nullptr;
    int (*code_size)(void*, size_t*) = nullptr;
    int (*get_code)(void*, char*) = nullptr;
    int (*destroy)(void**) = nullptr;
    std::string error;
};

static inline RtcApi& rtc_api() {
    static RtcApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // by soname first: a copy that is already in the process (PyTorch bundles one next to its
        // HIP runtime, and baseband_tasks_amd.hip preloads it with that runtime) is the one to use
        const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
        const char* forced = getenv("BBT_HIPRTC_LIB");
        if (forced && *forced) api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        const char* only = getenv("BBT_HIPRTC_ONLY");      // (tests: no other copy if the named one fails)
        for (const char* nm : names) {
            if (api.handle || (forced && *forced && only && *only == '1')) break;
            api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.handle) {
            const char* why = dlerror();         // (one call: dlerror() clears the message it returns)
            api.error = std::string("cannot load libhiprtc.so: ") + (why ? why : "?");
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(api.handle, s);
            if (!p && api.error.empty()) api.error = std::string("libhiprtc.so lacks ") + s;
            return p;
        };
        api.create = (decltype(api.create))sym("hiprtcCreateProgram");
        api.compile = (decltype(api.compile))sym("hiprtcCompileProgram");
        api.log_size = (decltype(api.log_size))sym("hiprtcGetProgramLogSize");
        api.get_log = (decltype(api.get_log))sym("hiprtcGetProgramLog");
        api.code_size = (decltype(api.code_size))sym("hiprtcGetCodeSize");
        api.get_code = (decltype(api.get_code))sym("hiprtcGetCode");
        api.destroy = (decltype(api.destroy))sym("hiprtcDestroyProgram");
    });
    return api;
}

// Where the headers are: BBT_CSRC, or csrc/ next to the directory of the library this code is in
// (baseband-tasks_amd/lib/libbbt_hip.so -> baseband-tasks_amd/csrc).
static inline std::string rtc_include_dir() {
    const char* env = getenv("BBT_CSRC");
    if (env && *env) return env;
    Dl_info info;
    if (dladdr((const void*)&rtc_api, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t a = p.rfind('/');
        if (a != std::string::npos) {
            std::string dir = p.substr(0, a);              // .../lib  (or .../build for a dev harness)
            const size_t b = dir.rfind('/');
            if (b != std::string::npos) {
                std::string cand = dir.substr(0, b) + "/csrc";
                struct stat st;
                if (stat((cand + "/gen2_kernels.hpp").c_str(), &st) == 0) return cand;
                cand = dir.substr(0, b) + "/baseband-tasks_amd/csrc";
                if (stat((cand + "/gen2_kernels.hpp").c_str(), &st) == 0) return cand;
            }
        }
    }
    return "";
}

static inline std::string rtc_arch() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0])
        return prop.gcnArchName;
    return "gfx950";
}

// Compile `source` to a code object for the current device's architecture.  0 = ok.
// `cached` (optional) in: false = do not read the disk cache (its file is then written anew);
// out: whether the code object came from there.
static inline int rtc_compile(const std::string& source, std::vector<char>* code, std::string* log,
                              bool* cached = nullptr) {
    const bool read_cache = cached == nullptr || *cached;
    if (cached) *cached = false;
    RtcApi& api = rtc_api();
    if (!api.error.empty()) {
        *log = api.error;
        return 1;
    }
    const std::string inc = rtc_include_dir();
    if (inc.empty()) {
        *log = "the kernel headers (csrc/gen2_kernels.hpp) were not found: set BBT_CSRC";
        return 1;
    }
    // optional disk cache, keyed by a hash of everything that determines the code object
    std::string cache_file;
    const std::string arch = rtc_arch();
    const std::string defines = getenv("BBT_RTC_DEFINES") ? getenv("BBT_RTC_DEFINES") : "";
    if (const char* dir = getenv("BBT_RTC_CACHE")) {
        if (*dir) {
            struct stat st;
            std::string stamp;
            for (const char* h : {"fft_gen2.hpp", "gen2_kernels.hpp", "gen_functors.hpp", "fft_generic.hpp", "fft_core.hpp",
                                  "osm_chunk.hpp"})
                if (stat((inc + "/" + h).c_str(), &st) == 0)
                    stamp += std::to_string((long long)st.st_mtime) + ":" + std::to_string((long long)st.st_size) + ";";
            const size_t key = std::hash<std::string>()(source + "|" + arch + "|" + defines + "|" + stamp);
            char name[64];
            snprintf(name, sizeof name, "/bbt_g2_%016zx.co", key);
            cache_file = std::string(dir) + name;
            std::ifstream f(cache_file, std::ios::binary);
            if (f && read_cache) {
                code->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
                if (!code->empty()) {
                    if (cached) *cached = true;
                    return 0;
                }
            }
        }
    }
    void* prog = nullptr;
    if (api.create(&prog, source.c_str(), "bbt_g2.hip", 0, nullptr, nullptr) != 0) {
        *log = "hiprtcCreateProgram failed";
        return 1;
    }
    const std::string o_arch = "--offload-arch=" + arch, o_inc = "-I" + inc;
    // (-simplifycfg-sink-common=false: sinking the common tails of unrolled butterflies turns
    // static register indices into selected ones, and a thread's points land in scratch)
    std::vector<const char*> opts = {o_arch.c_str(), o_inc.c_str(), "-O3", "-std=c++17", "-Wno-unused-value",
                                     "-mllvm", "-simplifycfg-sink-common=false"};
    std::vector<std::string> extra;      // dev: BBT_RTC_DEFINES="-DBBT_G2_NT_LOAD=1 -D..." (part of the cache key)
    {
        std::istringstream in(defines);
        std::string tok;
        while (in >> tok) extra.push_back(tok);
        for (const std::string& t : extra) opts.push_back(t.c_str());
    }
    const int rc = api.compile(prog, (int)opts.size(), opts.data());
    size_t n = 0;
    if (api.log_size(prog, &n) == 0 && n > 1) {
        log->resize(n);
        api.get_log(prog, &(*log)[0]);
    }
    if (rc != 0) {
        api.destroy(&prog);
        if (log->empty()) *log = "hiprtcCompileProgram failed";
        return 1;
    }
    if (api.code_size(prog, &n) != 0 || n == 0) {
        api.destroy(&prog);
        *log = "hiprtcGetCodeSize failed";
        return 1;
    }
    code->resize(n);
    api.get_code(prog, code->data());
    api.destroy(&prog);
    if (!cache_file.empty()) {
        const std::string tmp = cache_file + ".tmp" + std::to_string((long long)getpid());
        std::ofstream f(tmp, std::ios::binary);
        if (f) {
            f.write(code->data(), (std::streamsize)code->size());
            f.close();
            // (a short write -- disk full -- must not become a cache entry)
            if (f.fail() || rename(tmp.c_str(), cache_file.c_str()) != 0) unlink(tmp.c_str());
        }
    }
    return 0;
}

// A compiled translation unit loaded on one device, and its entry points by name.
struct RtcModule {
    hipModule_t mod = nullptr;
    std::map<std::string, hipFunction_t> fn;
};

// source text -> module on the current device (kept for the life of the process).  0 = ok.
static inline int rtc_module(const std::string& source, RtcModule** out, std::string* log) {
    static std::mutex mu;
    static std::map<std::pair<int, std::string>, RtcModule*> modules;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        *log = "hipGetDevice failed";
        return 1;
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = modules.find({dev, source});
    if (it != modules.end()) {
        *out = it->second;
        return 0;
    }
    std::vector<char> code;
    bool cached = true;
    if (rtc_compile(source, &code, log, &cached)) return 1;
    RtcModule* m = new RtcModule;
    hipError_t e = hipModuleLoadData(&m->mod, code.data());
    if (e != hipSuccess && cached) {
        // a damaged or foreign file in BBT_RTC_CACHE: compile again (which replaces the file)
        (void)hipGetLastError();
        cached = false;
        code.clear();
        if (rtc_compile(source, &code, log, &cached)) {
            delete m;
            return 1;
        }
        e = hipModuleLoadData(&m->mod, code.data());
    }
    if (e != hipSuccess) {
        *log = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
        delete m;
        return 1;
    }
    modules[{dev, source}] = m;
    *out = m;
    return 0;
}
static inline int rtc_function(RtcModule* m, const char* name, hipFunction_t* fn, std::string* log) {
    static std::mutex mu;               // (host threads that make plans of one length at the same time)
    std::lock_guard<std::mutex> lock(mu);
    auto it = m->fn.find(name);
    if (it == m->fn.end()) {
        hipFunction_t f;
        const hipError_t e = hipModuleGetFunction(&f, m->mod, name);
        if (e != hipSuccess) {
            *log = std::string("hipModuleGetFunction(") + name + "): " + hipGetErrorString(e);
            return 1;
        }
        it = m->fn.emplace(name, f).first;
    }
    *fn = it->second;
    return 0;
}

}  // namespace bbt
