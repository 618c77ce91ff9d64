// rtc.cpp — run-time compilation of the specialized residual program (residual_jit.cpp) for the device an engine runs on.
//
// hiprtc is loaded on first use (dlopen: a host without it keeps working — such an engine interprets its residual rules with
// residual_kernel, and says so in its warnings). The code object is loaded as a HIP module owned by the engine.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace pwaf {

namespace {
struct Rtc {
    void *lib = nullptr;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
    bool ok = false;
    std::string why;
};

Rtc &rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why = std::string("libhiprtc.so cannot be loaded: ") + dlerror(); return; }
        auto sym = [&](const char *n) { return dlsym(r.lib, n); };
        r.create = (decltype(r.create))sym("hiprtcCreateProgram");
        r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
        r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
        r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
        r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
        r.code = (decltype(r.code))sym("hiprtcGetCode");
        r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
        r.ok = r.create && r.compile && r.log_size && r.log && r.code_size && r.code && r.destroy;
        if (!r.ok) r.why = "libhiprtc.so lacks an entry point";
    });
    return r;
}
}  // namespace

// Compiles `source` for `arch` ("gfx950"; a target id's feature suffix is dropped: the code object then runs under any setting).
// No device is needed (the CPU suite checks that the generated program of its fuzz rules compiles for gfx950).
namespace {
// Code objects by (program text, architecture): the engines of a node (pwaf_node_create: one per device, the same rule set) and a
// re-created engine compile once per process.
struct Compiled {
    std::string source, arch;
    std::vector<char> code;
};
std::mutex g_cache_mu;
std::vector<Compiled> g_cache;  // (a handful of rule sets per process: a linear scan)
constexpr size_t kCacheEntries = 8;
}  // namespace

bool rtc_compile(const std::string &source, const std::string &arch_id, std::vector<char> &code, std::string &why) {
    const std::string arch = arch_id.substr(0, arch_id.find(':'));
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (const Compiled &c : g_cache)
            if (c.arch == arch && c.source == source) { code = c.code; return true; }
    }
    Rtc &r = rtc();
    if (!r.ok) { why = r.why; return false; }
    void *prog = nullptr;
    if (r.create(&prog, source.c_str(), "pwaf_residual.hip", 0, nullptr, nullptr) != 0) { why = "hiprtcCreateProgram failed"; return false; }
    const std::string a = "--offload-arch=" + arch;
    const char *opts[] = {a.c_str(), "-O3", "-std=c++17", "-Wno-pragma-once-outside-header"};
    const int rc = r.compile(prog, 4, opts);
    if (rc != 0) {
        size_t n = 0;
        r.log_size(prog, &n);
        std::string log(n + 1, '\0');
        if (n) r.log(prog, &log[0]);
        why = "hiprtc could not compile the specialized residual program: " + log.substr(0, 2000);
        r.destroy(&prog);
        return false;
    }
    size_t n = 0;
    if (r.code_size(prog, &n) != 0 || n == 0) { why = "hiprtcGetCodeSize failed"; r.destroy(&prog); return false; }
    code.resize(n);
    const int rc2 = r.code(prog, code.data());
    r.destroy(&prog);
    if (rc2 != 0) { why = "hiprtcGetCode failed"; return false; }
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (g_cache.size() >= kCacheEntries) g_cache.erase(g_cache.begin());
        g_cache.push_back(Compiled{source, arch, code});
    }
    return true;
}

bool jit_load(const std::vector<char> &code, JitKernel &out, std::string &why) {
    hipModule_t mod = nullptr;
    hipError_t e = hipModuleLoadData(&mod, code.data());
    if (e != hipSuccess) { why = std::string("hipModuleLoadData: ") + hipGetErrorString(e); return false; }
    hipFunction_t fn = nullptr;
    e = hipModuleGetFunction(&fn, mod, "rvm_jit_kernel");
    if (e != hipSuccess) { why = std::string("hipModuleGetFunction: ") + hipGetErrorString(e); (void)hipModuleUnload(mod); return false; }
    out.module = mod;
    out.function = fn;
    return true;
}

void jit_release(JitKernel &k) {
    if (k.module) (void)hipModuleUnload((hipModule_t)k.module);
    k.module = k.function = nullptr;
}

int launch_residual_jit(const JitKernel &k, const ResidualJitArgs &a, uint32_t n_cus, void *stream) {
    if (a.n == 0 || a.n_rules == 0) return 0;
    ResidualJitArgs args = a;
    size_t size = sizeof args;
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    // one workgroup per 256 requests: the groups are short and uneven (a rule that walks a string next to seven that compare two
    // integers), the dispatcher balances them; a grid capped at the occupancy ran a third of its second round on an idle chip
    (void)n_cus;
    uint32_t wg = 256, blocks = (a.n + 255) / 256;
#ifdef PWAF_PROFILING
    if (const char *w = getenv("PWAF_JIT_WG")) { wg = (uint32_t)atoi(w); blocks = (a.n + wg - 1) / wg; }              // timing experiments
    if (const char *c = getenv("PWAF_JIT_CAP")) blocks = std::min<uint32_t>(blocks, std::max(1u, n_cus) * (uint32_t)atoi(c));
#endif
    return (int)hipModuleLaunchKernel((hipFunction_t)k.function, blocks, 1, 1, wg, 1, 1, 0, (hipStream_t)stream, nullptr, extra);
}

}  // namespace pwaf
