// l64_jit.cpp -- run-time compilation of the lane-resident stage-1 kernel (l64_kernel.h) for one channel plan.
//
// The FFT graph a plan needs is known when its handle is created: which residues { bin mod 2^s } are live after each of
// the first six stages.  With those masks as compile-time constants the kernel is straight-line code holding exactly the
// plan's butterflies (for the 8-channel plan of BASELINE configs[1]: 152 of the 384 output halves, most of them plain adds).
// hipRTC compiles the same source text hipcc compiles ahead of time for the full graph (embedded below by the Makefile),
// the code object is loaded with the module API and kept for the life of the process, keyed by (device, hop, masks).
// hipRTC is looked up with dlopen: the library has no link-time dependency on it, and where it is missing, or the
// compilation fails, the caller runs the ahead-of-time full-graph instance instead.
//
// A compilation takes 0.3-0.6 s, and the reference's input ring holds 0.5 s of u8 IQ (config.cpp:799-805, overflow rule
// input-helpers.cpp:56-61): it must not happen inside a processing call.  So mi_demod_create() asks for the kernel (init_demod()
// runs before the input threads start, rtl_airband.cpp:1058-1082), and the code object is kept on disk under mi_set_cache_dir() /
// $MI_AIRBAND_CACHE_DIR / $XDG_CACHE_HOME/mi_airband / ~/.cache/mi_airband, keyed by (arch, hop, masks, waves per SIMD, source
// text, compiler options): the next process start loads it in a few milliseconds.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace mi {

struct L64Jit {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    size_t lds_attr = 0;  // dynamic LDS size the function has been told about
    int minwaves = 2;     // waves per SIMD the instance was compiled for (2 or 3 workgroups per CU)
};

namespace {

const char kL64Source[] =
#include "build/l64_src.inc"
    ;

// the few hipRTC entry points used, resolved at run time
struct Rtc {
    void* lib = nullptr;
    int (*create)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*compile)(void*, int, const char**) = nullptr;
    int (*log_size)(void*, size_t*) = nullptr;
    int (*get_log)(void*, char*) = nullptr;
    int (*code_size)(void*, size_t*) = nullptr;
    int (*get_code)(void*, char*) = nullptr;
    int (*destroy)(void**) = nullptr;
    bool ok = false;
};

Rtc& rtc() {
    static Rtc r = [] {
        Rtc x;
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib)
                break;
        }
        if (!x.lib)
            return x;
        auto sym = [&](const char* n) { return dlsym(x.lib, n); };
        x.create = reinterpret_cast<decltype(x.create)>(sym("hiprtcCreateProgram"));
        x.compile = reinterpret_cast<decltype(x.compile)>(sym("hiprtcCompileProgram"));
        x.log_size = reinterpret_cast<decltype(x.log_size)>(sym("hiprtcGetProgramLogSize"));
        x.get_log = reinterpret_cast<decltype(x.get_log)>(sym("hiprtcGetProgramLog"));
        x.code_size = reinterpret_cast<decltype(x.code_size)>(sym("hiprtcGetCodeSize"));
        x.get_code = reinterpret_cast<decltype(x.get_code)>(sym("hiprtcGetCode"));
        x.destroy = reinterpret_cast<decltype(x.destroy)>(sym("hiprtcDestroyProgram"));
        x.ok = x.create && x.compile && x.log_size && x.get_log && x.code_size && x.get_code && x.destroy;
        return x;
    }();
    return r;
}

struct Key {
    int device, hop;
    uint64_t need[6];
    bool operator<(const Key& o) const {
        if (device != o.device)
            return device < o.device;
        if (hop != o.hop)
            return hop < o.hop;
        return std::memcmp(need, o.need, sizeof(need)) < 0;
    }
};
struct Entry {
    L64Jit jit;
    bool usable = false;
    std::string why;
};
std::mutex g_mu;
std::map<Key, Entry*> g_cache;  // entries live until the process ends (their modules too)
std::string g_cache_dir;        // mi_set_cache_dir(); empty = the environment / home default
bool g_cache_dir_set = false;
int g_n_compiled = 0, g_n_from_disk = 0;

uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i)
        h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

// the directory code objects are kept in ("" = nowhere: caching switched off or no usable place)
std::string cache_dir() {
    std::string d;
    if (g_cache_dir_set) {
        d = g_cache_dir;
    } else if (const char* e = std::getenv("MI_AIRBAND_CACHE_DIR")) {
        d = e;
    } else if (const char* x = std::getenv("XDG_CACHE_HOME"); x && *x) {
        d = std::string(x) + "/mi_airband";
    } else if (const char* hme = std::getenv("HOME"); hme && *hme) {
        d = std::string(hme) + "/.cache/mi_airband";
    }
    if (d.empty())
        return d;
    // (mkdir -p of the last two components is all the defaults need)
    const size_t slash = d.find_last_of('/');
    if (slash != std::string::npos && slash > 0)
        (void)mkdir(d.substr(0, slash).c_str(), 0755);
    (void)mkdir(d.c_str(), 0755);
    struct stat st {};
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || access(d.c_str(), W_OK | X_OK) != 0)
        return std::string();
    return d;
}

struct DiskHeader {
    char magic[8];  // "MIL64CO1"
    uint64_t key;   // hash of arch + options + source
    uint64_t size;  // bytes of code that follow
    uint64_t code_hash;
};

bool disk_load(const std::string& path, uint64_t key, std::vector<char>& code) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
        return false;
    DiskHeader hd{};
    bool ok = std::fread(&hd, sizeof(hd), 1, f) == 1 && std::memcmp(hd.magic, "MIL64CO1", 8) == 0 && hd.key == key && hd.size > 0 && hd.size < (64u << 20);
    if (ok) {
        code.resize(hd.size);
        ok = std::fread(code.data(), 1, hd.size, f) == hd.size && fnv1a(code.data(), code.size()) == hd.code_hash;
    }
    std::fclose(f);
    return ok;
}

void disk_store(const std::string& path, uint64_t key, const std::vector<char>& code) {
    // written under a private name and renamed: a concurrent start never sees half a file
    const std::string tmp = path + ".tmp" + std::to_string(static_cast<long>(getpid()));
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f)
        return;
    DiskHeader hd{};
    std::memcpy(hd.magic, "MIL64CO1", 8);
    hd.key = key;
    hd.size = code.size();
    hd.code_hash = fnv1a(code.data(), code.size());
    const bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1 && std::fwrite(code.data(), 1, code.size(), f) == code.size();
    if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path.c_str()) != 0)
        (void)std::remove(tmp.c_str());
}

}  // namespace

const L64Jit* l64_jit_get(int device, int hop, const uint64_t need[6], const char** why) {
    static const char* none = "";
    if (why)
        *why = none;
    Key k{};
    k.device = device;
    k.hop = hop;
    std::memcpy(k.need, need, sizeof(k.need));
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find(k);
    if (it != g_cache.end()) {
        if (why)
            *why = it->second->why.c_str();
        return it->second->usable ? &it->second->jit : nullptr;
    }
    Entry* e = new Entry();
    g_cache[k] = e;
    auto fail = [&](const std::string& msg) -> const L64Jit* {
        e->why = msg;
        if (std::getenv("MI_AIRBAND_DEBUG"))
            std::fprintf(stderr, "mi_airband: lane-resident stage 1 not compiled for this plan (%s); the full-graph instance runs\n", msg.c_str());
        if (why)
            *why = e->why.c_str();
        return nullptr;
    };
    hipDeviceProp_t prop{};
    std::string arch = "gfx950";
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.gcnArchName[0])
        arch = prop.gcnArchName;
    // three workgroups per CU (LDS: 50 KB each) where the plan leaves few enough points live; the compiler keeps within the
    // matching register budget
    const int live = __builtin_popcountll(need[5]) > 16 ? 2 : 3;
    const char* mw = std::getenv("MI_AIRBAND_L64_MINWAVES");
    const int minwaves = (mw && *mw) ? std::max(1, std::min(4, std::atoi(mw))) : live;
    std::vector<std::string> opts = {"--offload-arch=" + arch, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-DMI_L64_JIT=1",
                                     "-DL64_HOP=" + std::to_string(hop), "-DL64_MINWAVES=" + std::to_string(minwaves)};
    for (int s = 0; s < 6; ++s) {
        char buf[64];
        std::snprintf(buf, sizeof(buf), "-DL64_N%d=0x%llxull", s + 1, static_cast<unsigned long long>(need[s]));
        opts.push_back(buf);
    }
    // the code object on disk, if an earlier start left one for exactly this source text and these options
    uint64_t key = fnv1a(kL64Source, sizeof(kL64Source));
    for (const std::string& o : opts)
        key = fnv1a(o.data(), o.size() + 1, key);
    int rt_version = 0;
    (void)hipRuntimeGetVersion(&rt_version);
    key = fnv1a(&rt_version, sizeof(rt_version), key);
    const std::string dir = cache_dir();
    char name[64];
    std::snprintf(name, sizeof(name), "/l64_%016llx.co", static_cast<unsigned long long>(key));
    const std::string path = dir.empty() ? std::string() : dir + name;
    std::vector<char> code;
    bool from_disk = !path.empty() && disk_load(path, key, code);
    if (!from_disk) {
        Rtc& r = rtc();
        if (!r.ok)
            return fail("hipRTC not available");
        void* prog = nullptr;
        if (r.create(&prog, kL64Source, "l64_kernel.hip", 0, nullptr, nullptr) != 0 || !prog)
            return fail("hiprtcCreateProgram failed");
        std::vector<const char*> copts;
        for (const std::string& o : opts)
            copts.push_back(o.c_str());
        const int rc = r.compile(prog, static_cast<int>(copts.size()), copts.data());
        if (rc != 0) {
            size_t n = 0;
            std::string log;
            if (r.log_size(prog, &n) == 0 && n > 1) {
                log.resize(n);
                r.get_log(prog, log.data());
            }
            r.destroy(&prog);
            return fail("hiprtcCompileProgram failed: " + log.substr(0, 2000));
        }
        size_t csz = 0;
        if (r.code_size(prog, &csz) != 0 || csz == 0) {
            r.destroy(&prog);
            return fail("hiprtcGetCodeSize failed");
        }
        code.resize(csz);
        const int gc = r.get_code(prog, code.data());
        r.destroy(&prog);
        if (gc != 0)
            return fail("hiprtcGetCode failed");
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != device && hipSetDevice(device) != hipSuccess)
        return fail("hipSetDevice failed");
    hipError_t he = hipModuleLoadData(&e->jit.mod, code.data());
    if (he == hipSuccess)
        he = hipModuleGetFunction(&e->jit.fn, e->jit.mod, "l64_entry");
    if (cur != device)
        (void)hipSetDevice(cur);
    if (he != hipSuccess) {
        (void)hipGetLastError();
        if (from_disk)
            (void)std::remove(path.c_str());  // (a file this runtime cannot load: the next start compiles again)
        return fail(std::string("loading the compiled kernel failed: ") + hipGetErrorString(he));
    }
    if (from_disk) {
        ++g_n_from_disk;
    } else {
        ++g_n_compiled;
        if (!path.empty())
            disk_store(path, key, code);
    }
    e->jit.minwaves = minwaves;
    e->usable = true;
    if (std::getenv("MI_AIRBAND_DEBUG")) {
        int regs = -1;
        (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, e->jit.fn);
        std::fprintf(stderr, "mi_airband: lane-resident stage 1 %s for this plan: %d live classes, %d waves per SIMD asked for, %d VGPRs\n",
                     from_disk ? ("loaded from " + path).c_str() : "compiled", __builtin_popcountll(need[5]), minwaves, regs);
    }
    return &e->jit;
}

void l64_jit_set_cache_dir(const char* dir) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_cache_dir = dir ? dir : "";
    g_cache_dir_set = true;
}

void l64_jit_counts(int* compiled, int* from_disk) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (compiled)
        *compiled = g_n_compiled;
    if (from_disk)
        *from_disk = g_n_from_disk;
}

int l64_jit_minwaves(const L64Jit* j) {
    return j ? j->minwaves : 2;
}

hipError_t l64_jit_launch(const L64Jit* j, const L64Args& a, unsigned gx, unsigned gy, size_t lds, hipStream_t s) {
    L64Jit* m = const_cast<L64Jit*>(j);
    if (lds > 48 * 1024 && lds > m->lds_attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(m->fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess)
            (void)hipGetLastError();  // (not every runtime wants it for module functions: the launch below decides)
        m->lds_attr = lds;
    }
    L64Args args = a;
    size_t sz = sizeof(args);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(m->fn, gx, gy, 1, 256, 1, 1, static_cast<unsigned>(lds), s, nullptr, cfg);
}

}  // namespace mi
