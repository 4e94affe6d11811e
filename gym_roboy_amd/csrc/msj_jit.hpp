// msj_jit.hpp - run-time specialisation of the env-per-lane kernels (msj_kernels.hpp) to ONE robot: the robot's
// closed-form constants are written into the source as hex-float literals and the kernels compiled with hiprtc,
// exactly what msj_baked.hpp does ahead of time for MsjRobot (+6 % on the RK4 step: no scalar loads in the tendon
// loop, sub-expressions shared between tendons, literal addends).  Host code.
//
// hiprtc is loaded with dlopen, so the library has no link-time dependency on it; if it is missing, or the
// compilation fails, the handle keeps running the kernarg instances (an optimisation, not a fallback of the
// physics: both are the same source).  ROBOY_SIM_JIT=0 switches it off.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "msj_math.hpp"

namespace rbj {

struct Module {
    hipModule_t mod = nullptr;
    hipFunction_t step[2] = {nullptr, nullptr};   // msj_step_env_per_lane<0, 256, 8, true> / msj_step_env_per_lane_rs<1, 256, true>
    hipFunction_t env[2] = {nullptr, nullptr};    // msj_env_step_kernel<INTEG, 256, 8 / 4, Const8, true>
    hipFunction_t rollout[2] = {nullptr, nullptr};   // msj_rollout_fused<INTEG, 256, 4, true>
};

struct Rtc {
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcAddNameExpression) add_name = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetLoweredName) lowered = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcVersion) version = nullptr;
    bool ok = false;
};

inline const Rtc &rtc() {
    static Rtc r = [] {
        Rtc x;
        void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
#define RBJ_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(h, name))
        RBJ_SYM(create, "hiprtcCreateProgram"); RBJ_SYM(add_name, "hiprtcAddNameExpression");
        RBJ_SYM(compile, "hiprtcCompileProgram"); RBJ_SYM(log_size, "hiprtcGetProgramLogSize");
        RBJ_SYM(log, "hiprtcGetProgramLog"); RBJ_SYM(lowered, "hiprtcGetLoweredName");
        RBJ_SYM(code_size, "hiprtcGetCodeSize"); RBJ_SYM(code, "hiprtcGetCode"); RBJ_SYM(destroy, "hiprtcDestroyProgram");
        RBJ_SYM(version, "hiprtcVersion");
#undef RBJ_SYM
        x.ok = x.create && x.add_name && x.compile && x.log_size && x.log && x.lowered && x.code_size && x.code && x.destroy;
        return x;
    }();
    return r;
}

// the aggregate initialiser of an MsjConst<float, 8>, every float as an exact hex literal
inline std::string table_text(const rb::MsjConst<float, 8> &c) {
    std::string s = "{{";
    char buf[64];
    auto f = [&](float v) { std::snprintf(buf, sizeof buf, "%af,", double(v)); s += buf; };
    for (int k = 0; k < 8; ++k) {
        const auto &t = c.ten[k];
        s += "{{"; f(t.A[0]); f(t.A[1]); f(t.A[2]); s += "},{"; f(t.Bv[0]); f(t.Bv[1]); f(t.Bv[2]); s += "},{";
        f(t.B2[0]); f(t.B2[1]); f(t.B2[2]); s += "},";
        f(t.ab2); f(t.il0s); f(t.elcs); f(t.ksg); f(t.fmaxv); s += "{0.0f,0.0f}},";
    }
    s += "},{"; for (int a = 0; a < 6; ++a) f(c.IO[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.mc[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.g[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.arm[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.damp[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.qlo[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.qhi[a]);
    s += "},{"; for (int a = 0; a < 3; ++a) f(c.qdmax[a]);
    s += "},"; f(c.kps); f(c.pe_k2s); f(c.inv_pe_den); f(c.fv_c1l); f(c.fv_c2l); f(c.fv_c2s); f(c.fv_k); f(c.h);
    s += std::to_string(c.nsub) + "," + std::to_string(c.simple) + "," + std::to_string(c.nt) + "}";
    return s;
}

// directory of this shared library (the kernel headers travel next to it)
inline std::string library_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void *>(&library_dir), &info) || !info.dli_fname) return ".";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

inline bool enabled() {
    const char *e = std::getenv("ROBOY_SIM_JIT");
    return !(e && e[0] == '0');
}

// ---- code-object cache: a robot's specialised kernels are compiled once per (source text, library build, device arch) ----
// The generated joint-tree source of a 20-joint robot takes hiprtc 10-80 s; the code object (a few hundred KB) is kept under
// ROBOY_SIM_JIT_CACHE (default $XDG_CACHE_HOME/gym_roboy_amd or ~/.cache/gym_roboy_amd; "0" switches the cache off) as
// <key>.rbjc = magic, the lowered kernel names, the code.  The key covers everything that decides the code: the source text
// (robot constants included), the options, the architecture, and this library's file size + modification time (the kernel
// headers the source includes travel with the library and change only when it is rebuilt).  Written to a temporary name
// and renamed, so concurrent ranks never see half a file; anything unreadable or inconsistent is ignored and rebuilt.
struct CacheStats { std::atomic<long long> hits{0}, compiles{0}, stores{0}; };
inline CacheStats &cache_stats() { static CacheStats s; return s; }

inline std::string cache_dir() {
    const char *e = std::getenv("ROBOY_SIM_JIT_CACHE");
    std::string d;
    if (e) {
        if (!e[0] || (e[0] == '0' && !e[1])) return "";
        d = e;
    } else {
        const char *x = std::getenv("XDG_CACHE_HOME"), *h = std::getenv("HOME");
        if (x && x[0]) d = std::string(x) + "/gym_roboy_amd";
        else if (h && h[0]) { (void)mkdir((std::string(h) + "/.cache").c_str(), 0700); d = std::string(h) + "/.cache/gym_roboy_amd"; }
        else return "";
    }
    // The files are code objects this process will load and run: the directory is created for the owner alone and is used
    // only if it belongs to the current user and nobody else can write to it (a shared or planted directory is ignored:
    // the kernels are then compiled, never loaded from disk).  Never pruned: remove the directory to clear it.
    (void)mkdir(d.c_str(), 0700);
    struct stat st;
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return "";
    if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return "";
    return d;
}

inline uint64_t fnv1a64(const void *data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t k = 0; k < n; ++k) { h ^= p[k]; h *= 1099511628211ull; }
    return h;
}

inline std::string cache_key(const std::string &src, const std::string &arch, const char *const *names, int nk, const std::string &options) {
    uint64_t h = fnv1a64(src.data(), src.size());
    h = fnv1a64(arch.data(), arch.size(), h);
    h = fnv1a64(options.data(), options.size(), h);
    // the compiler and the runtime that will load the code object: a toolchain upgrade must not be served stale code
    int ver[4] = {0, 0, 0, 0};
    if (rtc().version) (void)rtc().version(&ver[0], &ver[1]);
    (void)hipRuntimeGetVersion(&ver[2]);
    (void)hipDriverGetVersion(&ver[3]);
    h = fnv1a64(ver, sizeof ver, h);
    for (int k = 0; k < nk; ++k) h = fnv1a64(names[k], std::strlen(names[k]) + 1, h);
    Dl_info info;
    struct stat st;
    if (dladdr(reinterpret_cast<const void *>(&cache_dir), &info) && info.dli_fname && stat(info.dli_fname, &st) == 0) {
        const long long stamp[3] = {static_cast<long long>(st.st_size), static_cast<long long>(st.st_mtim.tv_sec), static_cast<long long>(st.st_mtim.tv_nsec)};
        h = fnv1a64(stamp, sizeof stamp, h);
    }
    char buf[40];
    std::snprintf(buf, sizeof buf, "%016llx-%zu", static_cast<unsigned long long>(h), src.size());
    return buf;
}

constexpr char CACHE_MAGIC[8] = {'R', 'B', 'J', 'C', '0', '0', '1', '\n'};

inline bool cache_load(const std::string &path, int nk, std::vector<std::string> &lowered, std::string &code) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    struct stat fst;
    if (fstat(fileno(f), &fst) != 0 || fst.st_uid != geteuid() || !S_ISREG(fst.st_mode) || (fst.st_mode & (S_IWGRP | S_IWOTH))) {
        std::fclose(f);                      // not this user's file, or writable by others: never load it
        return false;
    }
    bool ok = false;
    char magic[8];
    uint64_t n_names = 0, n_code = 0;
    if (std::fread(magic, 1, 8, f) == 8 && !std::memcmp(magic, CACHE_MAGIC, 8) && std::fread(&n_names, 8, 1, f) == 1 && n_names == uint64_t(nk)) {
        ok = true;
        lowered.assign(nk, "");
        for (int k = 0; k < nk && ok; ++k) {
            uint64_t len = 0;
            ok = std::fread(&len, 8, 1, f) == 1 && len > 0 && len < 4096;
            if (ok) { lowered[k].resize(len); ok = std::fread(&lowered[k][0], 1, len, f) == len; }
        }
        ok = ok && std::fread(&n_code, 8, 1, f) == 1 && n_code > 0 && n_code < (1ull << 30);
        if (ok) {
            code.resize(n_code);
            uint64_t sum = 0;
            ok = std::fread(&code[0], 1, n_code, f) == n_code && std::fread(&sum, 8, 1, f) == 1 && sum == fnv1a64(code.data(), code.size());
        }
    }
    std::fclose(f);
    return ok;
}

inline void cache_store(const std::string &path, const std::vector<std::string> &lowered, const std::string &code) {
    const std::string tmp = path + ".tmp." + std::to_string(static_cast<long long>(getpid()));
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600);
    FILE *f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) ::close(fd); return; }
    const uint64_t n_names = lowered.size(), n_code = code.size(), sum = fnv1a64(code.data(), code.size());
    bool ok = std::fwrite(CACHE_MAGIC, 1, 8, f) == 8 && std::fwrite(&n_names, 8, 1, f) == 1;
    for (const std::string &l : lowered) {
        const uint64_t len = l.size();
        ok = ok && std::fwrite(&len, 8, 1, f) == 1 && std::fwrite(l.data(), 1, len, f) == len;
    }
    ok = ok && std::fwrite(&n_code, 8, 1, f) == 1 && std::fwrite(code.data(), 1, n_code, f) == n_code && std::fwrite(&sum, 8, 1, f) == 1;
    ok = (std::fclose(f) == 0) && ok;
    if (ok && std::rename(tmp.c_str(), path.c_str()) == 0) cache_stats().stores++;
    else (void)std::remove(tmp.c_str());
}

inline bool load_code(const std::string &code, const std::vector<std::string> &lowered, int nk, hipModule_t &mod, hipFunction_t **slots, std::string &why) {
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { why = "hipModuleLoadData failed"; mod = nullptr; return false; }
    for (int k = 0; k < nk; ++k)
        if (hipModuleGetFunction(slots[k], mod, lowered[k].c_str()) != hipSuccess) {
            why = "hipModuleGetFunction failed for " + lowered[k];
            (void)hipModuleUnload(mod);
            mod = nullptr;
            return false;
        }
    return true;
}

// Compile `src` with hiprtc for the current device, load the code object and look up the kernels `names`
// (C++ name expressions, e.g. template instances); false (with a message) if anything along the way is not
// available.  A failed compilation also prints the head of the hiprtc log to stderr, once per process: a lost
// specialisation must not go unnoticed in a bench run.
inline bool compile_and_load(const std::string &src, const char *file_name, const char *const *names, int nk,
                             hipModule_t &mod, hipFunction_t **slots, std::string &why) {
    int dev = 0;
    hipDeviceProp_t prop;
    std::string arch = "--offload-arch=gfx950";
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
        std::string a = prop.gcnArchName;                 // "gfx950:sramecc+:xnack-"
        arch = "--offload-arch=" + a.substr(0, a.find(':'));
    }
    const std::string dir = cache_dir();
    const std::string opt_text = "-O3 -std=c++17 -fno-slp-vectorize";
    const std::string cached = dir.empty() ? "" : dir + "/" + cache_key(src, arch, names, nk, opt_text) + ".rbjc";
    if (!cached.empty()) {
        std::vector<std::string> lowered;
        std::string code;
        if (cache_load(cached, nk, lowered, code) && load_code(code, lowered, nk, mod, slots, why)) { cache_stats().hits++; return true; }
    }
    const Rtc &r = rtc();
    if (!r.ok) { why = "hiprtc is not available"; return false; }
    hiprtcProgram prog = nullptr;
    if (r.create(&prog, src.c_str(), file_name, 0, nullptr, nullptr) != HIPRTC_SUCCESS) { why = "hiprtcCreateProgram failed"; return false; }
    for (int k = 0; k < nk; ++k) r.add_name(prog, names[k]);
    const std::string inc = "-I" + library_dir();
    cache_stats().compiles++;
    const char *opts[] = {arch.c_str(), "-O3", "-std=c++17", "-fno-slp-vectorize", inc.c_str()};      // (opt_text above: part of the cache key)
    const hiprtcResult rc = r.compile(prog, 5, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        r.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) r.log(prog, &log[0]);
        why = "hiprtc compilation failed: " + log.substr(0, 600);
        static bool reported = false;
        if (!reported) { reported = true; std::fprintf(stderr, "roboy_sim: %s: %s\n", file_name, why.c_str()); }
        r.destroy(&prog);
        return false;
    }
    size_t n = 0;
    r.code_size(prog, &n);
    std::string code(n, '\0');
    r.code(prog, &code[0]);
    std::vector<std::string> lowered(nk);
    for (int k = 0; k < nk; ++k) {
        const char *low = nullptr;
        if (r.lowered(prog, names[k], &low) != HIPRTC_SUCCESS || !low) { why = "hiprtcGetLoweredName failed"; r.destroy(&prog); return false; }
        lowered[k] = low;
    }
    r.destroy(&prog);
    if (!load_code(code, lowered, nk, mod, slots, why)) return false;
    if (!cached.empty()) cache_store(cached, lowered, code);
    return true;
}

// the env-per-lane ball-joint kernels on this robot's constants
inline bool build(const rb::MsjConst<float, 8> &c, Module &out, std::string &why) {
    const std::string src = "#define RB_JIT_TABLE " + table_text(c) + "\n#include \"msj_kernels.hpp\"\n";
    constexpr int NK = 6;
    // (Euler: tendon loop fully unrolled, the constants become literals; RK4: the rolled-stages form, 9 = rbk::RS -
    // roboy_sim.hip: RB_BAKED_UNROLL_*)
    const char *names[NK] = {"rbk::msj_step_env_per_lane<0, 256, 8, true>", "rbk::msj_step_env_per_lane_rs<1, 256, true>",
                             "rbk::msj_env_step_kernel<0, 256, 8, rbk::Const8, true>",
                             "rbk::msj_env_step_kernel<1, 256, 9, rbk::Const8, true>",
                             "rbk::msj_rollout_fused<0, 256, 8, true>", "rbk::msj_rollout_fused<1, 256, 9, true>"};
    hipFunction_t *slots[NK] = {&out.step[0], &out.step[1], &out.env[0], &out.env[1], &out.rollout[0], &out.rollout[1]};
    if (!compile_and_load(src, "roboy_msj_jit.hip", names, NK, out.mod, slots, why)) { out = Module(); return false; }
    return true;
}

inline void unload(Module &m) {
    if (m.mod) (void)hipModuleUnload(m.mod);
    m = Module();
}

}  // namespace rbj
