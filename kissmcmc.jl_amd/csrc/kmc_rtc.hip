// kmc_rtc.hip -- user-supplied log-densities compiled at run time (hiprtc) into the same kernels as the menu densities
// (the reference's arbitrary closure `pdf(theta)`, src/samplers.jl:257), with a disk cache of the code objects.
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <csignal>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <regex>
#include <sstream>

#include "kmc_sampler.hpp"
#include "kmc_recognise.hpp"

using namespace kmc;
using namespace kmc_host;

extern "C" char** environ;

namespace {
uint64_t fnv1a(uint64_t h, const void* data, size_t n)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
// The ROCm installation's offline compiler as a child process, instead of hiprtc in this process.  Why: inside a PyTorch process
// hiprtc resolves to the comgr the torch wheel bundles (an older compiler: `dlopen("libamd_comgr.so.3")` by name finds the copy torch
// loaded first), whose code for these kernels was measured 8-20 % slower than what the installation's own clang emits for the same
// source (profiles/NOTES.md round 4).  A compile costs ~1.2 s instead of ~0.4 s, once per (density, kernel geometry): the code objects
// are cached on disk like hiprtc's.  KMC_DEBUG=rtc=hipcc asks for it (any compile), KMC_DEBUG=rtc=hiprtc forbids it; unset: the samplers ask
// for it for big ensembles only (kmc_sampler_create: >= 16 384 walkers, multi-launch kernels), where it pays.
thread_local int g_offline_wanted = 0;
// A profiler's environment: its preloaded library initialises the GPU in every process that inherits it, and hipcc is a driver that
// execs clang and lld -- the launcher-hop-under-the-profiler pattern this pool's machines do not survive.  Two guards: under a profiler
// no child compiler is started at all (in-process hiprtc instead), and a child never inherits these variables (child_environment).
constexpr const char* kToolLibs[] = {"rocprof", "roctracer", "rocprofiler", "librocp", "omnitrace", "rocsys"};
bool tool_library(const std::string& path)
{
    for (const char* lib : kToolLibs)
        if (path.find(lib) != std::string::npos) return true;
    return false;
}
bool tool_variable(const char* entry)
{
    for (const char* prefix : {"HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCP_", "ROCPROFILER_", "ROCPROF", "ROCTRACER_", "HSA_VEN_AMD_AQLPROFILE", "AQLPROFILE_"})
        if (std::strncmp(entry, prefix, std::strlen(prefix)) == 0) return true;
    return false;
}
// Any tool library the HSA runtime is told to load counts (HSA_TOOLS_LIB / ROCP_TOOL_LIBRARIES non-empty, whatever its name); of LD_PRELOAD only
// the entries that name a known tool.  KMC_DEBUG=rtc=hiprtc is the manual override for a tool this does not recognise.
bool under_a_profiler()
{
    for (const char* var : {"HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR"})
        if (const char* v = std::getenv(var)) if (*v) return true;
    const char* v = std::getenv("LD_PRELOAD");
    return v && tool_library(v);
}
bool offline_compiler_wanted()
{
    static const bool profiled = under_a_profiler();
    if (profiled) return false;                          // (whatever KMC_DEBUG=rtc says: the child compiler is never started under a tool)
    std::string e;
    if (debug_opt("rtc", &e)) {
        if (e == "hiprtc") return false;
        if (e == "hipcc") return true;
    }
    return g_offline_wanted != 0;
}
// the parent's environment without any tool variable, LD_PRELOAD without its tool entries (an allocator or a sanitizer runtime stays):
// what a child compiler is started with.  `preload` owns the rewritten LD_PRELOAD entry.
std::vector<char*> child_environment(std::string* preload)
{
    std::vector<char*> env;
    for (char** e = environ; e && *e; ++e) {
        if (tool_variable(*e)) continue;
        if (std::strncmp(*e, "LD_PRELOAD=", 11) != 0) { env.push_back(*e); continue; }
        std::string kept;
        for (const char* p = *e + 11; *p;) {                                    // entries separated by ':' or ' ' (ld.so(8))
            const char* q = p;
            while (*q && *q != ':' && *q != ' ') ++q;
            const std::string item(p, q);
            if (!item.empty() && !tool_library(item)) kept += (kept.empty() ? "" : ":") + item;
            p = *q ? q + 1 : q;
        }
        if (kept.empty()) continue;
        *preload = "LD_PRELOAD=" + kept;
        env.push_back(&(*preload)[0]);
    }
    env.push_back(nullptr);
    return env;
}
std::string find_hipcc()
{
    for (const char* var : {"HIP_PATH", "ROCM_PATH"})
        if (const char* r = std::getenv(var)) { const std::string p = std::string(r) + "/bin/hipcc"; if (::access(p.c_str(), X_OK) == 0) return p; }
    if (::access("/opt/rocm/bin/hipcc", X_OK) == 0) return "/opt/rocm/bin/hipcc";
    return std::string();
}
// KMC_OK: *code holds the object; KMC_ERR_BAD_ARG: the compiler ran and rejected the program (*log); anything else: could not run it
kmc_status compile_offline(const std::string& text, int nheaders, const char* const* header_text, const char* const* header_names,
                           int nopts, const char* const* opts, std::vector<char>* code, std::string* log)
{
    static const std::string hipcc = find_hipcc();
    if (hipcc.empty()) return KMC_ERR_UNSUPPORTED;
    char tmpl[] = "/tmp/kmc_rtc_XXXXXX";
    const char* dir = ::mkdtemp(tmpl);
    if (!dir) return KMC_ERR_UNSUPPORTED;
    const std::string d(dir);
    auto put = [&](const std::string& name, const char* body, size_t n) { std::ofstream o(d + "/" + name, std::ios::binary); o.write(body, (std::streamsize)n); return (bool)o; };
    bool ok = put("prog.hip", text.data(), text.size());
    for (int i = 0; i < nheaders && ok; ++i) ok = put(header_names[i], header_text[i], std::strlen(header_text[i]));
    kmc_status st = KMC_ERR_UNSUPPORTED;
    if (ok) {
        std::vector<std::string> args = {hipcc};
        for (int i = 0; i < nopts; ++i) args.push_back(opts[i]);
        for (const char* a : {"--cuda-device-only", "--no-gpu-bundle-output", "-Wno-unused-command-line-argument", "-c"}) args.push_back(a);
        args.push_back("-I" + d);
        args.push_back(d + "/prog.hip");
        args.push_back("-o");
        args.push_back(d + "/prog.co");
        std::vector<char*> argv;
        for (std::string& a : args) argv.push_back(&a[0]);
        argv.push_back(nullptr);
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        const std::string errf = d + "/err.txt";
        posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
        posix_spawn_file_actions_addopen(&fa, 2, errf.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
        pid_t pid = 0;
        std::string preload;
        std::vector<char*> envp = child_environment(&preload);
        const int rc = posix_spawn(&pid, hipcc.c_str(), &fa, nullptr, argv.data(), envp.data());     // a CHILD process: this one is never replaced
        posix_spawn_file_actions_destroy(&fa);
        if (rc == 0) {
            int status = 0;
            // (bounded: a compiler that does not come back within two minutes is killed and hiprtc takes over)
            bool done = false, reaped_elsewhere = false, not_ours = false;
            for (int tick = 0; tick < 12000 && !done; ++tick) {
                const pid_t w = ::waitpid(pid, &status, WNOHANG);
                if (w == pid) done = true;
                else if (w < 0 && errno == ECHILD) {
                    // the host reaps children itself (SIGCHLD set to SIG_IGN, or a handler that waits for any child): no status to
                    // be had, and the pid may already be somebody else's -- never signal it; the object file says how the compile went
                    not_ours = true;
                    if (::kill(pid, 0) != 0) { reaped_elsewhere = done = true; }
                    else ::usleep(10000);
                }
                else if (w < 0 && errno != EINTR) break;
                else ::usleep(10000);
            }
            if (reaped_elsewhere) status = (::access((d + "/prog.co").c_str(), R_OK) == 0) ? 0 : 0x0100;
            else if (!done && not_ours) status = 0x7f00;                      // (still a process of that number after two minutes: not known to be ours)
            else if (!done) { (void)::kill(pid, SIGKILL); while (::waitpid(pid, &status, 0) < 0 && errno == EINTR) {} status = 0x7f00; }
            if (WIFEXITED(status) && WEXITSTATUS(status) == 0) {
                std::ifstream f(d + "/prog.co", std::ios::binary | std::ios::ate);
                const std::streamsize n = f ? (std::streamsize)f.tellg() : 0;
                if (n > 64) { code->resize((size_t)n); f.seekg(0); if (f.read(code->data(), n)) st = KMC_OK; }
            } else if (WIFEXITED(status) && WEXITSTATUS(status) == 1) {       // clang's status for diagnostics
                *log = kmc_host::read_file(errf);
                st = KMC_ERR_BAD_ARG;
            }
        }
    }
    for (const char* n : {"prog.hip", "prog.co", "err.txt"}) (void)::unlink((d + "/" + n).c_str());
    for (int i = 0; i < nheaders; ++i) (void)::unlink((d + "/" + header_names[i]).c_str());
    (void)::rmdir(dir);
    return st;
}
std::string rtc_cache_dir()
{
    if (const char* d = std::getenv("KMC_CACHE_DIR")) return std::strcmp(d, "off") == 0 ? std::string() : std::string(d);     // KMC_CACHE_DIR=off: no disk cache
    if (const char* x = std::getenv("XDG_CACHE_HOME")) if (x[0]) return std::string(x) + "/kissmcmc_hip";
    if (const char* h = std::getenv("HOME")) if (h[0]) return std::string(h) + "/.cache/kissmcmc_hip";
    return std::string();
}
}  // namespace

kmc_status kmc_host::rtc_compile_cached(const std::string& text, const char* program_name, int nheaders, const char* const* header_text,
                                        const char* const* header_names, int nopts, const char* const* opts, std::vector<char>* code, std::string* log)
{
    // key
    uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
    auto mix = [&](const void* p, size_t n) { h1 = fnv1a(h1, p, n); h2 = fnv1a(h2 ^ (uint64_t)n, p, n); h2 = (h2 << 7) | (h2 >> 57); };
    mix(text.data(), text.size());
    for (int i = 0; i < nheaders; ++i) { mix(header_names[i], std::strlen(header_names[i])); mix(header_text[i], std::strlen(header_text[i])); }
    for (int i = 0; i < nopts; ++i) mix(opts[i], std::strlen(opts[i]));
    int vmaj = 0, vmin = 0;
    (void)hiprtcVersion(&vmaj, &vmin);
    mix(&vmaj, sizeof(vmaj)); mix(&vmin, sizeof(vmin));
    int vrt = 0, vdrv = 0;                                  // ... and the runtime / driver builds (a patch update keeps hiprtc's major.minor)
    if (hipRuntimeGetVersion(&vrt) != hipSuccess) (void)hipGetLastError();
    if (hipDriverGetVersion(&vdrv) != hipSuccess) (void)hipGetLastError();
    mix(&vrt, sizeof(vrt)); mix(&vdrv, sizeof(vdrv));
    const int offline = offline_compiler_wanted() ? 1 : 0;        // (the two compilers' objects are not interchangeable in the cache)
    mix(&offline, sizeof(offline));
    const std::string dir = rtc_cache_dir();
    char name[64];
    std::snprintf(name, sizeof(name), "/%016llx%016llx.co", (unsigned long long)h1, (unsigned long long)h2);
    const std::string path = dir.empty() ? std::string() : dir + name;
    if (!path.empty()) {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (f) {
            const std::streamsize n = f.tellg();
            if (n > 64) {
                code->resize((size_t)n);
                f.seekg(0);
                const bool read_ok = f.read(code->data(), n) && f.gcount() == n;
                const bool elf = read_ok && std::memcmp(code->data(), "\x7f" "ELF", 4) == 0;
                const bool bundle = read_ok && std::memcmp(code->data(), "__CLANG_OFFLOAD_BUNDLE__", 24) == 0;
                if (elf || bundle) return KMC_OK;                                       // a code object as hiprtc gave it
            }
            code->clear();
        }
    }
    if (offline_compiler_wanted()) {
        const kmc_status ost = compile_offline(text, nheaders, header_text, header_names, nopts, opts, code, log);
        if (ost == KMC_ERR_BAD_ARG) return ost;                       // the program does not compile: the caller words the message
        if (ost != KMC_OK) code->clear();                             // no offline compiler here (or it failed to run): hiprtc below
    }
    if (code->empty()) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, text.c_str(), program_name, nheaders, const_cast<const char**>(header_text), const_cast<const char**>(header_names)) != HIPRTC_SUCCESS)
        return fail(KMC_ERR_HIP, "hiprtcCreateProgram failed");
    const hiprtcResult r = hiprtcCompileProgram(prog, nopts, const_cast<const char**>(opts));
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        log->assign(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &(*log)[0]);
        hiprtcDestroyProgram(&prog);
        return KMC_ERR_BAD_ARG;                          // the caller words the message
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    code->resize(n);
    hiprtcGetCode(prog, code->data());
    hiprtcDestroyProgram(&prog);
    }
    if (!path.empty()) {                                 // best effort: write beside, then rename (concurrent processes: last one wins, same bytes)
        (void)::mkdir(dir.substr(0, dir.find_last_of('/')).c_str(), 0755);
        (void)::mkdir(dir.c_str(), 0755);
        char tmp[96];
        std::snprintf(tmp, sizeof(tmp), ".tmp.%ld", (long)::getpid());
        const std::string tpath = path + tmp;
        std::ofstream o(tpath, std::ios::binary | std::ios::trunc);
        if (o && o.write(code->data(), (std::streamsize)code->size()) && (o.close(), !o.fail())) {
            if (std::rename(tpath.c_str(), path.c_str()) != 0) (void)std::remove(tpath.c_str());
        } else {
            (void)std::remove(tpath.c_str());
        }
    }
    return KMC_OK;
}

namespace kmc_host {
// the samplers' hint for the compiles of the calling thread: a big ensemble is worth the offline compiler's extra second (see offline_compiler_wanted)
void set_offline_compiler_hint(bool wanted) { g_offline_wanted = wanted ? 1 : 0; }

// ---- a function body that is a sum over elements: the text matcher lives in kmc_recognise.hpp (host-only: it is also built into a sanitizer fuzz harness) ----
bool recognise_separable(kmc_user_density* ud)
{
    if (!ud->is_body || ud->nblob > 0 || debug_opt("no-body-routing")) return false;
    SumForm f;
    if (!recognise_sum_form(ud->body, &f)) return false;
    ud->sep_nacc = f.nacc;
    ud->sep = true;
    ud->sep_pair = f.pair;
    ud->sep_functor = f.functor;
    return true;
}

// a function body in the vector kernels, evaluated per walker on the whole proposal (KMC_DEBUG=no-body-vec: the staged kernel instead)
bool body_vec_possible(const kmc_user_density* ud, int64_t ndim)
{
    return ud->is_body && ndim >= 1 && ndim <= kBodyVecMaxDim && !debug_opt("no-body-vec");
}

bool staged_possible(const kmc_user_density* ud, bool f32, int64_t ndim, bool p2p = false)
{
    const char* env = std::getenv("KMC_PLAN");
    // (a blob of more than a few doubles per lane would push the staged kernel's rows out of the registers)
    return ud->is_body && ud->nblob <= 8 && !f32 && !p2p && ndim >= 1 && ndim <= kStagedMaxDim && !(env && std::strcmp(env, "generic") == 0);
}

}  // namespace kmc_host
std::string kmc_host::read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

namespace {
// directory of this shared library (the kernel headers are shipped next to it in csrc/)
std::string library_dir()
{
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&kmc_version), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

}  // namespace

std::string kmc_host::user_density_alias(const kmc_user_density* ud, int64_t ndim)
{
    if (!ud->is_body) return "using UD = kmc::TermPairDensity<UserF>;\n";
    if (ud->nblob > 0) return "using UD = kmc::BodyBlobDensity<UserB, " + std::to_string(ndim > 0 ? ndim : 1) + ", " + std::to_string(ud->nblob) + ">;\n";
    return "using UD = kmc::BodyDensity<UserB, " + std::to_string(ndim > 0 ? ndim : 1) + ">;\n";
}
std::string kmc_host::user_functor_source(const kmc_user_density* ud)
{
    std::ostringstream src;
    if (ud->is_body) {
        src << "namespace {\nstruct UserB {\n"
            << "  __device__ static double eval(const double* x, int n, const double* p" << (ud->nblob > 0 ? ", double* blob" : "") << ") { (void)x; (void)n; (void)p;\n"
            << ud->body << "\n  }\n};\n}\n";
        return src.str();
    }
    src << "namespace {\nstruct UserF {\n"
        << "  static constexpr bool kHasPair = " << (ud->has_pair ? "true" : "false") << ";\n"
        << "  __device__ static double term(double x, int d, int n, const double* p) { (void)d; (void)n; (void)p; return (" << ud->term << "); }\n"
        << "  __device__ static double pair(double x, double y, int d, int n, const double* p) { (void)x; (void)y; (void)d; (void)n; (void)p; return ("
        << (ud->has_pair ? ud->pair : std::string("0.0")) << "); }\n};\n}\n";
    return src.str();
}
std::string kmc_host::user_header_dir()
{
    const char* envdir = std::getenv("KMC_CSRC_DIR");
    return envdir ? std::string(envdir) : library_dir() + "/csrc";
}

namespace kmc_host {
kmc_status compile_user(kmc_user_density* ud, bool with_vec, int L, int K, int iter, bool ragged,
                        int resident_K, bool resident_ragged, int island_S, bool f32, const std::vector<char>** out, int64_t ndim = 0,
                        bool p2p = false, int generation_nd = 0)
{
    // resident_K also sizes the island kernel (same row striping: 2 lanes per walker, K chunks)
    // a body in the vector kernels: lane-striped when it is a recognised sum over elements (sep), else rows lane-striped and the body
    // evaluated per walker on the whole proposal (kmc_kernels.hpp, RowEvalTrait; no blobs, ndim <= kBodyVecMaxDim)
    if (ud->is_body && ((with_vec && !sep_routed(ud) && !body_vec_possible(ud, ndim)) || island_S > 0))
        return fail(KMC_ERR_UNSUPPORTED, "this body density runs in the one-walker-per-lane kernels only");
    const bool staged = !with_vec && staged_possible(ud, f32, ndim, p2p);
    char key[128];
    std::snprintf(key, sizeof(key), "%d:%d,%d,%d,%d|%d,%d|%d|%d|%lld|%d|%d|%d|%d|%d", (int)with_vec, L, K, iter, (int)ragged, resident_K,
                  (int)resident_ragged, island_S, (int)f32, ud->is_body ? (long long)ndim : 0ll, (int)staged, (int)p2p, ud->nblob, (int)sep_routed(ud) + 2 * (int)offline_compiler_wanted(),
                  generation_nd);
    const char* peer = p2p ? "true" : "false";         // KMC_P2P: partner rows read from their owners (pull)
    const char* rowt = f32 ? "float" : "double";       // storage type of the walker rows (KMC_F32 / KMC_F64)
    std::lock_guard<std::mutex> lock(ud->mu);
    auto it = ud->code.find(key);
    if (it != ud->code.end()) { *out = &it->second; return KMC_OK; }

    const char* envdir = std::getenv("KMC_CSRC_DIR");
    const std::string dir = envdir ? std::string(envdir) : library_dir() + "/csrc";
    const std::string h_dev = read_file(dir + "/kmc_device.hpp"), h_ker = read_file(dir + "/kmc_kernels.hpp"),
                      h_isl = read_file(dir + "/kmc_islands.hpp"), h_gen = read_file(dir + "/kmc_generation.hpp");
    if (h_dev.empty() || h_ker.empty() || h_isl.empty() || h_gen.empty())
        return fail(KMC_ERR_BAD_ARG, "user density: kernel headers not found in " + dir + " (set KMC_CSRC_DIR)");

    std::ostringstream src;
    src << "#include \"kmc_islands.hpp\"\n#include \"kmc_generation.hpp\"\n" << user_functor_source(ud) << user_density_alias(ud, ndim)
        << (ud->is_body && with_vec && sep_routed(ud) ? ud->sep_functor + (ud->sep_nacc > 1 ? "using UDV = kmc::SepDensityN<UserS>;\n" : "using UDV = kmc::SepDensity<UserS>;\n")
                                                : std::string("using UDV = UD;\n"))
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_generic(KMC_FRONT_PARAMS, const kmc::HalfStepArgs a) { kmc::half_step_generic_body<UD, " << peer << ", " << rowt << ">(KMC_FRONT_PACK, a); }\n"
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_logpdf(const kmc::LogpdfArgs a) { kmc::logpdf_rows_body<UD>(a); }\n"
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_init_ball(const kmc::InitBallArgs a) { kmc::init_ball_body<UD>(a); }\n";
    if (ud->is_body && with_vec && sep_routed(ud))
        src << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_logpdf_sep(const kmc::LogpdfArgs a) { kmc::logpdf_rows_body<UDV>(a); }\n";
    if (staged)
        src << "extern \"C\" __global__ __launch_bounds__(" << kStagedTPB << ") void kmc_user_staged(KMC_FRONT_PARAMS, const kmc::HalfStepArgs a) { kmc::half_step_staged_body<UD, "
            << ndim << ">(KMC_FRONT_PACK, a); }\n";
    if (with_vec)
        src << "extern \"C\" __global__ __launch_bounds__(" << vec_tpb(L) << ") void kmc_user_vec(KMC_FRONT_PARAMS, const kmc::HalfStepArgs a) { kmc::half_step_vec_body<UDV, "
            << L << ", " << K << ", " << iter << ", " << peer << ", " << (ragged ? "true" : "false") << ", " << rowt << ">(KMC_FRONT_PACK, a); }\n";
    if (resident_K == 2048 && island_S == 0 && ud->is_body)   // two walkers per thread (1026 .. 2048 walkers)
        src << "extern \"C\" __global__ __launch_bounds__(1024) void kmc_user_resident(const kmc::ResidentArgs a) { kmc::resident_lane2_body<UD, " << ndim << ">(a); }\n";
    else if (resident_K <= -100 && island_S == 0 && !ud->is_body)
        src << "extern \"C\" __global__ __launch_bounds__(1024) void kmc_user_resident(const kmc::ResidentArgs a) { kmc::resident_lane2_body<UD, " << -resident_K - 100 << ">(a); }\n";
    else if (resident_K > 0 && island_S == 0 && ud->is_body)       // one walker per thread; resident_K = the workgroup size it is launched with at most
        src << "extern \"C\" __global__ __launch_bounds__(" << resident_K << ") void kmc_user_resident(const kmc::ResidentArgs a) { kmc::resident_lane_body<UD, " << ndim << ", true, " << rowt << ">(a); }\n";
    if (resident_K > 0 && island_S == 0 && !ud->is_body)
        src << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_resident(const kmc::ResidentArgs a) { kmc::resident_body<UD, "
            << resident_K << ", " << (resident_ragged ? "true" : "false") << ">(a); }\n";
    if (resident_K < 0 && resident_K > -100 && island_S == 0 && !ud->is_body)      // short rows: one walker per thread, ndim <= ND = -resident_K
        src << "extern \"C\" __global__ __launch_bounds__(1024) void kmc_user_resident(const kmc::ResidentArgs a) { kmc::resident_lane_body<UD, " << -resident_K << ", true, " << rowt << ">(a); }\n";
    if (resident_K > 0 && island_S > 0)
        src << "extern \"C\" __global__ __launch_bounds__(" << island_S << ") void kmc_user_island(const kmc::IslandArgs a) { kmc::island_epoch_body<UD, "
            << island_S << ", " << resident_K << ", " << (resident_ragged ? "true" : "false") << ">(a); }\n";
    if (generation_nd > 0)          // one launch per generation, one walker per lane (kmc_generation.hpp): the density element by element / the body as written
        src << "extern \"C\" __global__ __launch_bounds__(" << kGenerationTPB << ") void kmc_user_generation(KMC_GEN_FRONT_PARAMS, const kmc::GenerationArgs a) { kmc::generation_lane_body<UD, "
            << generation_nd << ">(KMC_GEN_FRONT_PACK, a); }\n";
    if (generation_nd < 0)          // ... rows lane-striped (L = -generation_nd / 100, K = -generation_nd % 100): term / pair densities, bodies recognised as sums
        src << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_generation(KMC_GEN_FRONT_PARAMS, const kmc::GenerationArgs a) { kmc::generation_group_body<UDV, "
            << (-generation_nd) / 100 << ", " << (-generation_nd) % 100 << ">(KMC_GEN_FRONT_PACK, a); }\n";
    const std::string text = src.str();

    const char* headers[4] = {h_ker.c_str(), h_dev.c_str(), h_isl.c_str(), h_gen.c_str()};
    const char* names[4] = {"kmc_kernels.hpp", "kmc_device.hpp", "kmc_islands.hpp", "kmc_generation.hpp"};
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=14"};
    std::vector<char> code;
    std::string log;
    const kmc_status cst = rtc_compile_cached(text, "kmc_user_density.hip", 4, headers, names, 6, opts, &code, &log);
    if (cst == KMC_ERR_BAD_ARG) return fail(KMC_ERR_BAD_ARG, "user density does not compile:\n" + log);
    if (cst != KMC_OK) return cst;
    auto ins = ud->code.emplace(key, std::move(code));
    *out = &ins.first->second;
    return KMC_OK;
}

kmc_status load_user(kmc_user_density* ud, bool with_vec, int L, int K, int iter, bool ragged, UserKernels* uk,
                     int resident_K, bool resident_ragged, int island_S, bool f32, int64_t ndim, bool p2p, int generation_nd)
{
    const std::vector<char>* code = nullptr;
    KMC_TRY(compile_user(ud, with_vec, L, K, iter, ragged, resident_K, resident_ragged, island_S, f32, &code, ndim, p2p, generation_nd));
    {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(ud->mu);
        auto& slot = ud->modules[{static_cast<const void*>(code), dev}];
        if (!slot) {
            hipModule_t m = nullptr;
            HIP_TRY(hipModuleLoadData(&m, code->data()));
            slot = std::shared_ptr<void>(static_cast<void*>(m), [](void* p) { if (p) (void)hipModuleUnload(static_cast<hipModule_t>(p)); });
        }
        uk->keep = slot;
        uk->mod = static_cast<hipModule_t>(slot.get());
    }
    HIP_TRY(hipModuleGetFunction(&uk->generic, uk->mod, "kmc_user_generic"));
    HIP_TRY(hipModuleGetFunction(&uk->logpdf, uk->mod, "kmc_user_logpdf"));
    HIP_TRY(hipModuleGetFunction(&uk->init_ball, uk->mod, "kmc_user_init_ball"));
    if (with_vec) HIP_TRY(hipModuleGetFunction(&uk->vec, uk->mod, "kmc_user_vec"));
    uk->logpdf_sep = nullptr;
    if (with_vec && ud->is_body && sep_routed(ud)) HIP_TRY(hipModuleGetFunction(&uk->logpdf_sep, uk->mod, "kmc_user_logpdf_sep"));
    if (!with_vec && staged_possible(ud, f32, ndim, p2p)) HIP_TRY(hipModuleGetFunction(&uk->staged, uk->mod, "kmc_user_staged"));
    if (resident_K != 0 && island_S == 0) HIP_TRY(hipModuleGetFunction(&uk->resident, uk->mod, "kmc_user_resident"));
    if (resident_K > 0 && island_S > 0) HIP_TRY(hipModuleGetFunction(&uk->island, uk->mod, "kmc_user_island"));
    uk->generation = nullptr;
    if (generation_nd != 0) HIP_TRY(hipModuleGetFunction(&uk->generation, uk->mod, "kmc_user_generation"));
    return KMC_OK;
}

}  // namespace kmc_host
KMC_EXPORT kmc_status kmc_user_density_create(const char* term_expr, const char* pair_expr, kmc_user_density** out)
{
    if (!term_expr || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    kmc_user_density* ud = new kmc_user_density();
    ud->term = term_expr;
    ud->has_pair = pair_expr != nullptr && pair_expr[0] != '\0';
    if (ud->has_pair) ud->pair = pair_expr;
    const std::vector<char>* code = nullptr;
    const kmc_status st = compile_user(ud, false, 0, 0, 0, false, 0, false, 0, false, &code);   // syntax check now, not at first use
    if (st != KMC_OK) { delete ud; return st; }
    *out = ud;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_user_density_create_body(const char* body, kmc_user_density** out)
{
    if (!body || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    kmc_user_density* ud = new kmc_user_density();
    ud->body = body;
    ud->is_body = true;
    const std::vector<char>* code = nullptr;
    const kmc_status st = compile_user(ud, false, 0, 0, 0, false, 0, false, 0, false, &code, 4);   // syntax check now (any ndim)
    if (st != KMC_OK) { delete ud; return st; }
    if (recognise_separable(ud)) {
        // the generated functor must compile too (a body the recogniser misread: then it is simply not routed)
        const std::vector<char>* vcode = nullptr;
        if (compile_user(ud, true, 4, 1, 1, false, 0, false, 0, false, &vcode, 8) != KMC_OK) { ud->sep = false; ud->sep_functor.clear(); }
    }
    *out = ud;
    return KMC_OK;
}

// 1 when the samplers run this body density in the lane-striped vector kernels (a recognised sum over elements), else 0
KMC_EXPORT int kmc_user_density_is_separable(const kmc_user_density* ud) { return ud && ud->sep ? 1 : 0; }

// ... whose body also fills blob[0 .. nblob): double logpdf(const double* x, int n, const double* p, double* blob)
KMC_EXPORT kmc_status kmc_user_density_create_body_blob(const char* body, int nblob, kmc_user_density** out)
{
    if (!body || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (nblob < 1 || nblob > 1024) return fail(KMC_ERR_BAD_ARG, "nblob must be in 1 .. 1024 (doubles per evaluation)");
    *out = nullptr;
    kmc_user_density* ud = new kmc_user_density();
    ud->body = body;
    ud->is_body = true;
    ud->nblob = nblob;
    const std::vector<char>* code = nullptr;
    const kmc_status st = compile_user(ud, false, 0, 0, 0, false, 0, false, 0, false, &code, 4);   // syntax check now (any ndim)
    if (st != KMC_OK) { delete ud; return st; }
    *out = ud;
    return KMC_OK;
}

KMC_EXPORT int kmc_user_density_nblob(const kmc_user_density* ud) { return ud ? ud->nblob : -1; }

KMC_EXPORT void kmc_user_density_destroy(kmc_user_density* ud) { delete ud; }

