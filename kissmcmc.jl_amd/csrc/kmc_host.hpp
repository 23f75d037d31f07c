// kmc_host.hpp -- what the host-side translation units of the library share (kmc_sampler.hip and its siblings: samplers;
// kmc_metropolis_api.hip: many-chain Metropolis; kmc_acorr.hip: autocorrelation diagnostics).  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kissmcmc_hip.h"
#include "kmc_tables.hpp"

#define KMC_EXPORT extern "C" __attribute__((visibility("default")))

namespace kmc_host {

// message of the last failure on the calling thread (kmc_last_error)
extern thread_local std::string g_err;
inline kmc_status fail(kmc_status st, const std::string& msg)
{
    g_err = msg;
    return st;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return ::kmc_host::fail(e_ == hipErrorOutOfMemory ? KMC_ERR_OOM : KMC_ERR_HIP,     \
                                    std::string(#expr) + ": " + hipGetErrorString(e_));        \
        }                                                                                      \
    } while (0)

#define KMC_TRY(expr)                                                                          \
    do {                                                                                       \
        kmc_status s_ = (expr);                                                                \
        if (s_ != KMC_OK) return s_;                                                           \
    } while (0)

// KMC_DEBUG="opt[=value],opt,...": the test-only, A/B and diagnostic switches of the library behind ONE environment variable (the
// documented ones are in README.md; none is needed in normal use).  Read anew at every call (tests set it between samplers).
// Returns whether `name` is listed; *value gets what follows its '=' ("" when nothing does).
inline bool debug_opt(const char* name, std::string* value = nullptr)
{
    const char* e = std::getenv("KMC_DEBUG");
    if (!e) return false;
    const size_t n = std::strlen(name);
    for (const char* p = e; *p;) {
        const char* q = std::strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : std::strlen(p);
        if (len >= n && std::strncmp(p, name, n) == 0 && (len == n || p[n] == '=')) {
            if (value) *value = len > n ? std::string(p + n + 1, len - n - 1) : std::string();
            return true;
        }
        if (!q) break;
        p = q + 1;
    }
    return false;
}
inline long debug_opt_long(const char* name, long absent)
{
    std::string v;
    return debug_opt(name, &v) && !v.empty() ? std::atol(v.c_str()) : absent;
}

// kernel tables of the menu densities (kmc_inst_*.hip) and the digest of kmc_config.params the kernels take
bool lookup(int density, int L, int K, int iter, bool p2p, bool ragged, bool f32, kmc::HalfStepFn* vec, kmc::HalfStepFn* gen, kmc::LogpdfFn* lp);
kmc_status digest_params(const kmc_config& c, kmc::DensityParams* dp);

// runtime-compiled user densities (hiprtc)
// Blocking copies / fills WITHOUT the legacy (null) stream: hipMemcpy, hipMemset and hipDeviceSynchronize go through it, and
// a legacy-stream operation issued while ANOTHER host thread captures a hipGraph fails ("would make the legacy stream depend
// on a capturing stream") and invalidates that capture.  Everything here runs on a stream the caller names and waits for it.
//
// And WITHOUT handing pageable host memory to an asynchronous copy: for those the runtime page-locks the caller's pages itself
// and keeps the mapping in a cache -- read-only when the memory was a copy's source.  A later device-to-host copy into a
// heap block that malloc placed on the same pages then died with "Memory access fault by GPU ... Write access to a read-only
// page" (intermittent, layout-dependent; found by running the tests in another order).  copy_sync stages pageable memory
// through page-locked bounce buffers of its own (two, so that the host memcpy of one chunk overlaps the DMA of the next); host
// memory that IS page-locked (hipHostMalloc, hipHostRegister) goes straight through.
hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
// (no 2-D variant on purpose: padded rows are repacked on the host or compacted on the device, then copied contiguously)
inline hipError_t fill_sync(void* dst, int value, size_t bytes, hipStream_t st)
{
    const hipError_t e = hipMemsetAsync(dst, value, bytes, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
// A large allocation that cannot fit is refused BEFORE hipMalloc is asked: after a failed allocation of hundreds of GB the
// runtime aborted the process a few calls later (observed with a 336 GB chain: 4 of 5 runs).
inline kmc_status check_device_room(size_t need, const char* what)
{
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (need > free_b)
        return fail(KMC_ERR_OOM, std::string(what) + " needs " + std::to_string(need >> 20) + " MiB of device memory, " + std::to_string(free_b >> 20) + " MiB are free");
    return KMC_OK;
}
// a private non-blocking stream for the duration of one call
struct ScopedStream {
    hipStream_t st = nullptr;
    hipError_t create() { return hipStreamCreateWithFlags(&st, hipStreamNonBlocking); }
    ~ScopedStream() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
};

// hiprtc compilation of `text` (+ named headers) for gfx950, with a DISK CACHE of the code objects: a script that defines its
// density in source pays the compiler (0.2-0.5 s per density and kernel geometry) once, not in every process.  Key: two
// 64-bit hashes over the program, every header, the options and the hiprtc version; directory $KMC_CACHE_DIR, else
// $XDG_CACHE_HOME/kissmcmc_hip, else $HOME/.cache/kissmcmc_hip; KMC_CACHE_DIR=off switches it off; a cache that cannot be
// read or written is simply not used.  On a compile error *log holds the compiler's messages.
kmc_status rtc_compile_cached(const std::string& text, const char* program_name, int nheaders, const char* const* header_text,
                              const char* const* header_names, int nopts, const char* const* opts, std::vector<char>* code, std::string* log);

kmc_status download_by_walker(const void* src_dev, bool is_float, int64_t nl, int64_t ld, int64_t width, int64_t K, double* dst_host, hipStream_t st);
std::string user_functor_source(const kmc_user_density* ud);      // the functor(s) ...
std::string user_density_alias(const kmc_user_density* ud, int64_t ndim);   // ... and "using UD = ...;" over them
std::string user_header_dir();                      // where the kernel headers live (KMC_CSRC_DIR or <library dir>/csrc)
std::string read_file(const std::string& path);

template <class Args>
hipError_t launch_module(hipFunction_t f, unsigned grid, unsigned tpb, hipStream_t st, const Args& args, unsigned lds_bytes = 0)
{
    Args copy = args;
    size_t size = sizeof(Args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &copy, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(f, grid, 1, 1, tpb, 1, 1, lds_bytes, st, nullptr, extra);
}

// Small device allocations, cached per device (kmc_diag.hip).  A sampler of the reference's own size lives for a millisecond or
// two, and hipMalloc / hipFree of its dozen buffers were more than half of that (create 0.35 ms, destroy 0.85 ms of a 1.7 ms
// README call): blocks of up to 8 MiB go back to a free list instead (at most 128 MiB held per device, exact rounded sizes, so
// samplers of one shape reuse each other's blocks).  A block is handed back only after its owner has synchronised the streams that
// touched it (kmc_sampler_destroy: its own streams; the whole device when a caller's stream was ever bound or the sampler is a shard).  Off with KMC_DEBUG=poison (guard bands); kmc_device_cache_release() returns everything.
hipError_t cache_alloc(void** p, size_t bytes);        // on the current device
void cache_free(void* p);                              // nullptr is fine; pointers the cache did not hand out go to hipFree
size_t cache_held_bytes(int device);

// integrated autocorrelation time of a device-resident chain [nsamples][nwalkers][ndim] (kmc_acorr.hip)
kmc_status int_acorr_check(int64_t nsamples, int64_t nwalkers, int64_t ndim, double c, const double* tau, const double* converged);
kmc_status int_acorr_device(const double* chain_dev, int64_t nsamples, int64_t nwalkers, int64_t ndim, double c, double* tau, double* converged);

// RCCL, loaded on demand (kmc_rccl.hip)
kmc_status rccl_version(int* version, const char** path);
kmc_status rccl_unique_id(void* id_out);
kmc_status rccl_comm_create(const void* id_bytes, int rank, int nranks, void** comm_out);
void rccl_comm_destroy(void* comm);
kmc_status rccl_all_gather_f64(void* comm, const double* send, double* recv, size_t count, hipStream_t stream);

}  // namespace kmc_host

// The opaque handle of kmc_user_density_create: the two expressions and the code objects compiled from them so far.
struct kmc_user_density {
    std::string term, pair;
    bool has_pair = false;
    std::string body;                                // kmc_user_density_create_body: the whole function body instead of term / pair
    bool is_body = false;
    int nblob = 0;                                   // kmc_user_density_create_body_blob: doubles the body writes to blob[] per evaluation
    // A body of the form  `double s = 0; for (int i = 0; i < n; ++i) s += f(x[i]);  return g(s);`  (optionally with the neighbour
    // x[i + 1] and the bound i + 1 < n) is a sum over elements: kmc_user_density_create_body recognises it (kmc_rtc.hip:
    // recognise_separable) and the samplers then run it in the lane-striped vector kernels like a term / pair density, with g as
    // the finish.  Everything else about the density (initial log-pdfs, resident kernels, Metropolis) keeps evaluating the body.
    bool sep = false;
    bool sep_pair = false;                           // the loop reads x[i + 1] / runs to n - 1: its body is the PAIR function
    int sep_nacc = 1;                                // sums the loop feeds (1: SepDensity; 2..4: SepDensityN)
    std::string sep_functor;                         // "struct UserS { term, pair, finish };" generated from the body
    // The recogniser reads TEXT; before a sampler runs the generated form it is evaluated next to the body itself on test points
    // (kmc_sampler.hip: check_sum_form) -- once per (ndim, parameter values): what agrees at one row length and one parameter set says
    // nothing about another.  sep_case: 1 they agree there; 3 no test point had a finite value there (nothing was shown: THAT sampler runs the
    // body as written, the density keeps its routing for other cases).  A case where the two DISAGREE refutes the recogniser: `sep` is
    // cleared for good (sep_note says why) and every later sampler evaluates the body per walker, as written.
    std::map<std::pair<int64_t, uint64_t>, int> sep_case;    // (ndim, digest of the parameter values) -> verdict; under `mu`
    std::mutex check_mu;                             // held by the sampler that finds out (plan, load, check): others over the same density wait
    std::string sep_note;                            // why not (describe())
    std::mutex mu;
    std::map<std::string, std::vector<char>> code;   // geometry key -> gfx950 code object
    // ... and the modules loaded from them, per device: shared by every sampler over this density (hipModuleLoadData is ~0.5 ms,
    // half of what a sampler of the reference's sizes lives); unloaded when the last holder -- this object or a sampler -- lets go
    std::map<std::pair<const void*, int>, std::shared_ptr<void>> modules;
};
// Is the body run in its per-element form BY THE SAMPLER BEING BUILT on this thread?  The density's own flag, unless that sampler's check was blind
// (kmc_sampler_create holds a SepOff for the rest of its set-up; afterwards the sampler's own `sep_off` says it).
inline thread_local bool g_sep_off = false;
struct SepOff { bool prev; explicit SepOff(bool off) : prev(g_sep_off) { if (off) g_sep_off = true; } ~SepOff() { g_sep_off = prev; } };
inline bool sep_routed(const kmc_user_density* ud) { return ud && ud->sep && !g_sep_off; }
