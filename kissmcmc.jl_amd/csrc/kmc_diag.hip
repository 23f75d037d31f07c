// kmc_diag.hip -- diagnostics: the accept-term probe of the C ABI (include/kissmcmc_hip.h: "diagnostics"), the native
// backtrace of an abort() (KMC_DEBUG=abort-backtrace), the guard bands behind a sampler's device allocations (KMC_DEBUG=poison), and the
// device's free memory.
//
// kmc_debug_accept_terms: the random side of the accept test of reference src/samplers.jl:260,
//     (N-1) * log(z) + p1 - p0 >= log(rand()),
// exactly as the half-step kernels compute it (kmc_device.hpp: draw_step -- Philox block, z = (u c1 + c0)^2 and the two
// logarithms from the kernels' own log_pos_normal), for a run of walkers of one step.  The kernels and the CPU oracle take
// these two logarithms from different implementations (each < 1 ulp); tests/test_gpu_accept_margin.py measures the gap and
// what it means for "identical accept decisions".
#include <thread>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "kmc_sampler.hpp"

using namespace kmc;
using namespace kmc_host;

// Diagnostics (KMC_DEBUG=abort-backtrace[=/path/to/file] in the environment when the library is loaded): the native call stack of an abort()
// raised anywhere in the process (the HIP runtime aborts on internal errors without a message), on stderr.
namespace {
#ifndef KMC_DIAG_ALWAYS
#define KMC_DIAG_ALWAYS 0          // -DKMC_DIAG_ALWAYS=1: a diagnostics build that always installs the handler (file /tmp/kmc_abort_bt.txt)
#endif
char g_abort_path[512] = "";
bool abort_backtrace_wanted()
{
    std::string v;
    if (!debug_opt("abort-backtrace", &v)) return KMC_DIAG_ALWAYS != 0;
    if (!v.empty() && v[0] == '/') std::snprintf(g_abort_path, sizeof(g_abort_path), "%s", v.c_str());
    return true;
}
void abort_backtrace(int sig)
{
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\n[kissmcmc_hip] SIGABRT, native stack:\n";
    int fd = 2;                                          // abort-backtrace=/path/to/file: there (a test runner may have captured fd 2)
    const char* where = g_abort_path[0] ? g_abort_path : nullptr;       // (resolved when the handler was installed: nothing here may allocate)
    if (!where && KMC_DIAG_ALWAYS) where = "/tmp/kmc_abort_bt.txt";
    if (where && where[0] == '/') { const int f = open(where, O_WRONLY | O_CREAT | O_APPEND, 0644); if (f >= 0) fd = f; }
    (void)!write(fd, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, fd);
    // what the runtime printed before it aborted: a test runner that captures fd 2 keeps it in a temporary file
    struct stat st;
    if (fd != 2 && fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        static char buf[8192];
        const off_t from = st.st_size > (off_t)sizeof(buf) ? st.st_size - (off_t)sizeof(buf) : 0;
        const ssize_t got = pread(2, buf, sizeof(buf), from);
        const char hdr[] = "[kissmcmc_hip] tail of the captured stderr:\n";
        (void)!write(fd, hdr, sizeof(hdr) - 1);
        if (got > 0) (void)!write(fd, buf, (size_t)got);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
struct AbortBacktraceInstaller {
    AbortBacktraceInstaller() { if (abort_backtrace_wanted()) signal(SIGABRT, abort_backtrace); }
} g_abort_backtrace_installer;
}  // namespace

void kmc_host::reinstall_abort_backtrace()
{
    if (abort_backtrace_wanted()) signal(SIGABRT, abort_backtrace);      // (somebody may have replaced it)
}

namespace kmc_host {
void check_guards(kmc_sampler* s)
{
    std::vector<unsigned char> h(kGuardBytes);
    for (const auto& g : s->guards) {
        if (copy_sync(h.data(), g.first, kGuardBytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (size_t i = 0; i < kGuardBytes; ++i)
            if (h[i] != 0xA5) {
                std::fprintf(stderr, "[kissmcmc_hip] KMC_DEBUG=poison: byte %zu behind a device allocation of %zu bytes was overwritten (%s)\n", i, g.second,
                             s->plan.vec ? "vec kernels" : "generic / staged kernels");
                std::abort();
            }
    }
    s->guards.clear();
}
// ---- the allocation cache (kmc_host.hpp) ----------------------------------------------------------------------------
namespace {
constexpr size_t kCacheMaxBlock = (size_t)8 << 20, kCacheCap = (size_t)128 << 20;
struct DevCache {
    std::map<size_t, std::vector<void*>> free_blocks;
    size_t held = 0;
};
std::mutex g_cache_mu;
std::map<int, DevCache> g_cache;                       // by device ordinal
std::map<void*, std::pair<int, size_t>> g_cache_live;  // blocks handed out: device, rounded size
bool cache_enabled()
{
    static const bool on = !debug_opt("poison");
    return on;
}
}  // namespace

hipError_t cache_alloc(void** p, size_t bytes)
{
    const size_t r = (bytes + 255) & ~(size_t)255;
    if (!cache_enabled() || bytes == 0 || r > kCacheMaxBlock) {
        const hipError_t e0 = hipMalloc(p, bytes);
        if (e0 == hipSuccess && cache_enabled()) {      // (an address the cache once handed out and somebody gave to hipFree: forget it)
            std::lock_guard<std::mutex> lock(g_cache_mu);
            g_cache_live.erase(*p);
        }
        return e0;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        DevCache& c = g_cache[dev];
        auto it = c.free_blocks.find(r);
        if (it != c.free_blocks.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            c.held -= r;
            g_cache_live[*p] = {dev, r};
            return hipSuccess;
        }
    }
    e = hipMalloc(p, r);
    if (e != hipSuccess) {                             // out of memory: give the cache back and try once more
        (void)hipGetLastError();
        kmc_device_cache_release();
        e = hipMalloc(p, r);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lock(g_cache_mu);
    g_cache_live[*p] = {dev, r};
    return hipSuccess;
}

void cache_free(void* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        auto it = g_cache_live.find(p);
        if (it != g_cache_live.end()) {
            const int dev = it->second.first;
            const size_t r = it->second.second;
            g_cache_live.erase(it);
            DevCache& c = g_cache[dev];
            if (c.held + r <= kCacheCap) {
                c.free_blocks[r].push_back(p);
                c.held += r;
                return;
            }
        }
    }
    (void)hipFree(p);
}

size_t cache_held_bytes(int device)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    auto it = g_cache.find(device);
    return it == g_cache.end() ? 0 : it->second.held;
}

}  // namespace kmc_host

// every block the allocation cache holds goes back to the device (all devices); the cache fills again as samplers come and go
KMC_EXPORT void kmc_device_cache_release(void)
{
    std::map<int, kmc_host::DevCache> taken;
    {
        std::lock_guard<std::mutex> lock(kmc_host::g_cache_mu);
        taken.swap(kmc_host::g_cache);
    }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto& d : taken) {
        (void)hipSetDevice(d.first);
        for (auto& b : d.second.free_blocks) for (void* p : b.second) (void)hipFree(p);
    }
    if (have) (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

// Fault a host buffer in (include/kissmcmc_hip.h): a fresh allocation's pages are created one fault at a time by whoever writes
// first -- after a run that is the chain read-out's copy threads, at 2-3 GB/s each; done here, by a helper thread of the caller's while
// the device samples, it is off the call's critical path.
KMC_EXPORT void kmc_host_prefault(void* buffer, uint64_t nbytes, int nthreads)
{
    if (!buffer || nbytes == 0) return;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
    constexpr uint64_t kPage = 4096;
    const uint64_t npages = (nbytes + kPage - 1) / kPage;
    if ((uint64_t)nthreads > npages) nthreads = (int)npages;
    auto work = [=](int t) {
        volatile unsigned char* base = static_cast<volatile unsigned char*>(buffer);
        const uint64_t p0 = npages * (uint64_t)t / (uint64_t)nthreads, p1 = npages * (uint64_t)(t + 1) / (uint64_t)nthreads;
        for (uint64_t pg = p0; pg < p1; ++pg) { const uint64_t off = pg * kPage; base[off] = base[off]; }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}

// free / total bytes of a device's memory (hipMemGetInfo), for callers that decide between a device chain and a streamed one
KMC_EXPORT kmc_status kmc_device_free_bytes(int device, uint64_t* free_bytes, uint64_t* total_bytes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(KMC_ERR_NO_DEVICE, "no HIP device visible"); }
    if (device < 0 || device >= ndev) return fail(KMC_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    f += kmc_host::cache_held_bytes(device);            // (blocks the allocation cache holds are the caller's to have)
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return KMC_OK;
}


namespace {

__global__ __launch_bounds__(256) void accept_terms_kernel(DrawConsts dc, uint64_t step, uint64_t walker0, int64_t n,
                                                           uint32_t* partner, double* z, double* t1, double* lu)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Draw d = draw_step(dc, step, walker0 + (uint64_t)i);
    if (partner) partner[i] = d.partner;
    z[i] = d.z;
    t1[i] = d.t1;
    lu[i] = d.lu;
}

}  // namespace

KMC_EXPORT kmc_status kmc_debug_accept_terms(uint64_t seed, uint64_t step, uint64_t walker0, int64_t n, int64_t nhalf, double a_scale,
                                             int64_t ndim, int device, int64_t* partner_host, double* z_host, double* t1_host, double* lu_host)
{
    if (n < 0 || nhalf <= 0 || nhalf >= (int64_t)1 << 31 || !(a_scale > 1.0) || ndim < 1 || !z_host || !t1_host || !lu_host)
        return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (n == 0) return KMC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(KMC_ERR_NO_DEVICE, "no HIP device visible"); }
    if (device < 0 || device >= ndev) return fail(KMC_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    ScopedStream ss;
    HIP_TRY(ss.create());
    DrawConsts dc{};
    dc.seed_lo = (uint32_t)seed; dc.seed_hi = (uint32_t)(seed >> 32);
    dc.nhalf = (uint32_t)nhalf;
    dc.c0 = std::sqrt(1.0 / a_scale);                                  // as make_args (src/samplers.jl:227, hoisted)
    dc.c1 = std::sqrt(a_scale) - std::sqrt(1.0 / a_scale);
    dc.nm1 = (double)(ndim - 1);
    const int64_t piece = (int64_t)1 << 22;                             // 4 Mi draws per launch: 112 MiB of device scratch
    const int64_t m = n < piece ? n : piece;
    char* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, (size_t)m * (3 * sizeof(double) + sizeof(uint32_t))));
    double* dz = reinterpret_cast<double*>(buf);
    double* dt1 = dz + m;
    double* dlu = dt1 + m;
    uint32_t* dpart = reinterpret_cast<uint32_t*>(dlu + m);
    std::vector<uint32_t> hp(partner_host ? (size_t)m : 0);
    hipError_t e = hipSuccess;
    for (int64_t i0 = 0; i0 < n && e == hipSuccess; i0 += m) {
        const int64_t k = n - i0 < m ? n - i0 : m;
        hipLaunchKernelGGL(accept_terms_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, ss.st, dc, step, walker0 + (uint64_t)i0, k,
                           partner_host ? dpart : nullptr, dz, dt1, dlu);
        e = hipGetLastError();
        if (e == hipSuccess) e = copy_sync(z_host + i0, dz, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ss.st);
        if (e == hipSuccess) e = copy_sync(t1_host + i0, dt1, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ss.st);
        if (e == hipSuccess) e = copy_sync(lu_host + i0, dlu, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ss.st);
        if (e == hipSuccess && partner_host) {
            e = copy_sync(hp.data(), dpart, (size_t)k * sizeof(uint32_t), hipMemcpyDeviceToHost, ss.st);
            for (int64_t j = 0; j < k; ++j) partner_host[i0 + j] = (int64_t)hp[(size_t)j];
        }
    }
    (void)hipFree(buf);
    HIP_TRY(e);
    return KMC_OK;
}
