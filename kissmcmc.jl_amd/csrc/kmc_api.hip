// kmc_api.hip -- host side of the C ABI declared in include/kissmcmc_hip.h.
//
// Implements the reference's `emcee` front-end bookkeeping (src/samplers.jl:188-216) and the
// `_emcee` generation loop (src/samplers.jl:232-293) as a stream of half-step kernel launches:
// two dependent launches per generation, replayed from a hipGraph in chunks of
// kGraphChunk generations (the kernel boundary is the join of src/samplers.jl:273).
#include <dlfcn.h>
#include <pthread.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <deque>
#include <condition_variable>
#include <cmath>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define KMC_DEFINE_DRIVER_KERNELS
#include "kmc_host.hpp"

using namespace kmc;
using namespace kmc_host;

KMC_EXPORT int kmc_version(void);

thread_local std::string kmc_host::g_err;

namespace {

constexpr int64_t kGraphChunk = 64;   // generations per hipGraph replay (128 kernel nodes + 1)
constexpr int kUExec = 6;             // executables of the "updated graph" launch mode (kmc_sampler::uexec)


struct Plan {
    HalfStepFn fn = nullptr;
    bool vec = false;
    bool ragged = false;
    int L = 1, K = 1, ITER = 1;
};

// ---- kernel table ------------------------------------------------------------------------

template <int L, int K, int ITER, class T>
FlushFn flush_one()
{
    if constexpr (ITER <= L && ITER * K <= 16) return flush_moments_vec<L, K, ITER, T>;
    else return nullptr;
}

template <int L, int K>
FlushFn flush_iter(int iter, bool f32)
{
    switch (iter) {
    case 1: return f32 ? flush_one<L, K, 1, float>() : flush_one<L, K, 1, double>();
    case 2: return f32 ? flush_one<L, K, 2, float>() : flush_one<L, K, 2, double>();
    case 4: return f32 ? flush_one<L, K, 4, float>() : flush_one<L, K, 4, double>();
    case 8: return f32 ? nullptr : flush_one<L, K, 8, double>();
    case 16: return f32 ? nullptr : flush_one<L, K, 16, double>();
    default: return nullptr;
    }
}

FlushFn flush_lookup(int L, int K, int iter, bool f32)
{
#define KMC_LK(l, k) if (L == l && K == k) return flush_iter<l, k>(iter, f32);
    KMC_LK(1, 1) KMC_LK(2, 1) KMC_LK(4, 1) KMC_LK(8, 1) KMC_LK(16, 1) KMC_LK(32, 1) KMC_LK(64, 1)
    KMC_LK(4, 2) KMC_LK(8, 2) KMC_LK(16, 2) KMC_LK(32, 2) KMC_LK(64, 2)
    KMC_LK(4, 4) KMC_LK(8, 4) KMC_LK(64, 4)
    KMC_LK(64, 8)
#undef KMC_LK
    return nullptr;
}

}  // namespace
bool kmc_host::lookup(int density, int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: table_gaussian_iso(L, K, iter, p2p, ragged, f32, vec, gen, lp); return true;
    case KMC_EXPONENTIAL: table_exponential(L, K, iter, p2p, ragged, f32, vec, gen, lp); return true;
    case KMC_ROSENBROCK: table_rosenbrock(L, K, iter, p2p, ragged, f32, vec, gen, lp); return true;
    case KMC_LOGNORMAL: table_lognormal(L, K, iter, p2p, ragged, f32, vec, gen, lp); return true;
    case KMC_MVNORMAL2: table_mvnormal2(L, K, iter, p2p, ragged, f32, vec, gen, lp); return true;
    default: return false;
    }
}
namespace {

IslandFn island_fn(int density, int S, int K, bool ragged)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return island_gaussian_iso(S, K, ragged);
    case KMC_EXPONENTIAL: return island_exponential(S, K, ragged);
    case KMC_ROSENBROCK: return island_rosenbrock(S, K, ragged);
    case KMC_LOGNORMAL: return island_lognormal(S, K, ragged);
    case KMC_MVNORMAL2: return island_mvnormal2(S, K, ragged);
    default: return nullptr;
    }
}

ResidentFn resident_fn(int density, int tpb, int K, bool ragged)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return resident_gaussian_iso(tpb, K, ragged);
    case KMC_EXPONENTIAL: return resident_exponential(tpb, K, ragged);
    case KMC_ROSENBROCK: return resident_rosenbrock(tpb, K, ragged);
    case KMC_LOGNORMAL: return resident_lognormal(tpb, K, ragged);
    case KMC_MVNORMAL2: return resident_mvnormal2(tpb, K, ragged);
    default: return nullptr;
    }
}

InitBallFn init_ball_fn(int density)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return init_ball_gaussian_iso();
    case KMC_EXPONENTIAL: return init_ball_exponential();
    case KMC_ROSENBROCK: return init_ball_rosenbrock();
    case KMC_LOGNORMAL: return init_ball_lognormal();
    case KMC_MVNORMAL2: return init_ball_mvnormal2();
    default: return nullptr;
    }
}

// Philox4x32-10 on the host (only for the island deal; Salmon et al., SC'11).
void philox_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The deal of epoch e: slot s holds walker (A*s + C) mod N.  Epoch 0 is the identity; later epochs
// take A (made coprime to N by stepping upwards) and C from Philox(ctr = {e, "ISLA", 0}, key = seed).
void island_perm(uint64_t seed, int64_t epoch, int64_t N, int64_t* A, int64_t* C)
{
    if (epoch == 0 || N <= 2) { *A = 1; *C = 0; return; }
    const uint32_t ctr[4] = {(uint32_t)epoch, (uint32_t)((uint64_t)epoch >> 32), 0x49534c41u, 0u};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    philox_host(ctr, key, w);
    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    int64_t a = (int64_t)((((uint64_t)w[0] << 32) | w[1]) % (uint64_t)N);
    if (a < 1) a = 1;
    while (gcd(a, N) != 1) a = (a % N + 1 >= N) ? 1 : a + 1;
    *A = a;
    *C = (int64_t)((((uint64_t)w[2] << 32) | w[3]) % (uint64_t)N);
}

// Dealt sub-ensembles (kmc_config.deal_count): the Philox key of sub-ensemble r, and the affine shuffle (A, C) its S
// slots go through before the deal of `epoch` (A coprime to S) -- from Philox(ctr = {epoch, "DEAL", r}, key = seed).
constexpr uint64_t kDealSeedStride = 0x9E3779B97F4A7C15ull;
uint64_t deal_seed(uint64_t seed, int32_t rank) { return seed + (uint64_t)(rank + 1) * kDealSeedStride; }
void deal_perm(uint64_t seed, int64_t epoch, int32_t rank, int64_t S, int64_t* A, int64_t* C)
{
    const uint32_t ctr[4] = {(uint32_t)epoch, (uint32_t)((uint64_t)epoch >> 32), 0x4445414cu, (uint32_t)rank};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    philox_host(ctr, key, w);
    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    int64_t a = (int64_t)((((uint64_t)w[0] << 32) | w[1]) % (uint64_t)S);
    if (a < 1) a = 1;
    while (gcd(a, S) != 1) a = (a + 1 >= S) ? 1 : a + 1;
    *A = a;
    *C = (int64_t)((((uint64_t)w[2] << 32) | w[3]) % (uint64_t)S);
}

// Default geometry per ndim; KMC_PLAN="L,K,ITER" (or "generic") overrides for tuning.
Plan make_plan(const kmc_config& c, int64_t n_active)
{
    Plan p;
    HalfStepFn vec = nullptr, gen = nullptr;
    LogpdfFn lp = nullptr;
    int L = 0, K = 0, iter = 1;
    const char* env = std::getenv("KMC_PLAN");
    bool force_generic = false;
    if (env && std::strcmp(env, "generic") == 0) force_generic = true;
    else if (env && std::sscanf(env, "%d,%d,%d", &L, &K, &iter) == 3) { /* forced */ }
    else {
        L = 0;
        // a row = ceil(ndim/2) 16-byte chunks, striped over L lanes x K chunks (2*L*K >= ndim; the
        // ragged tail is masked).  Measured on MI355X (scripts/quick_bench.py): 4 lanes x 2 chunks
        // per 64 B of row is the sweet spot.
        const int64_t chunks = (c.ndim + 1) / 2;
        auto pow2ceil = [](int64_t v) { int p = 1; while (p < v) p <<= 1; return p; };
        if (chunks <= 4) { L = pow2ceil(chunks); K = 1; }
        else if (chunks <= 128) { L = pow2ceil((chunks + 1) / 2); K = 2; }
        else if (chunks <= 256) { L = 64; K = 4; }
        else if (chunks <= 512) { L = 64; K = 8; }
        // walkers per group (ITER): two amortise the per-walker scalar work (Philox, two logs) over
        // the wave -- once that still leaves 1.5 waves per SIMD (measured: 2048 single-walker waves run 3-6 % faster
        // as they are, C3 and 32 768 x 32; 3072 and more are faster paired); more only while the grid keeps >= 4096
        // waves (large ensembles)
        iter = 1;
        if (L > 0) {
            const int64_t waves1 = n_active * L / 64;
            if (2 <= L && 2 * K <= 16 && waves1 >= 3072) iter = 2;
            while (iter >= 2 && iter * 2 <= L && iter * 2 * K <= 16 && waves1 / (iter * 2) >= 4096 && iter < 16) iter *= 2;
        }
    }
    const bool ragged = L > 0 && 2 * L * K != c.ndim;
    const bool f32 = c.dtype == KMC_F32;
    if ((ragged || f32) && iter > 4) iter = 4;
    if ((c.flags & KMC_P2P) && iter > 8) iter = 8;
    p.ragged = ragged;
    if (c.density == KMC_HOST_DENSITY) {
        // bound by the host callback: the one-walker-per-lane kernel, any ndim
        p.fn = half_step_host(); p.vec = false; p.ragged = false; p.L = 1; p.K = 1; p.ITER = 1;
        return p;
    }
    if (c.density == KMC_USER_DENSITY) {
        // kernels are compiled for exactly this geometry when the sampler is created
        const bool body = c.user_density && static_cast<const kmc_user_density*>(c.user_density)->is_body;   // one walker per lane
        if (!body && !force_generic && L > 0 && 2 * L * K >= c.ndim && iter <= L && iter * K <= 16) {
            p.vec = true; p.L = L; p.K = K; p.ITER = iter;
        } else {
            p.vec = false; p.L = 1; p.K = 1; p.ITER = 1;
        }
        return p;
    }
    lookup(c.density, L, K, iter, (c.flags & KMC_P2P) != 0, ragged, f32, &vec, &gen, &lp);
    if (!force_generic && L > 0 && 2 * L * K >= c.ndim && vec != nullptr) {
        p.fn = vec; p.vec = true; p.L = L; p.K = K; p.ITER = iter;
    } else {
        p.fn = gen; p.vec = false; p.L = 1; p.K = 1; p.ITER = 1;
    }
    return p;
}

}  // namespace
kmc_status kmc_host::digest_params(const kmc_config& c, DensityParams* dp)
{
    for (double& v : dp->p) v = 0.0;
    dp->ndim = (int32_t)c.ndim;
    dp->pad_ = 0;
    const double* p = c.params;
    switch (c.density) {
    case KMC_USER_DENSITY:
        if (!c.user_density) return fail(KMC_ERR_BAD_ARG, "KMC_USER_DENSITY needs kmc_config.user_density");
        for (int i = 0; i < 6; ++i) dp->p[i] = p[i];
        return KMC_OK;
    case KMC_HOST_DENSITY:
        return KMC_OK;
    case KMC_GAUSSIAN_ISO:
        if (!(p[1] > 0.0)) return fail(KMC_ERR_BAD_ARG, "gaussian: sigma must be > 0");
        dp->p[0] = p[0]; dp->p[1] = 1.0 / p[1];
        return KMC_OK;
    case KMC_EXPONENTIAL:
        if (!(p[0] > 0.0)) return fail(KMC_ERR_BAD_ARG, "exponential: rate must be > 0");
        dp->p[0] = p[0];
        return KMC_OK;
    case KMC_ROSENBROCK:
        if (!(p[2] > 0.0)) return fail(KMC_ERR_BAD_ARG, "rosenbrock: scale must be > 0");
        dp->p[0] = p[0]; dp->p[1] = p[1]; dp->p[2] = 1.0 / p[2];
        return KMC_OK;
    case KMC_LOGNORMAL:
        if (!(p[1] > 0.0)) return fail(KMC_ERR_BAD_ARG, "lognormal: sigma must be > 0");
        dp->p[0] = p[0]; dp->p[1] = p[1];
        return KMC_OK;
    case KMC_MVNORMAL2:
        for (int i = 0; i < 5; ++i) dp->p[i] = p[i];
        return KMC_OK;
    default:
        return fail(KMC_ERR_BAD_ARG, "unknown density id");
    }
}
namespace {

}  // namespace

// ---- copy_sync (kmc_host.hpp): blocking copies on a named stream, pageable memory staged through page-locked bounce buffers ----
namespace {
constexpr size_t kBounceBytes = (size_t)16 << 20;
struct Bounce {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int device = -1;
};
std::mutex g_bounce_mu;
std::vector<Bounce*> g_bounce_free;          // never released: a handful of 32 MiB pairs for the life of the process

Bounce* bounce_acquire(hipError_t* e)
{
    int dev = 0;
    *e = hipGetDevice(&dev);
    if (*e != hipSuccess) return nullptr;
    {
        std::lock_guard<std::mutex> lock(g_bounce_mu);
        for (size_t i = 0; i < g_bounce_free.size(); ++i)
            if (g_bounce_free[i]->device == dev) { Bounce* b = g_bounce_free[i]; g_bounce_free.erase(g_bounce_free.begin() + (long)i); return b; }
    }
    Bounce* b = new Bounce();
    b->device = dev;
    for (int i = 0; i < 2 && *e == hipSuccess; ++i) {
        *e = hipHostMalloc(&b->buf[i], kBounceBytes, hipHostMallocDefault);
        if (*e == hipSuccess) *e = hipEventCreateWithFlags(&b->ev[i], hipEventDisableTiming);
    }
    if (*e != hipSuccess) {
        for (int i = 0; i < 2; ++i) { if (b->buf[i]) (void)hipHostFree(b->buf[i]); if (b->ev[i]) (void)hipEventDestroy(b->ev[i]); }
        delete b;
        return nullptr;
    }
    return b;
}
void bounce_release(Bounce* b)
{
    std::lock_guard<std::mutex> lock(g_bounce_mu);
    g_bounce_free.push_back(b);
}
// memcpy of a bounce chunk by a few threads: one thread moves ~12 GB/s (and takes the first-touch page faults of a fresh
// destination), which would make a chain download slower than the runtime's own pageable path was
class CopyPool {
public:
    void copy(char* d, const char* s, size_t n)
    {
        constexpr size_t kMin = (size_t)2 << 20;
        if (n < 2 * kMin || forked_child()) { std::memcpy(d, s, n); return; }      // (a forked child has no helper threads)
        start();
        const size_t parts = std::min<size_t>(nworkers_ + 1, n / kMin);
        const size_t each = (n / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lock(m_);
            for (size_t p = 1; p < parts; ++p) {
                const size_t off = p * each;
                if (off >= n) break;
                q_.push_back(Job{d + off, s + off, std::min(each, n - off)});
                ++pending_;
            }
        }
        cv_.notify_all();
        std::memcpy(d, s, std::min(each, n));
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [&] { return pending_ == 0; });
    }
private:
    struct Job { char* d; const char* s; size_t n; };
    static bool& forked_child() { static bool f = false; return f; }
    void start()
    {
        std::lock_guard<std::mutex> lock(m_);
        if (started_) return;
        started_ = true;
        pthread_atfork(nullptr, nullptr, [] { forked_child() = true; });
        unsigned hw = std::thread::hardware_concurrency();
        nworkers_ = hw >= 8 ? 3 : (hw >= 4 ? 1 : 0);
        for (size_t i = 0; i < nworkers_; ++i)
            std::thread([this] {
                for (;;) {
                    Job j;
                    {
                        std::unique_lock<std::mutex> lock(m_);
                        cv_.wait(lock, [&] { return !q_.empty(); });
                        j = q_.front();
                        q_.pop_front();
                    }
                    std::memcpy(j.d, j.s, j.n);
                    {
                        std::lock_guard<std::mutex> lock(m_);
                        if (--pending_ == 0) done_.notify_all();
                    }
                }
            }).detach();                              // parked on the condition variable for the life of the process
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<Job> q_;
    size_t pending_ = 0, nworkers_ = 0;
    bool started_ = false;
};
CopyPool& copy_pool() { static CopyPool* p = new CopyPool(); return *p; }       // (leaked on purpose: its threads outlive static destruction)
std::mutex g_pool_mu;                               // one big copy at a time uses the helpers

void big_memcpy(void* d, const void* s, size_t n)
{
    if (n < ((size_t)4 << 20)) { std::memcpy(d, s, n); return; }
    std::lock_guard<std::mutex> lock(g_pool_mu);
    copy_pool().copy(static_cast<char*>(d), static_cast<const char*>(s), n);
}

bool host_is_page_locked(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // unknown to the runtime: pageable
    return at.type == hipMemoryTypeHost;
}
}  // namespace

hipError_t kmc_host::copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    const bool h2d = kind == hipMemcpyHostToDevice, d2h = kind == hipMemcpyDeviceToHost;
    if ((!h2d && !d2h) || host_is_page_locked(h2d ? src : dst)) {
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
        return e != hipSuccess ? e : hipStreamSynchronize(st);
    }
    hipError_t e = hipSuccess;
    Bounce* b = bounce_acquire(&e);
    if (!b) return e;
    const size_t nchunks = (bytes + kBounceBytes - 1) / kBounceBytes;
    auto len = [&](size_t c) { return c + 1 < nchunks ? kBounceBytes : bytes - c * kBounceBytes; };
    if (h2d) {
        for (size_t c = 0; c < nchunks && e == hipSuccess; ++c) {
            const int i = (int)(c & 1);
            if (c >= 2) e = hipEventSynchronize(b->ev[i]);                       // the DMA that read this buffer two chunks ago
            if (e != hipSuccess) break;
            big_memcpy(b->buf[i], static_cast<const char*>(src) + c * kBounceBytes, len(c));
            e = hipMemcpyAsync(static_cast<char*>(dst) + c * kBounceBytes, b->buf[i], len(c), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipEventRecord(b->ev[i], st);
        }
        const hipError_t es = hipStreamSynchronize(st);
        if (e == hipSuccess) e = es;
    } else {
        e = hipMemcpyAsync(b->buf[0], src, len(0), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(b->ev[0], st);
        for (size_t c = 0; c < nchunks && e == hipSuccess; ++c) {
            const int i = (int)(c & 1);
            if (c + 1 < nchunks) {                                                // the next chunk's DMA runs under this chunk's memcpy
                e = hipMemcpyAsync(b->buf[1 - i], static_cast<const char*>(src) + (c + 1) * kBounceBytes, len(c + 1), hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipEventRecord(b->ev[1 - i], st);
                if (e != hipSuccess) break;
            }
            e = hipEventSynchronize(b->ev[i]);
            if (e == hipSuccess) big_memcpy(static_cast<char*>(dst) + c * kBounceBytes, b->buf[i], len(c));
        }
        const hipError_t es = hipStreamSynchronize(st);
        if (e == hipSuccess) e = es;
    }
    bounce_release(b);
    return e;
}

// ---- hiprtc with a disk cache (kmc_host.hpp) -----------------------------------------------------------------
namespace {
uint64_t fnv1a(uint64_t h, const void* data, size_t n)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
std::string rtc_cache_dir()
{
    if (std::getenv("KMC_NO_DISK_CACHE")) return std::string();
    if (const char* d = std::getenv("KMC_CACHE_DIR")) return std::string(d);
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

// ------------------------------------------------------------------------------------------
// User-supplied densities: two C expressions compiled at run time (hiprtc) into the same kernels.
// ------------------------------------------------------------------------------------------

namespace {

struct UserKernels {
    hipModule_t mod = nullptr;
    hipFunction_t vec = nullptr, generic = nullptr, logpdf = nullptr, resident = nullptr, island = nullptr, init_ball = nullptr;
    hipFunction_t staged = nullptr;     // body densities, double rows, ndim <= kStagedMaxDim: half_step_staged_body
};

bool staged_possible(const kmc_user_density* ud, bool f32, int64_t ndim, bool p2p = false)
{
    const char* env = std::getenv("KMC_PLAN");
    return ud->is_body && !f32 && !p2p && ndim >= 1 && ndim <= kStagedMaxDim && !(env && std::strcmp(env, "generic") == 0);
}

}  // namespace
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

// the user's two expressions as a functor for TermPairDensity
}  // namespace
std::string kmc_host::user_density_alias(const kmc_user_density* ud, int64_t ndim)
{
    if (!ud->is_body) return "using UD = kmc::TermPairDensity<UserF>;\n";
    return "using UD = kmc::BodyDensity<UserB, " + std::to_string(ndim > 0 ? ndim : 1) + ">;\n";
}
std::string kmc_host::user_functor_source(const kmc_user_density* ud)
{
    std::ostringstream src;
    if (ud->is_body) {
        src << "namespace {\nstruct UserB {\n"
            << "  __device__ static double eval(const double* x, int n, const double* p) { (void)x; (void)n; (void)p;\n" << ud->body << "\n  }\n};\n}\n";
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
namespace {

kmc_status compile_user(kmc_user_density* ud, bool with_vec, int L, int K, int iter, bool ragged,
                        int resident_K, bool resident_ragged, int island_S, bool f32, const std::vector<char>** out, int64_t ndim = 0,
                        bool p2p = false)
{
    // resident_K also sizes the island kernel (same row striping: 2 lanes per walker, K chunks)
    if (ud->is_body && (with_vec || resident_K > 0 || island_S > 0))
        return fail(KMC_ERR_UNSUPPORTED, "a body density runs in the one-walker-per-lane kernels only");
    char key[112];
    std::snprintf(key, sizeof(key), "%d:%d,%d,%d,%d|%d,%d|%d|%d|%lld|%d|%d", (int)with_vec, L, K, iter, (int)ragged, resident_K,
                  (int)resident_ragged, island_S, (int)f32, ud->is_body ? (long long)ndim : 0ll, (int)staged_possible(ud, f32, ndim, p2p), (int)p2p);
    const char* peer = p2p ? "true" : "false";         // KMC_P2P: partner rows read from their owners (pull)
    const char* rowt = f32 ? "float" : "double";       // storage type of the walker rows (KMC_F32 / KMC_F64)
    std::lock_guard<std::mutex> lock(ud->mu);
    auto it = ud->code.find(key);
    if (it != ud->code.end()) { *out = &it->second; return KMC_OK; }

    const char* envdir = std::getenv("KMC_CSRC_DIR");
    const std::string dir = envdir ? std::string(envdir) : library_dir() + "/csrc";
    const std::string h_dev = read_file(dir + "/kmc_device.hpp"), h_ker = read_file(dir + "/kmc_kernels.hpp"),
                      h_isl = read_file(dir + "/kmc_islands.hpp");
    if (h_dev.empty() || h_ker.empty() || h_isl.empty())
        return fail(KMC_ERR_BAD_ARG, "user density: kernel headers not found in " + dir + " (set KMC_CSRC_DIR)");

    std::ostringstream src;
    src << "#include \"kmc_islands.hpp\"\n" << user_functor_source(ud) << user_density_alias(ud, ndim)
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_generic(KMC_FRONT_PARAMS, const kmc::HalfStepArgs a) { kmc::half_step_generic_body<UD, " << peer << ", " << rowt << ">(KMC_FRONT_PACK, a); }\n"
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_logpdf(const kmc::LogpdfArgs a) { kmc::logpdf_rows_body<UD>(a); }\n"
        << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_init_ball(const kmc::InitBallArgs a) { kmc::init_ball_body<UD>(a); }\n";
    if (staged_possible(ud, f32, ndim, p2p))
        src << "extern \"C\" __global__ __launch_bounds__(" << kStagedTPB << ") void kmc_user_staged(KMC_FRONT_PARAMS, const kmc::HalfStepArgs a) { kmc::half_step_staged_body<UD, "
            << ndim << ">(KMC_FRONT_PACK, a); }\n";
    if (with_vec)
        src << "extern \"C\" __global__ __launch_bounds__(" << vec_tpb(L) << ") void kmc_user_vec(KMC_FRONT_PARAMS, const kmc::HalfStepArgs a) { kmc::half_step_vec_body<UD, "
            << L << ", " << K << ", " << iter << ", " << peer << ", " << (ragged ? "true" : "false") << ", " << rowt << ">(KMC_FRONT_PACK, a); }\n";
    if (resident_K > 0 && island_S == 0)
        src << "extern \"C\" __global__ __launch_bounds__(256) void kmc_user_resident(const kmc::ResidentArgs a) { kmc::resident_body<UD, "
            << resident_K << ", " << (resident_ragged ? "true" : "false") << ">(a); }\n";
    if (resident_K > 0 && island_S > 0)
        src << "extern \"C\" __global__ __launch_bounds__(" << island_S << ") void kmc_user_island(const kmc::IslandArgs a) { kmc::island_epoch_body<UD, "
            << island_S << ", " << resident_K << ", " << (resident_ragged ? "true" : "false") << ">(a); }\n";
    const std::string text = src.str();

    const char* headers[3] = {h_ker.c_str(), h_dev.c_str(), h_isl.c_str()};
    const char* names[3] = {"kmc_kernels.hpp", "kmc_device.hpp", "kmc_islands.hpp"};
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=14"};
    std::vector<char> code;
    std::string log;
    const kmc_status cst = rtc_compile_cached(text, "kmc_user_density.hip", 3, headers, names, 6, opts, &code, &log);
    if (cst == KMC_ERR_BAD_ARG) return fail(KMC_ERR_BAD_ARG, "user density does not compile:\n" + log);
    if (cst != KMC_OK) return cst;
    auto ins = ud->code.emplace(key, std::move(code));
    *out = &ins.first->second;
    return KMC_OK;
}

kmc_status load_user(kmc_user_density* ud, bool with_vec, int L, int K, int iter, bool ragged, UserKernels* uk,
                     int resident_K = 0, bool resident_ragged = false, int island_S = 0, bool f32 = false, int64_t ndim = 0, bool p2p = false)
{
    const std::vector<char>* code = nullptr;
    KMC_TRY(compile_user(ud, with_vec, L, K, iter, ragged, resident_K, resident_ragged, island_S, f32, &code, ndim, p2p));
    HIP_TRY(hipModuleLoadData(&uk->mod, code->data()));
    HIP_TRY(hipModuleGetFunction(&uk->generic, uk->mod, "kmc_user_generic"));
    HIP_TRY(hipModuleGetFunction(&uk->logpdf, uk->mod, "kmc_user_logpdf"));
    HIP_TRY(hipModuleGetFunction(&uk->init_ball, uk->mod, "kmc_user_init_ball"));
    if (with_vec) HIP_TRY(hipModuleGetFunction(&uk->vec, uk->mod, "kmc_user_vec"));
    if (staged_possible(ud, f32, ndim, p2p)) HIP_TRY(hipModuleGetFunction(&uk->staged, uk->mod, "kmc_user_staged"));
    if (resident_K > 0 && island_S == 0) HIP_TRY(hipModuleGetFunction(&uk->resident, uk->mod, "kmc_user_resident"));
    if (resident_K > 0 && island_S > 0) HIP_TRY(hipModuleGetFunction(&uk->island, uk->mod, "kmc_user_island"));
    return KMC_OK;
}


}  // namespace

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
    *out = ud;
    return KMC_OK;
}

KMC_EXPORT void kmc_user_density_destroy(kmc_user_density* ud) { delete ud; }

struct kmc_sampler {
    kmc_config cfg{};
    int64_t h = 0, h_loc = 0, active_begin = 0, nlocal = 0, nsamples = 0;
    int64_t ld = 0;                    // device row stride in doubles (ndim rounded up to even)
    DensityParams dp{};
    Plan plan{};
    LogpdfFn logpdf_fn = nullptr;
    kmc_user_density* user = nullptr;     // KMC_USER_DENSITY: kernels come from a runtime-compiled module
    UserKernels uk{};
    int grid = 0;
    int tpb = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double* d_pos = nullptr;           // rows [nrows][ld]; float elements when f32 (KMC_F32)
    bool f32 = false;
    bool own_pos = true;
    double* d_logp = nullptr;
    uint32_t* d_naccept = nullptr;
    int64_t* d_gen = nullptr;
    SchedEntry* d_sched = nullptr;
    double* d_chain = nullptr;
    double* d_chain_logp = nullptr;
    double* d_msum = nullptr;
    double* d_msumsq = nullptr;
    double2* d_mring = nullptr;       // moment ring (HalfStepArgs::mring): [waves][mring_depth][K][64] rows
    double* d_mring_w = nullptr;      //   and [waves][mring_depth] weights
    uint32_t* d_mcnt = nullptr;       // [waves] entries posted, then [waves] entries swept (one allocation)
    int mring_depth = 0;
    int64_t mring_waves = 0;
    int64_t gens_since_sweep = 0;
    uint32_t* d_klast = nullptr;      // vec kernels: samples already credited per walker (d_logp, d_naccept, d_klast: one block)
    double2* d_ring = nullptr;        // vec kernels: parked draws of the walkers' next steps, [4][nrows] x 32 B (HalfStepArgs::ring)
    int64_t macc_stride = 0, macc_elems = 0;
    // KMC_STREAM_CHAIN: d_chain / d_chain_logp are rings of ring_slots = 3 * ring_blk sample slots; completed blocks go to
    // the caller's host buffers on copy_stream while sampling goes on
    bool stream_chain = false;
    int64_t ring_blk = 0, ring_slots = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_filled[3] = {}, ev_copied[3] = {};
    double* dst_chain = nullptr;      // host [nsamples][nlocal][ndim]
    double* dst_logp = nullptr;       // host [nsamples][nlocal]
    bool dst_chain_reg = false, dst_logp_reg = false;      // page-locked in place by us
    int64_t blocks_copied = 0;        // blocks [0, blocks_copied) have their device-to-host copy enqueued
    int64_t blocks_waited = 0;        // compute stream already waits for the copies of the blocks that blocks < this overwrite
    void* comm = nullptr;             // replica sharding: RCCL communicator (kmc_sampler_rccl_init) for the all-gather after each half-step
    bool comm_graph_ok = true;        //   the all-gather can be captured into the hipGraph chunks (decided at the first capture)
    uint32_t* d_ids = nullptr;        // dealt sub-ensembles: global walker index held by each slot
    uint64_t user_seed = 0;           //   the caller's seed (cfg.seed is then this sub-ensemble's Philox key)
    int64_t moment_base = 0;  // samples that precede the restored state (kmc_sampler_set_state)
    int64_t generation = 0;   // generations enqueued so far
    int64_t dev_gen = 0;      // value the device counter will hold once the stream drains
    int64_t launches = 0;
    int launch_mode = 0;      // 0: not decided, 1: table graph, 2: eager launches, 3: updated graph, KMC_LAUNCH=updated only (kmc_sampler_run)
    float calib_graph_ms = 0.f, calib_eager_ms = 0.f;   // one chunk each, when measured
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    // "updated graph": a chain of kGraphChunk * 2 kernel nodes launched in the eager form (step among the preloaded
    // parameters, schedule entry in the args), their parameters rewritten before every replay; kUExec executables
    // take turns, so the host updates up to kUExec - 1 replays ahead of the one that is running (two were enough for
    // a quiet host -- the update of 128 nodes takes about as long as their replay -- but left a single replay of
    // slack: one scheduling hiccup of the host process starved the GPU, 4.27 instead of 3.8 us per launch in one run)
    hipGraph_t ugraph = nullptr;
    hipGraphExec_t uexec[kUExec] = {};
    hipEvent_t udone[kUExec] = {};
    bool uinflight[kUExec] = {};
    int unext = 0;
    std::vector<hipGraphNode_t> unodes;
    int64_t uchunk = 64;      // generations per replay of the updated graph
    bool updated_forced = false;   // KMC_LAUNCH=updated: no budget
    std::vector<std::pair<char*, size_t>> guards;      // KMC_POISON: (guard address, size of the allocation in front of it)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_run_events = false;
    bool positions_set = false;
    // host-evaluated density (KMC_HOST_DENSITY)
    bool host_eval = false;
    double* d_prop = nullptr;          // [h][ld] proposals of the current half-step
    double* d_p1 = nullptr;            // [h] their log-pdfs, as returned by the callback
    double* h_prop = nullptr;          // pinned, dense [h][ndim]
    double* h_p1 = nullptr;            // pinned [h]
    uint8_t* d_acc = nullptr;          // [h] accept outcomes of the current half-step (host_accepted only)
    uint8_t* h_acc = nullptr;          // pinned [h]
    // resident mode: exact sampler, whole (small) ensemble in one workgroup's LDS, many generations per launch
    bool resident = false;
    ResidentFn resident_kernel = nullptr;
    int resident_tpb = 256;
    // island mode (KMC_ISLANDS)
    bool islands = false;
    IslandFn island_kernel = nullptr;
    int island_K = 0;
    bool island_ragged = false;
    int64_t island_gens = 32, nislands = 0, island_size = kIslandSizeDefault;
    size_t island_lds = 0;
    double* d_isum = nullptr;                            // [nislands][4K] per-island moment sums
    double* d_isumsq = nullptr;
    // peer-to-peer sharding (KMC_P2P)
    bool p2p = false;
    bool connected = false;
    int64_t nrows = 0;                                   // rows held by this sampler (nwalkers, or nlocal for P2P)
    unsigned long long* d_flags = nullptr;               // fine-grained progress flags [shard_count]
    unsigned long long* d_err = nullptr;
    uint32_t* d_done = nullptr;                          // KMC_P2P_FOLD_SIGNAL: workgroups drained, per launch
    bool fold_signal = false;
    bool stream_by_walker = false;                       // KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER: host buffers are [walker][nsamples][..]
    double *dev_dst_chain = nullptr, *dev_dst_logp = nullptr;   // ... and these are their device-side addresses (page-locked)
    double *bw_scratch = nullptr, *bw_scratch_logp = nullptr;   // ... or one transposed block on the device, copied out as a 2-D window
    int64_t flushed_done = -1;                           // samples_done at the last flush of an incomplete block (nothing new: skip it)
    bool push = false;                                   // KMC_P2P_PUSH / KMC_P2P_LAZY: d_pos = (1 + shard_count) blocks, see HalfStepArgs::push
    bool lazy = false;                                   // KMC_P2P_LAZY: + accept-byte maps behind the blocks, stamps in d_lazy
    bool lazy_stats = false;                             // KMC_P2P_STATS=1: count remote draws / pulls (kmc_sampler_p2p_stats)
    unsigned char* d_lazy = nullptr;                     // {stamps[P][2][h_loc] {fetched, modified}, stats[2]}
    unsigned char* peer_amap_in[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* peer_pos[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long* peer_flags[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace {

// graph_mode: the generation is (device counter) + gen_offset, looked up in the device schedule
// table; otherwise gen_offset is the absolute generation and its schedule travels in the args.
HalfStepArgs make_args(const kmc_sampler* s, int half, bool graph_mode, int64_t gen_offset)
{
    HalfStepArgs a{};
    a.pos = s->d_pos;
    a.logp = s->d_logp;
    a.naccept = s->d_naccept;
    a.sched_table = s->d_sched;
    a.sched_index = graph_mode ? (int32_t)gen_offset : -1;
    a.sched_inline = make_sched(gen_offset, s->cfg.nburnin, s->cfg.nthin, s->nsamples, s->ring_slots);
    a.gw0 = (int64_t)half * s->h + s->active_begin;
    a.own_row0 = s->p2p ? (int64_t)half * s->h_loc : a.gw0;
    a.oth_row0 = s->p2p ? (int64_t)(1 - half) * s->h_loc : (int64_t)(1 - half) * s->h;
    a.hloc = (uint32_t)s->h_loc;
    a.hloc_shift = -1;
    if (s->h_loc > 0 && (s->h_loc & (s->h_loc - 1)) == 0) { a.hloc_shift = 0; while (((int64_t)1 << a.hloc_shift) < s->h_loc) ++a.hloc_shift; }
    a.nranks = s->cfg.shard_count;
    for (int r = 0; r < 8; ++r) a.peer_pos[r] = s->peer_pos[r];
    a.flags = s->d_flags;
    a.err = s->d_err;
    for (int r = 0; r < 8; ++r) a.peer_flags[r] = s->peer_flags[r];
    a.done_count = s->fold_signal ? s->d_done : nullptr;
    a.me = s->cfg.shard_rank;
    a.push = s->lazy ? 2 : s->push ? 1 : 0;
    if (s->lazy) {
        const size_t hl = (size_t)s->h_loc, P = (size_t)s->cfg.shard_count;
        a.lz_amap_in = s->peer_amap_in[s->cfg.shard_rank];
        a.lz_stamps = reinterpret_cast<uint2*>(s->d_lazy);
        a.lz_stats = s->lazy_stats ? reinterpret_cast<unsigned long long*>(a.lz_stamps + P * 2 * hl) : nullptr;
        for (int r = 0; r < 8; ++r) a.lz_peer_amap[r] = s->peer_amap_in[r];
    }
    a.shard_stride = (int64_t)s->nrows * s->ld;
    a.n_active = (int32_t)s->h_loc;
    a.half = half;
    a.ndim = (int32_t)s->cfg.ndim;
    a.ld = (int32_t)s->ld;
    a.dc.seed_lo = (uint32_t)s->cfg.seed;
    a.dc.seed_hi = (uint32_t)(s->cfg.seed >> 32);
    a.dc.nhalf = (uint32_t)s->h;
    a.dc.c0 = std::sqrt(1.0 / s->cfg.a_scale);                            // src/samplers.jl:227
    a.dc.c1 = std::sqrt(s->cfg.a_scale) - std::sqrt(1.0 / s->cfg.a_scale);
    a.dc.nm1 = (double)(s->cfg.ndim - 1);
    a.dp = s->dp;
    a.chain = s->d_chain;
    a.chain_logp = s->d_chain_logp;
    a.chain_rows = s->nlocal;
    a.chain_row0 = (int64_t)half * s->h_loc;
    a.msum = s->d_msum;
    a.msumsq = s->d_msumsq;
    a.macc_stride = s->macc_stride;
    a.klast = s->d_klast;
    a.mring = s->d_mring;
    a.mring_w = s->d_mring_w;
    a.mcnt = s->d_mcnt;
    a.mswept = s->d_mcnt ? s->d_mcnt + s->mring_waves : nullptr;
    a.mring_depth = s->mring_depth;
    a.ring = s->d_ring;
    a.ring_rows = s->nrows;
    a.ring_slot = (int32_t)(gen_offset & 3);
    return a;
}

// the leading scalar kernel parameters (kernarg preload, see HalfStepFront)
HalfStepFront front_of(const HalfStepArgs& a)
{
    HalfStepFront f{};
    f.pos = a.pos;
    f.sched = a.sched_index < 0 ? nullptr : a.sched_table + a.sched_index;
    f.step = (uint32_t)(2ull * (uint64_t)a.sched_inline.gen + (uint64_t)a.half);       // used when sched == nullptr
    f.gw0 = (uint32_t)a.gw0;
    f.logp = a.logp;
    f.nact_half = (uint32_t)a.n_active | ((uint32_t)a.half << 31);
    f.ring_now = a.ring ? a.ring + (int64_t)a.ring_slot * a.ring_rows * 2 : nullptr;
    f.seed_lo = a.dc.seed_lo; f.seed_hi = a.dc.seed_hi; f.nhalf = a.dc.nhalf;
    return f;
}

// kernarg image of (KMC_FRONT_PARAMS, const HalfStepArgs): the scalars at their natural alignment, then the struct
struct HalfStepLaunch {
    HalfStepFront f;
    HalfStepArgs  a;
};
static_assert(offsetof(HalfStepLaunch, a) == 56 && offsetof(HalfStepFront, ring_now) == 16 && offsetof(HalfStepFront, step) == 52, "kernarg layout of the half-step kernels");

hipError_t launch_half_kernel(const kmc_sampler* s, const HalfStepArgs& a);

kmc_status launch_half(kmc_sampler* s, int half, bool graph_mode, int64_t gen_offset)
{
    const HalfStepArgs a = make_args(s, half, graph_mode, gen_offset);
    HIP_TRY(launch_half_kernel(s, a));
    if (s->p2p && s->cfg.shard_count > 1 && !s->fold_signal) {
        // the kernel boundary puts this half-step's rows in memory; then publish the progress
        SignalArgs sg{};
        for (int r = 0; r < 8; ++r) sg.peer_flags[r] = s->peer_flags[r];
        sg.nranks = s->cfg.shard_count;
        sg.me = s->cfg.shard_rank;
        sg.sched_table = a.sched_table;
        sg.sched_inline = a.sched_inline;
        sg.sched_index = a.sched_index;
        sg.half = half;
        hipLaunchKernelGGL(p2p_signal, dim3(1), dim3(64), 0, s->stream, sg);
        HIP_TRY(hipGetLastError());
    }
    if (s->comm) {
        // replica sharding: every rank's slice of the half just updated, gathered in place (the join of :273, across GPUs)
        double* base = s->d_pos + (size_t)half * (size_t)s->h * (size_t)s->ld;
        const size_t count = (size_t)s->h_loc * (size_t)s->ld;
        KMC_TRY(rccl_all_gather_f64(s->comm, base + (size_t)s->cfg.shard_rank * count, base, count, s->stream));
    }
    return KMC_OK;
}

hipError_t launch_half_kernel(const kmc_sampler* s, const HalfStepArgs& a)
{
    const HalfStepFront f = front_of(a);
    if (s->user) {
        const HalfStepLaunch la{f, a};
        if (s->uk.staged) return launch_module(s->uk.staged, (unsigned)s->grid, (unsigned)s->tpb, s->stream, la, (unsigned)staged_lds_bytes((int)s->cfg.ndim));
        return launch_module(s->plan.vec ? s->uk.vec : s->uk.generic, (unsigned)s->grid, (unsigned)s->tpb, s->stream, la);
    }
    hipLaunchKernelGGL(s->plan.fn, dim3(s->grid), dim3(s->tpb), 0, s->stream, f.pos, f.sched, f.ring_now, f.logp, f.gw0, f.nact_half,
                       f.seed_lo, f.seed_hi, f.nhalf, f.step, a);
    return hipGetLastError();
}

// Fold the moment ring's posted entries into the accumulators (between graph chunks / before a read-out).
constexpr int64_t kSweepEvery = 64;        // generations between sweeps of eager launches (a graph chunk is 64 too)
hipError_t launch_sweep(kmc_sampler* s)
{
    s->gens_since_sweep = 0;
    if (!s->d_mring) return hipSuccess;
    SweepArgs a{};
    a.ring = s->d_mring; a.ring_w = s->d_mring_w; a.cnt = s->d_mcnt; a.swept = s->d_mcnt + s->mring_waves;
    a.msum = s->d_msum; a.msumsq = s->d_msumsq; a.macc_stride = s->macc_stride;
    a.K = s->plan.K; a.depth = s->mring_depth;
    hipLaunchKernelGGL(moments_sweep, dim3((unsigned)(s->mring_waves * s->plan.K)), dim3(64), 0, s->stream, a);
    hipLaunchKernelGGL(moments_swept, dim3((unsigned)((s->mring_waves + 255) / 256)), dim3(256), 0, s->stream,
                       s->d_mcnt, s->d_mcnt + s->mring_waves, s->mring_waves);
    return hipGetLastError();
}

void launch_advance(kmc_sampler* s, int n, int64_t by)
{
    hipLaunchKernelGGL(advance_schedule, dim3(1), dim3(64), 0, s->stream, s->d_gen, s->d_sched, n, by,
                       s->cfg.nburnin, s->cfg.nthin, s->nsamples, s->ring_slots);
}

kmc_status sync_device_counter(kmc_sampler* s)
{
    if (s->dev_gen != s->generation) {
        launch_advance(s, 0, s->generation - s->dev_gen);
        HIP_TRY(hipGetLastError());
        s->dev_gen = s->generation;
    }
    return KMC_OK;
}

kmc_status ensure_graph(kmc_sampler* s)
{
    if (s->graph_exec) return KMC_OK;
    HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
    kmc_status st = KMC_OK;
    launch_advance(s, (int)kGraphChunk, 0);                 // schedule table of this chunk
    for (int64_t g = 0; g < kGraphChunk && st == KMC_OK; ++g)
        for (int half = 0; half < 2 && st == KMC_OK; ++half) st = launch_half(s, half, true, g);
    if (st == KMC_OK) launch_advance(s, 0, kGraphChunk);   // device counter += chunk
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(s->stream, &graph);
    if (st != KMC_OK || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (s->comm) { s->comm_graph_ok = false; return KMC_OK; }   // RCCL refused the capture: this sampler launches eagerly
        if (st != KMC_OK) return st;
        HIP_TRY(e);
    }
    s->graph = graph;
    const hipError_t ei = hipGraphInstantiate(&s->graph_exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(graph);
        s->graph = nullptr; s->graph_exec = nullptr;
        if (s->comm) { s->comm_graph_ok = false; return KMC_OK; }
        HIP_TRY(ei);
    }
    return KMC_OK;
}

// ---- updated graph --------------------------------------------------------------------------------------
struct KernelParamPack {      // storage the kernelParams pointers of one node refer to
    HalfStepFront f;
    HalfStepArgs a;
    void* ptrs[11];
    void bind()
    {
        ptrs[0] = &f.pos; ptrs[1] = &f.sched; ptrs[2] = &f.ring_now; ptrs[3] = &f.logp; ptrs[4] = &f.gw0; ptrs[5] = &f.nact_half;
        ptrs[6] = &f.seed_lo; ptrs[7] = &f.seed_hi; ptrs[8] = &f.nhalf; ptrs[9] = &f.step; ptrs[10] = &a;
    }
};

hipKernelNodeParams node_params(const kmc_sampler* s, KernelParamPack* pk)
{
    hipKernelNodeParams np{};
    np.func = reinterpret_cast<void*>(s->plan.fn);
    np.gridDim = dim3((unsigned)s->grid);
    np.blockDim = dim3((unsigned)s->tpb);
    np.sharedMemBytes = 0;
    np.kernelParams = pk->ptrs;
    np.extra = nullptr;
    return np;
}

bool updated_graph_possible(const kmc_sampler* s)
{
    return !s->user && !s->p2p && !s->host_eval && !s->islands && !s->resident && !s->comm && s->plan.fn != nullptr;
}

kmc_status ensure_updated_graph(kmc_sampler* s)
{
    if (s->uexec[0]) return KMC_OK;
    HIP_TRY(hipGraphCreate(&s->ugraph, 0));
    s->unodes.assign((size_t)(2 * s->uchunk), nullptr);
    KernelParamPack pk;
    pk.bind();
    hipGraphNode_t prev = nullptr;
    for (int64_t g = 0; g < s->uchunk; ++g)
        for (int half = 0; half < 2; ++half) {
            pk.a = make_args(s, half, false, g);
            pk.f = front_of(pk.a);
            const hipKernelNodeParams np = node_params(s, &pk);
            hipGraphNode_t node = nullptr;
            HIP_TRY(hipGraphAddKernelNode(&node, s->ugraph, prev ? &prev : nullptr, prev ? 1 : 0, &np));
            s->unodes[(size_t)(2 * g + half)] = node;
            prev = node;
        }
    for (int i = 0; i < kUExec; ++i) {
        HIP_TRY(hipGraphInstantiate(&s->uexec[i], s->ugraph, nullptr, nullptr, 0));
        HIP_TRY(hipEventCreateWithFlags(&s->udone[i], hipEventDisableTiming));
    }
    return KMC_OK;
}

// hipGraphExecKernelNodeSetParams leaks ~80 bytes per call inside the runtime (kmc_sampler_run: launch modes): a process-wide
// budget of such calls -- 64 MiB worth by default
std::atomic<int64_t> g_update_calls{0};
bool update_budget_left()
{
    static const int64_t budget = [] {
        double mb = 64.0;
        if (const char* e = std::getenv("KMC_UPDATED_BUDGET_MB")) mb = std::atof(e);
        return (int64_t)(mb * 1048576.0 / 80.0);
    }();
    return g_update_calls.load(std::memory_order_relaxed) < budget;
}

// one replay of kGraphChunk generations starting at s->generation
kmc_status launch_updated_graph(kmc_sampler* s)
{
    KMC_TRY(ensure_updated_graph(s));
    g_update_calls.fetch_add(2 * s->uchunk, std::memory_order_relaxed);
    const int i = s->unext;
    if (s->uinflight[i]) { HIP_TRY(hipEventSynchronize(s->udone[i])); s->uinflight[i] = false; }
    KernelParamPack pk;
    pk.bind();
    for (int64_t g = 0; g < s->uchunk; ++g)
        for (int half = 0; half < 2; ++half) {
            pk.a = make_args(s, half, false, s->generation + g);
            pk.f = front_of(pk.a);
            const hipKernelNodeParams np = node_params(s, &pk);
            HIP_TRY(hipGraphExecKernelNodeSetParams(s->uexec[i], s->unodes[(size_t)(2 * g + half)], &np));
        }
    HIP_TRY(hipGraphLaunch(s->uexec[i], s->stream));
    HIP_TRY(hipEventRecord(s->udone[i], s->stream));
    s->uinflight[i] = true;
    s->unext = (i + 1) % kUExec;
    return KMC_OK;
}

// Host rows are dense [rows][ndim]; device rows have stride ld (= ndim, or ndim+1 for odd ndim).
// KMC_F32: the device rows are float (dst_dev / src_dev are then the raw buffers); the host side stays double.
hipError_t upload_rows(const kmc_sampler* s, double* dst_dev, const double* src_host, size_t rows)
{
    const size_t nd = (size_t)s->cfg.ndim, ld = (size_t)s->ld;
    if (rows == 0) return hipSuccess;
    if (s->f32) {
        std::vector<float> t(rows * ld, 0.0f);
        for (size_t r = 0; r < rows; ++r)
            for (size_t d = 0; d < nd; ++d) t[r * ld + d] = (float)src_host[r * nd + d];
        return copy_sync(dst_dev, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, s->stream);
    }
    if (ld == nd) return copy_sync(dst_dev, src_host, rows * nd * sizeof(double), hipMemcpyHostToDevice, s->stream);
    // padded rows (odd ndim): repacked on the host and copied contiguously -- not hipMemcpy2DAsync from pageable memory (an
    // abort inside the runtime, intermittent, was traced to kmc_sampler_set_positions with odd ndim while that was in use)
    std::vector<double> t(rows * ld, 0.0);
    for (size_t r = 0; r < rows; ++r) std::memcpy(&t[r * ld], src_host + r * nd, nd * sizeof(double));
    return copy_sync(dst_dev, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, s->stream);
}
hipError_t download_rows(const kmc_sampler* s, double* dst_host, const double* src_dev, size_t rows)
{
    const size_t nd = (size_t)s->cfg.ndim, ld = (size_t)s->ld;
    if (rows == 0) return hipSuccess;
    if (s->f32) {
        const size_t slab = (size_t)1 << 22;                   // rows per copy: bounds the host staging buffer
        std::vector<float> t((rows < slab ? rows : slab) * ld);
        for (size_t r0 = 0; r0 < rows; r0 += slab) {
            const size_t n = rows - r0 < slab ? rows - r0 : slab;
            const hipError_t e = copy_sync(t.data(), reinterpret_cast<const float*>(src_dev) + r0 * ld, n * ld * sizeof(float), hipMemcpyDeviceToHost, s->stream);
            if (e != hipSuccess) return e;
            for (size_t r = 0; r < n; ++r)
                for (size_t d = 0; d < nd; ++d) dst_host[(r0 + r) * nd + d] = (double)t[r * ld + d];
        }
        return hipSuccess;
    }
    if (ld == nd) return copy_sync(dst_host, src_dev, rows * nd * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    const size_t slab = (size_t)1 << 21;                       // padded rows: contiguous copies of slabs, unpacked on the host (see upload_rows)
    std::vector<double> t((rows < slab ? rows : slab) * ld);
    for (size_t r0 = 0; r0 < rows; r0 += slab) {
        const size_t n = rows - r0 < slab ? rows - r0 : slab;
        const hipError_t e = copy_sync(t.data(), src_dev + r0 * ld, n * ld * sizeof(double), hipMemcpyDeviceToHost, s->stream);
        if (e != hipSuccess) return e;
        for (size_t r = 0; r < n; ++r) std::memcpy(dst_host + (r0 + r) * nd, &t[r * ld], nd * sizeof(double));
    }
    return hipSuccess;
}

int64_t samples_done(const kmc_sampler* s)
{
    const int64_t post = s->generation - s->cfg.nburnin;
    if (post <= 0) return 0;
    const int64_t k = post / s->cfg.nthin;
    return k < s->nsamples ? k : s->nsamples;
}

// ---- KMC_STREAM_CHAIN ---------------------------------------------------------------------------------------
constexpr unsigned kStreamWalkerGrid = 192;      // workgroups of the background by-walker copy kernel (KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER)
int64_t samples_done_at(const kmc_sampler* s, int64_t generation)
{
    const int64_t post = generation - s->cfg.nburnin;
    if (post <= 0) return 0;
    const int64_t k = post / s->cfg.nthin;
    return k < s->nsamples ? k : s->nsamples;
}

// device ring -> host, samples [k0, k1) of ONE block, on the copy stream
kmc_status chain_copy_range(kmc_sampler* s, int64_t k0, int64_t k1)
{
    if (k1 <= k0) return KMC_OK;
    const size_t nl = (size_t)s->nlocal, nd = (size_t)s->cfg.ndim, ld = (size_t)s->ld, n = (size_t)(k1 - k0);
    const size_t slot0 = (size_t)(k0 % s->ring_slots);
    if (s->stream_by_walker) {
        // the reference's order: the caller's arrays are [walker][nsamples][ndim], a block is a run of n * ndim doubles
        // per walker.  Either transposed into a device scratch block and copied by the DMA engine as a 2-D window
        // (default), or written straight into the page-locked arrays by the kernel (KMC_BYWALKER_COPY=kernel; few
        // workgroups, so that it drains over PCIe without taking the sampler's wave slots).
        const int64_t ns = s->nsamples;
        if (s->d_chain && s->dst_chain) {
            if (s->bw_scratch) {
                hipLaunchKernelGGL(chain_by_walker<double>, dim3((unsigned)std::min<size_t>(nl, 65535u), 1), dim3(256), 0, s->copy_stream, s->d_chain + slot0 * nl * ld,
                                   s->bw_scratch, (int64_t)nl, (int32_t)ld, (int32_t)nd, (int64_t)n, (int64_t)0, (int64_t)nl, (int64_t)(n * nd));
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpy2DAsync(s->dst_chain + (size_t)k0 * nd, (size_t)ns * nd * sizeof(double), s->bw_scratch, n * nd * sizeof(double),
                                         n * nd * sizeof(double), nl, hipMemcpyDeviceToHost, s->copy_stream));
            } else {
                hipLaunchKernelGGL(chain_by_walker<double>, dim3(kStreamWalkerGrid, 1), dim3(256), 0, s->copy_stream, s->d_chain + slot0 * nl * ld,
                                   s->dev_dst_chain + (size_t)k0 * nd, (int64_t)nl, (int32_t)ld, (int32_t)nd, (int64_t)n, (int64_t)0, (int64_t)nl, ns * (int64_t)nd);
                HIP_TRY(hipGetLastError());
            }
        }
        if (s->d_chain_logp && s->dst_logp) {
            if (s->bw_scratch_logp) {
                hipLaunchKernelGGL(chain_by_walker<double>, dim3((unsigned)std::min<size_t>(nl, 65535u), 1), dim3(256), 0, s->copy_stream, s->d_chain_logp + slot0 * nl,
                                   s->bw_scratch_logp, (int64_t)nl, (int32_t)1, (int32_t)1, (int64_t)n, (int64_t)0, (int64_t)nl, (int64_t)n);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpy2DAsync(s->dst_logp + (size_t)k0, (size_t)ns * sizeof(double), s->bw_scratch_logp, n * sizeof(double), n * sizeof(double), nl,
                                         hipMemcpyDeviceToHost, s->copy_stream));
            } else {
                hipLaunchKernelGGL(chain_by_walker<double>, dim3(kStreamWalkerGrid, 1), dim3(256), 0, s->copy_stream, s->d_chain_logp + slot0 * nl,
                                   s->dev_dst_logp + (size_t)k0, (int64_t)nl, (int32_t)1, (int32_t)1, (int64_t)n, (int64_t)0, (int64_t)nl, ns);
                HIP_TRY(hipGetLastError());
            }
        }
        return KMC_OK;
    }
    if (s->d_chain && s->dst_chain) {
        const double* src = s->d_chain + slot0 * nl * ld;
        double* dst = s->dst_chain + (size_t)k0 * nl * nd;
        const bool locked = s->dst_chain_reg;            // else: a blocking copy through the bounce buffers (copy_sync), never an
                                                         //   asynchronous copy into pageable memory
        if (ld == nd) {
            if (locked) HIP_TRY(hipMemcpyAsync(dst, src, n * nl * nd * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
            else HIP_TRY(copy_sync(dst, src, n * nl * nd * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        } else {                                         // padded rows: compacted into the scratch block, then one contiguous copy
            int64_t grid = (int64_t)((n * nl * nd + 255) / 256);
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(rows_compact, dim3((unsigned)grid), dim3(256), 0, s->copy_stream, src, s->bw_scratch, (int64_t)(n * nl), (int32_t)ld, (int32_t)nd);
            HIP_TRY(hipGetLastError());
            if (locked) HIP_TRY(hipMemcpyAsync(dst, s->bw_scratch, n * nl * nd * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
            else HIP_TRY(copy_sync(dst, s->bw_scratch, n * nl * nd * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        }
    }
    if (s->d_chain_logp && s->dst_logp) {
        if (s->dst_logp_reg) HIP_TRY(hipMemcpyAsync(s->dst_logp + (size_t)k0 * nl, s->d_chain_logp + slot0 * nl, n * nl * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        else HIP_TRY(copy_sync(s->dst_logp + (size_t)k0 * nl, s->d_chain_logp + slot0 * nl, n * nl * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
    }
    return KMC_OK;
}

// Before enqueueing generations [.., g_end): the ring positions they will write must have been drained to the host.
kmc_status chain_before(kmc_sampler* s, int64_t g_end)
{
    if (!s->stream_chain) return KMC_OK;
    const int64_t k1 = samples_done_at(s, g_end);
    if (k1 <= 0) return KMC_OK;
    const int64_t bmax = (k1 - 1) / s->ring_blk;
    if (bmax - 3 >= s->blocks_copied)
        return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN: one launch unit would lap the device ring (internal)");
    for (; s->blocks_waited <= bmax; ++s->blocks_waited)
        if (s->blocks_waited >= 3)      // block b overwrites the ring position of block b - 3: its copy must be over
            HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_copied[s->blocks_waited % 3], 0));
    return KMC_OK;
}

// After enqueueing up to s->generation: every block that is complete now goes to the host behind the sampling.
kmc_status chain_after(kmc_sampler* s)
{
    if (!s->stream_chain) return KMC_OK;
    const int64_t done = samples_done(s);
    while ((s->blocks_copied + 1) * s->ring_blk <= done) {
        const int64_t b = s->blocks_copied;
        const int r = (int)(b % 3);
        HIP_TRY(hipEventRecord(s->ev_filled[r], s->stream));
        HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_filled[r], 0));
        KMC_TRY(chain_copy_range(s, b * s->ring_blk, (b + 1) * s->ring_blk));
        HIP_TRY(hipEventRecord(s->ev_copied[r], s->copy_stream));
        s->blocks_copied = b + 1;
    }
    return KMC_OK;
}

// At a synchronisation point: the samples of the last, incomplete block as well (it is copied again, whole, once complete).
kmc_status chain_flush(kmc_sampler* s)
{
    if (!s->stream_chain) return KMC_OK;
    KMC_TRY(chain_after(s));
    const int64_t done = samples_done(s), k0 = s->blocks_copied * s->ring_blk;
    if (done > k0 && done != s->flushed_done) {      // (an unchanged tail is in the host arrays already)
        HIP_TRY(hipStreamSynchronize(s->stream));
        KMC_TRY(chain_copy_range(s, k0, done));
        s->flushed_done = done;
    }
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    return KMC_OK;
}

void chain_unregister(kmc_sampler* s)
{
    if (s->dst_chain_reg) { (void)hipHostUnregister(s->dst_chain); s->dst_chain_reg = false; }
    if (s->dst_logp_reg) { (void)hipHostUnregister(s->dst_logp); s->dst_logp_reg = false; }
    (void)hipGetLastError();
}

// KMC_P2P: a half-step kernel that gave up waiting for a peer has flagged it in d_err, and everything computed after that
// is invalid -- every read-out of a P2P sampler checks (call after the stream has drained).
kmc_status check_p2p_err(kmc_sampler* s)
{
    if (!s->p2p || !s->d_err) return KMC_OK;
    unsigned long long e = 0;
    HIP_TRY(copy_sync(&e, s->d_err, sizeof(e), hipMemcpyDeviceToHost, s->stream));
    if (e != 0)
        return fail(KMC_ERR_HIP, "p2p: timed out waiting for a peer before half-step " + std::to_string(e - 1) + " (results are invalid)");
    return KMC_OK;
}

// Streaming moments of the multi-launch kernels are sojourn-weighted (a walker's value is credited when it is replaced):
// credit every walker's CURRENT value with the samples it has stood for so far (enqueued on the sampler's stream; after
// it every klast equals the number of samples taken).  Before a read-out, and before walkers change slots (deal).
kmc_status flush_moments_now(kmc_sampler* s)
{
    if (!s->d_msum || s->islands || s->resident) return KMC_OK;
    HIP_TRY(launch_sweep(s));                                   // posted ring entries first, in their order
    if (s->plan.vec) {
        FlushFn fl = flush_lookup(s->plan.L, s->plan.K, s->plan.ITER, s->f32);
        if (!fl) return fail(KMC_ERR_UNSUPPORTED, "no flush kernel for this geometry");
        for (int half = 0; half < 2; ++half) {
            FlushArgs fa{};
            fa.pos = s->d_pos;
            fa.klast = s->d_klast;
            fa.msum = s->d_msum;
            fa.msumsq = s->d_msumsq;
            fa.macc_stride = s->macc_stride;
            fa.row0 = s->p2p ? (int64_t)half * s->h_loc : (int64_t)half * s->h + s->active_begin;
            fa.n_active = (int32_t)s->h_loc;
            fa.nsamp = (uint32_t)samples_done(s);
            fa.ld = (int32_t)s->ld;
            hipLaunchKernelGGL(fl, dim3(s->grid), dim3(s->tpb), 0, s->stream, fa);
            HIP_TRY(hipGetLastError());
        }
    }
    return KMC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------
KMC_EXPORT int kmc_version(void) { return KMC_VERSION; }

KMC_EXPORT int kmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

KMC_EXPORT const char* kmc_last_error(void) { return g_err.c_str(); }

// Diagnostics (KMC_ABORT_BACKTRACE=1 in the environment when the library is loaded): the native call stack of an abort()
// raised anywhere in the process (the HIP runtime aborts on internal errors without a message), on stderr.
#ifndef KMC_DIAG_ALWAYS
#define KMC_DIAG_ALWAYS 0          // -DKMC_DIAG_ALWAYS=1: a diagnostics build that always installs the handler (file /tmp/kmc_abort_bt.txt)
#endif
namespace {
void abort_backtrace(int sig)
{
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\n[kissmcmc_hip] SIGABRT, native stack:\n";
    int fd = 2;                                          // KMC_ABORT_BACKTRACE=/path/to/file: there (a test runner may have captured fd 2)
    const char* where = std::getenv("KMC_ABORT_BACKTRACE");
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
    AbortBacktraceInstaller() { if (std::getenv("KMC_ABORT_BACKTRACE") || KMC_DIAG_ALWAYS) signal(SIGABRT, abort_backtrace); }
} g_abort_backtrace_installer;
}  // namespace

KMC_EXPORT const char* kmc_status_string(kmc_status st)
{
    switch (st) {
    case KMC_OK: return "ok";
    case KMC_ERR_A_SCALE: return "a_scale must be > 1";
    case KMC_ERR_ODD_WALKERS: return "Use an even number of walkers.";
    case KMC_ERR_TOO_FEW_WALKERS: return "Use more walkers: at least DOF+2, but better many more.";
    case KMC_ERR_BAD_ARG: return "bad argument";
    case KMC_ERR_NONFINITE_LOGP: return "initial walker with non-finite log-pdf";
    case KMC_ERR_HIP: return "HIP runtime error";
    case KMC_ERR_OOM: return "out of device memory";
    case KMC_ERR_NO_DEVICE: return "no HIP device";
    case KMC_ERR_UNSUPPORTED: return "unsupported configuration";
    }
    return "unknown status";
}

// src/samplers.jl:200-205
KMC_EXPORT kmc_status kmc_validate(const kmc_config* c)
{
    if (!c) return fail(KMC_ERR_BAD_ARG, "null config");
    if (c->nwalkers <= 0 || c->ndim <= 0 || c->nthin <= 0 || c->ngenerations < 0 || c->nburnin < 0)
        return fail(KMC_ERR_BAD_ARG, "nwalkers, ndim, nthin must be > 0 and ngenerations, nburnin >= 0");
    if (!(c->a_scale > 1.0)) return fail(KMC_ERR_A_SCALE, kmc_status_string(KMC_ERR_A_SCALE));
    if (c->nwalkers % 2 != 0) return fail(KMC_ERR_ODD_WALKERS, kmc_status_string(KMC_ERR_ODD_WALKERS));
    if (c->nwalkers < c->ndim + 2) return fail(KMC_ERR_TOO_FEW_WALKERS, kmc_status_string(KMC_ERR_TOO_FEW_WALKERS));
    if (c->nwalkers >= (int64_t)1 << 31 || c->ndim >= (int64_t)1 << 24)
        return fail(KMC_ERR_UNSUPPORTED, "ensemble too large");
    if (c->ngenerations >= (int64_t)1 << 31) return fail(KMC_ERR_UNSUPPORTED, "at most 2^31 - 1 generations (the step index is 32 bits)");
    if (c->density == KMC_USER_DENSITY && !c->user_density) return fail(KMC_ERR_BAD_ARG, "KMC_USER_DENSITY needs kmc_config.user_density");
    if (c->density == KMC_HOST_DENSITY) {
        if (!c->host_logpdf) return fail(KMC_ERR_BAD_ARG, "KMC_HOST_DENSITY needs kmc_config.host_logpdf");
        if ((c->flags & (KMC_P2P | KMC_ISLANDS)) || c->shard_count > 1)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_HOST_DENSITY runs on one GPU, without KMC_P2P / KMC_ISLANDS / sharding");
    }
    if (c->dtype != KMC_F64 && c->dtype != KMC_F32) return fail(KMC_ERR_UNSUPPORTED, "dtype must be KMC_F64 or KMC_F32");
    if (c->dtype == KMC_F32) {
        if ((c->flags & (KMC_P2P | KMC_ISLANDS)) || c->shard_count > 1)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_F32 rows: one GPU, without KMC_P2P / KMC_ISLANDS / sharding");
        if (c->density == KMC_HOST_DENSITY)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_F32 rows: densities evaluated on the device only (built-in or runtime-compiled)");
    }
    if (c->host_accepted && c->density != KMC_HOST_DENSITY) return fail(KMC_ERR_BAD_ARG, "kmc_config.host_accepted needs KMC_HOST_DENSITY");
    if (c->density == KMC_ROSENBROCK && c->ndim < 2) return fail(KMC_ERR_BAD_ARG, "rosenbrock needs ndim >= 2");
    if (c->density == KMC_MVNORMAL2 && c->ndim != 2) return fail(KMC_ERR_BAD_ARG, "mvnormal2 needs ndim == 2");
    const int P = c->shard_count <= 0 ? 1 : c->shard_count;
    if (c->shard_rank < 0 || c->shard_rank >= P) return fail(KMC_ERR_BAD_ARG, "shard_rank out of range");
    if ((c->nwalkers / 2) % P != 0) return fail(KMC_ERR_BAD_ARG, "nwalkers/2 must be divisible by shard_count");
    if ((c->flags & KMC_P2P) && P > 8) return fail(KMC_ERR_UNSUPPORTED, "KMC_P2P supports at most 8 shards (one node)");
    if (c->flags & KMC_ISLANDS) {
        const int64_t S = c->island_size > 0 ? c->island_size : kIslandSizeDefault;
        if ((S != 64 && S != 128 && S != 256) || c->nwalkers % S != 0 || c->ndim > 32 || c->ndim + 2 > S || P != 1 ||
            (c->flags & (KMC_P2P | KMC_STORE_CHAIN | KMC_STORE_LOGP)) || c->island_gens < 0)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_ISLANDS needs island_size in {64,128,256} >= ndim+2 dividing nwalkers, ndim <= 32, one shard and no chain storage");
        if (c->density == KMC_USER_DENSITY && (size_t)S * (size_t)(2 * c->ndim + 9) * sizeof(double) > 60 * 1024)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_ISLANDS with a user density: island_size * (ndim + 3) * 8 must stay below 60 KiB (use island_size 128 or 64)");
    }
    if (c->flags & KMC_STREAM_CHAIN) {
        if (!(c->flags & (KMC_STORE_CHAIN | KMC_STORE_LOGP))) return fail(KMC_ERR_BAD_ARG, "KMC_STREAM_CHAIN needs KMC_STORE_CHAIN and / or KMC_STORE_LOGP");
        if (c->dtype != KMC_F64 || P != 1 || (c->flags & (KMC_P2P | KMC_ISLANDS)))
            return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN: KMC_F64, one GPU, without KMC_P2P / KMC_ISLANDS / sharding");
    }
    if (c->deal_count < 0 || (c->deal_count > 0 && (c->deal_rank < 0 || c->deal_rank >= c->deal_count)))
        return fail(KMC_ERR_BAD_ARG, "deal_rank / deal_count out of range");
    if (c->deal_count > 0) {
        // (a stored chain is by SLOT: which walker a slot held when a sample was taken follows from kmc_deal_perm, distributed.py)
        if (P != 1 || (c->flags & (KMC_P2P | KMC_ISLANDS | KMC_STREAM_CHAIN)) || c->dtype != KMC_F64 || c->density == KMC_HOST_DENSITY)
            return fail(KMC_ERR_UNSUPPORTED, "dealt sub-ensembles: KMC_F64, a device density, one shard, no KMC_P2P / KMC_ISLANDS / KMC_STREAM_CHAIN");
        if (c->nwalkers % c->deal_count != 0)
            return fail(KMC_ERR_BAD_ARG, "dealt sub-ensembles: nwalkers (this sub-ensemble's size) must be divisible by deal_count");
        if (c->nwalkers * (int64_t)c->deal_count >= (int64_t)1 << 32) return fail(KMC_ERR_UNSUPPORTED, "dealt sub-ensembles: at most 2^32 - 1 walkers in all");
    }
    DensityParams dp;
    return digest_params(*c, &dp);
}

KMC_EXPORT double kmc_g_pdf(double z, double a)   // src/samplers.jl:224
{
    return (1.0 / a <= z && z <= a) ? 1.0 / std::sqrt(z) * 1.0 / (2.0 * (std::sqrt(a) - std::sqrt(1.0 / a))) : 0.0;
}

KMC_EXPORT double kmc_cdf_g_inv(double u, double a)   // src/samplers.jl:227
{
    const double t = std::fma(u, std::sqrt(a) - std::sqrt(1.0 / a), std::sqrt(1.0 / a));
    return t * t;
}

// ------------------------------------------------------------------------------------------
// sampler
// ------------------------------------------------------------------------------------------
// hipMalloc does not clear memory.  KMC_POISON=1 (diagnostics): every allocation of a sampler starts as 0xFF bytes (NaN doubles,
// 4 294 967 295 counters), so that anything the code forgot to initialise shows up in the tests instead of depending on what the
// allocator happened to return.
// ... and is followed by a 4 KiB guard of 0xA5 that kmc_sampler_destroy checks: a kernel that writes past the end of one of
// its buffers aborts the process there, with the size of the allocation (the tests then fail loudly).
constexpr size_t kGuardBytes = 4096;
template <class T>
hipError_t dev_alloc(kmc_sampler* s, T** p, size_t bytes)
{
    static const bool poison = std::getenv("KMC_POISON") != nullptr;
    if (!poison || bytes == 0) return hipMalloc(reinterpret_cast<void**>(p), bytes);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes + kGuardBytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(*p, 0xFF, bytes, s->stream);
    if (e == hipSuccess) e = hipMemsetAsync(reinterpret_cast<char*>(*p) + bytes, 0xA5, kGuardBytes, s->stream);
    s->guards.emplace_back(reinterpret_cast<char*>(*p) + bytes, bytes);
    return e;
}
void check_guards(kmc_sampler* s)
{
    std::vector<unsigned char> h(kGuardBytes);
    for (const auto& g : s->guards) {
        if (copy_sync(h.data(), g.first, kGuardBytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (size_t i = 0; i < kGuardBytes; ++i)
            if (h[i] != 0xA5) {
                std::fprintf(stderr, "[kissmcmc_hip] KMC_POISON: byte %zu behind a device allocation of %zu bytes was overwritten (%s)\n", i, g.second,
                             s->plan.vec ? "vec kernels" : "generic / staged kernels");
                std::abort();
            }
    }
    s->guards.clear();
}

KMC_EXPORT kmc_status kmc_sampler_create(const kmc_config* cfg, kmc_sampler** out)
{
    if (!out) return fail(KMC_ERR_BAD_ARG, "null out");
    *out = nullptr;
    KMC_TRY(kmc_validate(cfg));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(KMC_ERR_NO_DEVICE, "no HIP device visible: the emcee hot path has no CPU fallback");
    }
    if (cfg->device < 0 || cfg->device >= ndev) return fail(KMC_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));

    kmc_sampler* s = new kmc_sampler();
    s->cfg = *cfg;
    s->user_seed = cfg->seed;
    if (cfg->deal_count > 0) s->cfg.seed = deal_seed(cfg->seed, cfg->deal_rank);    // this sub-ensemble's Philox key
    if (s->cfg.shard_count <= 0) s->cfg.shard_count = 1;
    s->h = cfg->nwalkers / 2;
    s->h_loc = s->h / s->cfg.shard_count;
    s->active_begin = s->h_loc * s->cfg.shard_rank;
    s->nlocal = 2 * s->h_loc;
    s->nsamples = cfg->ngenerations > cfg->nburnin ? (cfg->ngenerations - cfg->nburnin) / cfg->nthin : 0;   // :234
    s->ld = cfg->ndim + (cfg->ndim & 1);      // whole two-element chunks: 16-byte aligned double rows, 8-byte aligned float rows
    s->f32 = cfg->dtype == KMC_F32;
    kmc_status st = digest_params(*cfg, &s->dp);
    if (st != KMC_OK) { delete s; return st; }
    s->plan = make_plan(s->cfg, s->h_loc);
    if (cfg->density == KMC_USER_DENSITY) {
        s->user = static_cast<kmc_user_density*>(cfg->user_density);
        // small ensembles: resident mode too (one workgroup, LDS within the default 64 KiB limit)
        int rK = 0, rK0 = 1;
        while (2 * rK0 < s->ld / 2) rK0 *= 2;
        const size_t rlds = ((size_t)cfg->nwalkers * (size_t)(4 * (rK0 + 1)) + (size_t)cfg->nwalkers) * sizeof(double);
        if (!s->user->is_body && !s->f32 && cfg->nwalkers <= 256 && cfg->ndim <= 32 && s->cfg.shard_count == 1 && !(cfg->flags & (KMC_P2P | KMC_NO_GRAPH | KMC_ISLANDS | KMC_STREAM_CHAIN)) &&
            rlds <= 60 * 1024 && std::getenv("KMC_NO_RESIDENT") == nullptr)
            rK = rK0;
        int iS = 0;
        if (cfg->flags & KMC_ISLANDS) {
            iS = cfg->island_size > 0 ? cfg->island_size : kIslandSizeDefault;
            rK = 1;
            while (2 * rK < s->ld / 2) rK *= 2;
        }
        if (s->user->is_body && iS > 0) { kmc_sampler_destroy(s); return fail(KMC_ERR_UNSUPPORTED, "KMC_ISLANDS needs a menu or term / pair density (a body density runs one walker per lane)"); }
        if (s->user->is_body && cfg->ndim > 1024) { kmc_sampler_destroy(s); return fail(KMC_ERR_UNSUPPORTED, "a body density holds the proposal per lane: ndim <= 1024"); }
        st = load_user(s->user, s->plan.vec, s->plan.L, s->plan.K, s->plan.ITER, s->plan.ragged, &s->uk, rK, 4 * rK != cfg->ndim, iS, s->f32, cfg->ndim, (cfg->flags & KMC_P2P) != 0);
        if (st != KMC_OK) { kmc_sampler_destroy(s); return st; }
        if (iS > 0) rK = 0;     // island mode is set up below, not resident mode
        if (rK > 0) {
            s->resident = true;
            s->island_K = rK;
            s->nislands = 1;
            s->island_lds = rlds < 4096 ? 4096 : rlds;
        }
    } else if (cfg->density == KMC_HOST_DENSITY) {
        s->host_eval = true;
    } else {
        HalfStepFn v, g;
        lookup(cfg->density, 0, 0, 1, false, false, false, &v, &g, &s->logpdf_fn);
    }
    // vec: a wave owns W = (64/L)*ITER walkers; generic: one walker per lane
    const int64_t per_wave = s->plan.vec ? (int64_t)(64 / s->plan.L) * s->plan.ITER : 64;
    const int64_t waves = (s->h_loc + per_wave - 1) / per_wave;
    if (cfg->flags & KMC_ISLANDS) {
        s->islands = true;
        s->island_gens = cfg->island_gens > 0 ? cfg->island_gens : 32;
        s->island_size = cfg->island_size > 0 ? cfg->island_size : kIslandSizeDefault;
        s->nislands = cfg->nwalkers / s->island_size;
        const int64_t chunks = s->ld / 2;                   // 16-byte chunks per row, 2 lanes per walker
        int K = 1;
        while (2 * K < chunks) K *= 2;
        s->island_K = K;
        s->island_ragged = 4 * K != cfg->ndim;
        s->island_lds = ((size_t)s->island_size * (size_t)(4 * (K + 1)) + (size_t)s->island_size) * sizeof(double);
        if (s->island_lds < 4096) s->island_lds = 4096;          // the moment reduction reuses the buffer
        hipError_t ea = hipSuccess;
        if (!s->user) {
            s->island_kernel = island_fn(cfg->density, (int)s->island_size, K, s->island_ragged);
            if (!s->island_kernel) { kmc_sampler_destroy(s); return fail(KMC_ERR_UNSUPPORTED, "no island kernel for this density / ndim"); }
            ea = hipFuncSetAttribute(reinterpret_cast<const void*>(s->island_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->island_lds);
        }
        if (ea != hipSuccess) { (void)hipGetLastError(); kmc_sampler_destroy(s); return fail(KMC_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(ea)); }
    }
    if (!s->islands && !s->f32 && cfg->density != KMC_USER_DENSITY && !s->host_eval && cfg->nwalkers <= 1024 && cfg->ndim <= 32 &&
        s->cfg.shard_count == 1 && !(cfg->flags & (KMC_P2P | KMC_NO_GRAPH | KMC_STREAM_CHAIN)) && std::getenv("KMC_NO_RESIDENT") == nullptr) {
        const int64_t chunks = s->ld / 2;
        int K = 1;
        while (2 * K < chunks) K *= 2;
        const int rtpb = cfg->nwalkers <= 256 ? 256 : (cfg->nwalkers <= 512 ? 512 : 1024);
        const size_t need = ((size_t)cfg->nwalkers * (size_t)(4 * (K + 1)) + (size_t)cfg->nwalkers) * sizeof(double);
        ResidentFn rf = need <= 156 * 1024 ? resident_fn(cfg->density, rtpb, K, 4 * K != cfg->ndim) : nullptr;
        if (rf) {
            s->resident_tpb = rtpb;
            s->island_lds = need;
            if (s->island_lds < 8192) s->island_lds = 8192;     // the moment reduction reuses the buffer
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(rf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)s->island_lds) == hipSuccess) {
                s->resident = true;
                s->resident_kernel = rf;
                s->island_K = K;
                s->nislands = 1;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    // vec kernels: vec_tpb(L) threads per workgroup; the generic kernel keeps 256
    const bool staged = s->user && s->uk.staged != nullptr;         // a body density's staged kernel: two waves per workgroup
    const int tpb = s->plan.vec ? vec_tpb(s->plan.L) : (staged ? kStagedTPB : 256);
    s->tpb = tpb;
    s->grid = (int)((waves * 64 + tpb - 1) / tpb);
    s->macc_stride = (int64_t)s->grid * tpb;
    s->macc_elems = s->plan.vec ? s->macc_stride * 2 * s->plan.K : s->macc_stride * cfg->ndim;

#define CREATE_TRY(expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            kmc_status r_ = fail(e_ == hipErrorOutOfMemory ? KMC_ERR_OOM : KMC_ERR_HIP,        \
                                 std::string(#expr) + ": " + hipGetErrorString(e_));           \
            kmc_sampler_destroy(s);                                                            \
            return r_;                                                                         \
        }                                                                                      \
    } while (0)

    CREATE_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->own_stream = true;
    CREATE_TRY(hipEventCreate(&s->ev0));
    CREATE_TRY(hipEventCreate(&s->ev1));
    s->p2p = (cfg->flags & KMC_P2P) != 0;
    s->nrows = s->p2p ? s->nlocal : cfg->nwalkers;
    const size_t nw = (size_t)s->nrows;
    if (s->p2p) {
        CREATE_TRY(hipExtMallocWithFlags((void**)&s->d_flags, 4096, hipDeviceMallocFinegrained));
        CREATE_TRY(hipMemsetAsync(s->d_flags, 0, 4096, s->stream));
        CREATE_TRY(dev_alloc(s, (void**)&s->d_err, 64));
        CREATE_TRY(hipMemsetAsync(s->d_err, 0, 64, s->stream));
        CREATE_TRY(dev_alloc(s, (void**)&s->d_done, 33 * 64));
        CREATE_TRY(hipMemsetAsync(s->d_done, 0, 33 * 64, s->stream));
        // the kernel can publish its own completion only where all its stores are write-through: the vector kernels
        s->fold_signal = (cfg->flags & KMC_P2P_FOLD_SIGNAL) != 0 && s->plan.vec && s->user == nullptr;
        s->push = (cfg->flags & (KMC_P2P_PUSH | KMC_P2P_LAZY)) != 0 && s->plan.vec && s->user == nullptr && !(cfg->flags & KMC_P2P_FINEGRAINED) &&
                  s->cfg.shard_count > 1;
        s->lazy = s->push && (cfg->flags & KMC_P2P_LAZY) != 0 && s->h_loc % 16 == 0 && !s->f32;
        if (const char* e = std::getenv("KMC_P2P_STATS")) s->lazy_stats = s->lazy && e[0] == '1';
    }
    // KMC_P2P_LAZY: room for every rank's accept-byte maps behind the row blocks (peers write them: same allocation)
    const size_t amap_bytes = s->lazy ? (size_t)s->cfg.shard_count * 4 * (size_t)s->h_loc : 0;
    const size_t ldz = (size_t)s->ld;
    const size_t esz = s->f32 ? sizeof(float) : sizeof(double);      // element size of rows and chain
    if (s->p2p && (cfg->flags & KMC_P2P_FINEGRAINED))   // peers map the rows uncached: nothing of them can go stale in a reader's L2
        CREATE_TRY(hipExtMallocWithFlags((void**)&s->d_pos, nw * ldz * sizeof(double), hipDeviceMallocFinegrained));
    else
        CREATE_TRY(dev_alloc(s, &s->d_pos, (s->push ? 1 + (size_t)s->cfg.shard_count : 1) * nw * ldz * esz + amap_bytes));
    CREATE_TRY(hipMemsetAsync(s->d_pos, 0, (s->push ? 1 + (size_t)s->cfg.shard_count : 1) * nw * ldz * esz + amap_bytes, s->stream));   // the pad column of odd ndim stays 0
    if (s->lazy) {
        const size_t P = (size_t)s->cfg.shard_count, hl = (size_t)s->h_loc;
        const size_t nb = 2 * P * 2 * hl * sizeof(uint32_t) + 16;
        CREATE_TRY(dev_alloc(s, (void**)&s->d_lazy, nb));
        CREATE_TRY(hipMemsetAsync(s->d_lazy, 0, nb, s->stream));
        s->peer_amap_in[s->cfg.shard_rank] = reinterpret_cast<unsigned char*>(s->d_pos) + (1 + P) * nw * ldz * esz;
    }
    // per-walker block {logp[nrows], naccept[nrows], klast[nrows]}: one allocation, so the half-step kernels reach all
    // three from one preloaded pointer (HalfStepFront::logp)
    CREATE_TRY(dev_alloc(s, &s->d_logp, nw * (sizeof(double) + 2 * sizeof(uint32_t))));
    s->d_naccept = reinterpret_cast<uint32_t*>(s->d_logp + nw);
    s->d_klast = s->d_naccept + nw;
    CREATE_TRY(hipMemsetAsync(s->d_klast, 0, nw * sizeof(uint32_t), s->stream));
    static_assert(kGraphChunk <= 64, "advance_schedule runs one 64-thread block");
    CREATE_TRY(dev_alloc(s, &s->d_gen, 64));
    CREATE_TRY(hipMemsetAsync(s->d_gen, 0, 64, s->stream));
    CREATE_TRY(dev_alloc(s, &s->d_sched, (size_t)kGraphChunk * sizeof(SchedEntry)));
    CREATE_TRY(hipMemsetAsync(s->d_naccept, 0, nw * sizeof(uint32_t), s->stream));
    if (cfg->deal_count > 0) CREATE_TRY(dev_alloc(s, (void**)&s->d_ids, nw * sizeof(uint32_t)));
    if (s->plan.vec && !s->islands && !s->resident && s->plan.L >= 16 && s->plan.L <= 32 && s->plan.L / s->plan.ITER >= 2 &&
        std::getenv("KMC_NO_DRAW_RING") == nullptr) {
        // draw ring: 4 slots x rows x 32 B; tags start at 0xffffffff (no step carries it), so nothing is "parked" yet
        const size_t nb = 4 * (size_t)s->nrows * 2 * sizeof(double2);
        CREATE_TRY(dev_alloc(s, (void**)&s->d_ring, nb));
        CREATE_TRY(hipMemsetAsync(s->d_ring, 0xff, nb, s->stream));
    }
    if (cfg->flags & KMC_MOMENTS) {
        CREATE_TRY(dev_alloc(s, &s->d_msum, (size_t)s->macc_elems * sizeof(double)));
        CREATE_TRY(dev_alloc(s, &s->d_msumsq, (size_t)s->macc_elems * sizeof(double)));
        CREATE_TRY(hipMemsetAsync(s->d_msum, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        CREATE_TRY(hipMemsetAsync(s->d_msumsq, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        if (s->plan.vec && s->plan.L == 64 && !s->islands && !s->resident && std::getenv("KMC_NO_MOMENT_RING") == nullptr) {
            // moment ring for long rows (kmc_kernels.hpp, HalfStepArgs::mring): up to 128 posted rows per wave,
            // within 512 MiB in all; swept every kSweepEvery generations
            const int64_t nwaves = s->macc_stride / 64;
            const size_t slot = (size_t)s->plan.K * 64 * sizeof(double2);       // one row
            int64_t depth = (int64_t)(((size_t)1 << 29) / ((size_t)nwaves * slot));
            if (depth > 128) depth = 128;
            if (const char* e = std::getenv("KMC_MOMENT_RING_DEPTH")) { const long v = std::atol(e); if (v >= 1 && v < depth) depth = v; }   // tests: force overflows
            if (depth >= 4 || (depth >= 2 && std::getenv("KMC_MOMENT_RING_DEPTH") != nullptr)) {
                CREATE_TRY(dev_alloc(s, (void**)&s->d_mring, (size_t)nwaves * (size_t)depth * slot));
                CREATE_TRY(dev_alloc(s, (void**)&s->d_mring_w, (size_t)nwaves * (size_t)depth * sizeof(double)));
                CREATE_TRY(dev_alloc(s, (void**)&s->d_mcnt, 2 * (size_t)nwaves * sizeof(uint32_t)));
                CREATE_TRY(hipMemsetAsync(s->d_mcnt, 0, 2 * (size_t)nwaves * sizeof(uint32_t), s->stream));
                s->mring_depth = (int)depth;
                s->mring_waves = nwaves;
            }
        }
        if (s->islands || s->resident) {
            const size_t ne = (size_t)s->nislands * 4 * (size_t)s->island_K;
            CREATE_TRY(dev_alloc(s, &s->d_isum, ne * sizeof(double)));
            CREATE_TRY(dev_alloc(s, &s->d_isumsq, ne * sizeof(double)));
            CREATE_TRY(hipMemsetAsync(s->d_isum, 0, ne * sizeof(double), s->stream));
            CREATE_TRY(hipMemsetAsync(s->d_isumsq, 0, ne * sizeof(double), s->stream));
        }
    }
    if (s->host_eval) {
        CREATE_TRY(dev_alloc(s, &s->d_prop, (size_t)s->h * ldz * sizeof(double)));
        CREATE_TRY(dev_alloc(s, &s->d_p1, (size_t)s->h * sizeof(double)));
        CREATE_TRY(hipHostMalloc((void**)&s->h_prop, (size_t)s->h * (size_t)cfg->ndim * sizeof(double), hipHostMallocDefault));
        CREATE_TRY(hipHostMalloc((void**)&s->h_p1, (size_t)s->h * sizeof(double), hipHostMallocDefault));
        if (cfg->host_accepted) {
            CREATE_TRY(dev_alloc(s, &s->d_acc, (size_t)s->h));
            CREATE_TRY(hipHostMalloc((void**)&s->h_acc, (size_t)s->h, hipHostMallocDefault));
        }
    }
    int64_t chain_slots = s->nsamples;
    if ((cfg->flags & KMC_STREAM_CHAIN) && s->nsamples > 0) {
        // a ring of three blocks; a block holds at least the samples of one launch unit (a graph replay), so a unit never
        // touches more than two blocks, and about 512 MiB otherwise (measured at C2: 128 MiB blocks stream 22-26 GB/s at nthin = 10, 512 MiB blocks 43-45 GB/s of the
        // 56 GB/s this link copies alone; many small blocks cost more at the block boundaries than their earlier start
        // returns -- nthin = 100: +19 % on the loop with 64 MiB blocks, +6 % with 512 MiB; KMC_CHAIN_BLOCK = samples per block, for tests)
        if (const char* e = std::getenv("KMC_UPD_CHUNK")) { const long v = std::atol(e); if (v >= 16 && v <= 1024) s->uchunk = v; }
        const int64_t unit = std::max<int64_t>(kGraphChunk, s->uchunk);
        const int64_t per_unit = (unit + cfg->nthin - 1) / cfg->nthin + 1;
        const size_t sample_bytes = (size_t)s->nlocal * ldz * sizeof(double);
        int64_t blk = (int64_t)(((size_t)512 << 20) / sample_bytes);
        if (blk < 1) blk = 1;
        if (blk > 4096) blk = 4096;
        if (const char* e = std::getenv("KMC_CHAIN_BLOCK")) { const long v = std::atol(e); if (v >= 1) blk = v; }
        if (blk < per_unit) blk = per_unit;
        s->stream_chain = true;
        s->stream_by_walker = (cfg->flags & KMC_CHAIN_BY_WALKER) != 0;
        const char* bwc = std::getenv("KMC_BYWALKER_COPY");
        if (s->stream_by_walker && !(bwc && std::strcmp(bwc, "kernel") == 0)) {
            if (cfg->flags & KMC_STORE_CHAIN) CREATE_TRY(dev_alloc(s, &s->bw_scratch, (size_t)blk * (size_t)s->nlocal * (size_t)cfg->ndim * sizeof(double)));
            if (cfg->flags & KMC_STORE_LOGP) CREATE_TRY(dev_alloc(s, &s->bw_scratch_logp, (size_t)blk * (size_t)s->nlocal * sizeof(double)));
        } else if (!s->stream_by_walker && (cfg->flags & KMC_STORE_CHAIN) && s->ld != cfg->ndim) {
            CREATE_TRY(dev_alloc(s, &s->bw_scratch, (size_t)blk * (size_t)s->nlocal * (size_t)cfg->ndim * sizeof(double)));   // (rows_compact)
        }
        s->ring_blk = blk;
        s->ring_slots = 3 * blk;
        chain_slots = s->ring_slots;
        CREATE_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 3; ++i) {
            CREATE_TRY(hipEventCreateWithFlags(&s->ev_filled[i], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&s->ev_copied[i], hipEventDisableTiming));
        }
    }
    if ((cfg->flags & (KMC_STORE_CHAIN | KMC_STORE_LOGP)) && s->nsamples > 0) {
        // A chain that cannot fit is refused HERE, not by a failing hipMalloc: after a failed allocation of hundreds of GB the
        // runtime aborted the process a few calls later (observed: 4 of 5 runs, in the first HIP call of the next sampler).
        size_t free_b = 0, total_b = 0;
        CREATE_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t need = (size_t)chain_slots * (size_t)s->nlocal *
                            (((cfg->flags & KMC_STORE_CHAIN) ? ldz * esz : 0) + ((cfg->flags & KMC_STORE_LOGP) ? sizeof(double) : 0));
        if (need > free_b) {
            kmc_status r_ = fail(KMC_ERR_OOM, "the chain needs " + std::to_string(need >> 20) + " MiB of device memory, " + std::to_string(free_b >> 20) +
                                              " MiB are free: thin it (nthin), or stream it to host memory (KMC_STREAM_CHAIN)");
            kmc_sampler_destroy(s);
            return r_;
        }
    }
    if ((cfg->flags & KMC_STORE_CHAIN) && s->nsamples > 0)
        CREATE_TRY(dev_alloc(s, &s->d_chain, (size_t)chain_slots * (size_t)s->nlocal * ldz * esz));
    if ((cfg->flags & KMC_STORE_LOGP) && s->nsamples > 0)
        CREATE_TRY(dev_alloc(s, &s->d_chain_logp, (size_t)chain_slots * (size_t)s->nlocal * sizeof(double)));
    CREATE_TRY(hipStreamSynchronize(s->stream));         // the fills above (asynchronous, one wait for all of them)
#undef CREATE_TRY
    if (s->p2p) {
        s->peer_pos[s->cfg.shard_rank] = s->d_pos;
        s->peer_flags[s->cfg.shard_rank] = s->d_flags;
        s->connected = s->cfg.shard_count == 1;
    }
    *out = s;
    return KMC_OK;
}

KMC_EXPORT void kmc_sampler_destroy(kmc_sampler* s)
{
    if (!s) return;
    (void)hipSetDevice(s->cfg.device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    if (!s->guards.empty() && s->stream) check_guards(s);
    for (int i = 0; i < kUExec; ++i) {
        if (s->uexec[i]) (void)hipGraphExecDestroy(s->uexec[i]);
        if (s->udone[i]) (void)hipEventDestroy(s->udone[i]);
    }
    if (s->ugraph) (void)hipGraphDestroy(s->ugraph);
    if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
    if (s->uk.mod) (void)hipModuleUnload(s->uk.mod);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->p2p) {
        for (int r = 0; r < 8; ++r) {
            if (r == s->cfg.shard_rank) continue;
            if (s->peer_pos[r]) (void)hipIpcCloseMemHandle(s->peer_pos[r]);
            if (s->peer_flags[r]) (void)hipIpcCloseMemHandle(s->peer_flags[r]);
        }
        (void)hipFree(s->d_flags);
        (void)hipFree(s->d_err);
        (void)hipFree(s->d_done);
    }
    if (s->comm) { rccl_comm_destroy(s->comm); s->comm = nullptr; }
    if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
    for (int i = 0; i < 3; ++i) {
        if (s->ev_filled[i]) (void)hipEventDestroy(s->ev_filled[i]);
        if (s->ev_copied[i]) (void)hipEventDestroy(s->ev_copied[i]);
    }
    chain_unregister(s);
    if (s->own_pos) (void)hipFree(s->d_pos);
    (void)hipFree(s->d_logp);          // the {logp, naccept, klast} block
    (void)hipFree(s->d_mring);
    (void)hipFree(s->d_ids);
    (void)hipFree(s->d_lazy);
    (void)hipFree(s->d_mring_w);
    (void)hipFree(s->d_mcnt);
    (void)hipFree(s->d_gen);
    (void)hipFree(s->d_sched);
    (void)hipFree(s->d_chain);
    (void)hipFree(s->bw_scratch);
    (void)hipFree(s->bw_scratch_logp);
    (void)hipFree(s->d_chain_logp);
    (void)hipFree(s->d_msum);
    (void)hipFree(s->d_msumsq);
    (void)hipFree(s->d_ring);
    (void)hipFree(s->d_isum);
    (void)hipFree(s->d_isumsq);
    (void)hipFree(s->d_prop);
    (void)hipFree(s->d_p1);
    if (s->h_prop) (void)hipHostFree(s->h_prop);
    if (s->h_p1) (void)hipHostFree(s->h_p1);
    (void)hipFree(s->d_acc);
    if (s->h_acc) (void)hipHostFree(s->h_acc);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    (void)hipGetLastError();
    delete s;
}

KMC_EXPORT kmc_status kmc_sampler_set_stream(kmc_sampler* s, void* hip_stream)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_bind_positions(kmc_sampler* s, void* pos_dev)
{
    if (!s || !pos_dev) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (s->p2p) return fail(KMC_ERR_UNSUPPORTED, "KMC_P2P samplers export their own position buffer");
    if (s->f32) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_bind_positions takes double rows; a KMC_F32 sampler keeps its own float rows");
    if (s->ld != s->cfg.ndim) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_bind_positions needs an even ndim (16-byte rows)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
    if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
    for (int i = 0; i < kUExec; ++i) {
        if (s->uexec[i]) { (void)hipGraphExecDestroy(s->uexec[i]); s->uexec[i] = nullptr; }
        s->uinflight[i] = false;
    }
    if (s->ugraph) { (void)hipGraphDestroy(s->ugraph); s->ugraph = nullptr; }
    if (s->own_pos) {
        for (size_t i = 0; i < s->guards.size(); ++i)                   // (KMC_POISON: this allocation's guard goes with it)
            if (s->guards[i].first - s->guards[i].second == reinterpret_cast<char*>(s->d_pos)) { s->guards.erase(s->guards.begin() + (long)i); break; }
        (void)hipFree(s->d_pos);
    }
    s->d_pos = static_cast<double*>(pos_dev);
    s->own_pos = false;
    s->positions_set = false;
    return KMC_OK;
}

// ---- replica sharding with a native RCCL all-gather ----------------------------------------
KMC_EXPORT kmc_status kmc_rccl_unique_id(void* id_out)
{
    if (!id_out) return fail(KMC_ERR_BAD_ARG, "null argument");
    return rccl_unique_id(id_out);
}

KMC_EXPORT kmc_status kmc_sampler_rccl_init(kmc_sampler* s, const void* id)
{
    if (!s || !id) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (s->p2p || s->islands || s->resident || s->host_eval || s->f32 || s->d_ids || s->stream_chain)
        return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_rccl_init: a replica-sharded double sampler with a device density (shard_rank / shard_count, no KMC_P2P)");
    if (s->comm) return KMC_OK;
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return rccl_comm_create(id, s->cfg.shard_rank, s->cfg.shard_count, &s->comm);
}

// ---- peer-to-peer sharding ---------------------------------------------------------------
struct P2PHandle { hipIpcMemHandle_t pos, flags; };
static_assert(sizeof(P2PHandle) == KMC_P2P_HANDLE_BYTES, "handle blob size");

KMC_EXPORT kmc_status kmc_sampler_p2p_export(kmc_sampler* s, void* handle_out)
{
    if (!s || !handle_out) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->p2p) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_P2P");
    HIP_TRY(hipSetDevice(s->cfg.device));
    P2PHandle h;
    HIP_TRY(hipIpcGetMemHandle(&h.pos, s->d_pos));
    HIP_TRY(hipIpcGetMemHandle(&h.flags, s->d_flags));
    std::memcpy(handle_out, &h, sizeof(h));
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_p2p_connect(kmc_sampler* s, const void* handles)
{
    if (!s || !handles) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->p2p) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_P2P");
    if (s->connected) return KMC_OK;
    HIP_TRY(hipSetDevice(s->cfg.device));
    const P2PHandle* h = static_cast<const P2PHandle*>(handles);
    for (int r = 0; r < s->cfg.shard_count; ++r) {
        if (r == s->cfg.shard_rank) continue;
        void* p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h[r].pos, hipIpcMemLazyEnablePeerAccess));
        s->peer_pos[r] = static_cast<double*>(p);
        if (s->lazy)
            s->peer_amap_in[r] = static_cast<unsigned char*>(p) + (size_t)(1 + s->cfg.shard_count) * (size_t)s->nrows * (size_t)s->ld * sizeof(double);
        HIP_TRY(hipIpcOpenMemHandle(&p, h[r].flags, hipIpcMemLazyEnablePeerAccess));
        s->peer_flags[r] = static_cast<unsigned long long*>(p);
    }
    s->connected = true;
    return KMC_OK;
}

// All the shards in ONE process (and, here, on one device): wire the peers' buffers directly, no IPC.  The shards
// then run concurrently on their own streams exactly like ranks on separate GPUs -- for single-process tests,
// timing and profiling of the exchange variants.
KMC_EXPORT kmc_status kmc_sampler_p2p_connect_local(kmc_sampler* s, kmc_sampler* const* shards)
{
    if (!s || !shards) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->p2p) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_P2P");
    for (int r = 0; r < s->cfg.shard_count; ++r) {
        const kmc_sampler* o = shards[r];
        if (!o || !o->p2p || o->cfg.shard_count != s->cfg.shard_count || o->cfg.shard_rank != r || o->cfg.device != s->cfg.device ||
            o->nrows != s->nrows || o->ld != s->ld || o->lazy != s->lazy || o->push != s->push)
            return fail(KMC_ERR_BAD_ARG, "kmc_sampler_p2p_connect_local: shards[r] must be shard r of the same configuration on the same device");
        s->peer_pos[r] = o->d_pos;
        s->peer_flags[r] = o->d_flags;
        s->peer_amap_in[r] = o->peer_amap_in[r];
    }
    s->connected = true;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_p2p_stats(kmc_sampler* s, uint64_t out[2])
{
    if (!s || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    out[0] = out[1] = 0;
    if (!s->lazy) return KMC_OK;
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t hl = (size_t)s->h_loc, P = (size_t)s->cfg.shard_count;
    unsigned long long v[2] = {0ull, 0ull};
    HIP_TRY(copy_sync(v, s->d_lazy + 2 * P * 2 * hl * sizeof(uint32_t), sizeof(v), hipMemcpyDeviceToHost, s->stream));
    out[0] = v[0]; out[1] = v[1];
    return KMC_OK;
}

namespace {

// Initial log-pdfs of the rows in d_pos (src/samplers.jl:209-210) into d_logp, on the sampler's stream.  KMC_F32: the
// log-pdf kernels read double rows, so the float rows are widened (exactly) into a scratch buffer first.
__global__ __launch_bounds__(256) void widen_rows(const float* src, double* dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}
__global__ __launch_bounds__(256) void narrow_rows(const double* src, float* dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

kmc_status eval_initial_logp(kmc_sampler* s)
{
    const size_t nw = (size_t)s->nrows, nd = (size_t)s->cfg.ndim;
    double* rows = s->d_pos;
    double* scratch = nullptr;
    if (s->f32) {
        const int64_t n = (int64_t)(nw * (size_t)s->ld);
        HIP_TRY(hipMalloc((void**)&scratch, (size_t)n * sizeof(double)));
        hipLaunchKernelGGL(widen_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, reinterpret_cast<const float*>(s->d_pos), scratch, n);
        rows = scratch;
    }
    const LogpdfArgs la{rows, s->d_logp, (int64_t)nw, (int32_t)nd, (int32_t)s->ld, s->dp};
    hipError_t e = hipSuccess;
    if (s->user) {
        e = launch_module(s->uk.logpdf, (unsigned)((nw + 255) / 256), 256u, s->stream, la);
    } else {
        hipLaunchKernelGGL(s->logpdf_fn, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s->stream, la);
        e = hipGetLastError();
    }
    if (scratch) {
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        (void)hipFree(scratch);
    }
    HIP_TRY(e);
    return KMC_OK;
}

// Everything set_positions does after the rows are in place: initial log-pdfs (src/samplers.jl:209-210)
// unless they are supplied, counters, accumulators, finiteness check.
kmc_status reset_run_state(kmc_sampler* s, bool eval_logp, int64_t generation, uint32_t klast_value)
{
    const size_t nw = (size_t)s->nrows;
    if (eval_logp) KMC_TRY(eval_initial_logp(s));
    std::vector<double> lp(nw);
    HIP_TRY(copy_sync(lp.data(), s->d_logp, nw * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (s->d_msum) {
        HIP_TRY(hipMemsetAsync(s->d_msum, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        HIP_TRY(hipMemsetAsync(s->d_msumsq, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        if (s->d_klast) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)s->d_klast, (int)klast_value, nw, s->stream));
        if (s->d_mcnt) HIP_TRY(hipMemsetAsync(s->d_mcnt, 0, 2 * (size_t)s->mring_waves * sizeof(uint32_t), s->stream));
        if (s->d_isum) {
            const size_t ne = (size_t)s->nislands * 4 * (size_t)s->island_K;
            HIP_TRY(hipMemsetAsync(s->d_isum, 0, ne * sizeof(double), s->stream));
            HIP_TRY(hipMemsetAsync(s->d_isumsq, 0, ne * sizeof(double), s->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->generation = generation;
    s->launches = 0;
    s->have_run_events = false;
    if (s->stream_chain) { HIP_TRY(hipStreamSynchronize(s->copy_stream)); s->blocks_copied = 0; s->blocks_waited = 0; s->flushed_done = -1; }
    for (size_t w = 0; w < nw; ++w)
        if (!std::isfinite(lp[w])) {
            s->positions_set = false;
            return fail(KMC_ERR_NONFINITE_LOGP, "walker " + std::to_string(w) + " has a non-finite initial log-pdf");
        }
    s->positions_set = true;
    return KMC_OK;
}

}  // namespace

// Device-side make_theta0s: src/samplers.jl:311-349 (see init_ball in kmc_kernels.hpp).
KMC_EXPORT kmc_status kmc_sampler_init_ball(kmc_sampler* s, const double* theta0, const double* ball_radius,
                                            uint64_t seed, int halving_steps, int ntries)
{
    if (!s || !theta0 || !ball_radius || halving_steps < 1 || ntries < 1) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (s->host_eval) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_init_ball evaluates the density on the device; with KMC_HOST_DENSITY build the ball on the host");
    if (s->push) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_init_ball fills this rank's rows only; with KMC_P2P_PUSH use kmc_sampler_set_positions (the peers' copies must be filled too)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t nd = (size_t)s->cfg.ndim;
    double* d_par = nullptr;
    unsigned long long* d_fail = nullptr;
    HIP_TRY(hipMalloc((void**)&d_par, 2 * nd * sizeof(double)));
    hipError_t e = hipMalloc((void**)&d_fail, sizeof(unsigned long long));
    if (e == hipSuccess) e = copy_sync(d_par, theta0, nd * sizeof(double), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = copy_sync(d_par + nd, ball_radius, nd * sizeof(double), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = fill_sync(d_fail, 0, sizeof(unsigned long long), s->stream);
    const size_t nelem = (size_t)s->nrows * (size_t)s->ld;
    double* d_ball = s->d_pos;                  // KMC_F32: the ball is drawn in double, then rounded into the float rows
    if (s->f32 && e == hipSuccess) e = hipMalloc((void**)&d_ball, nelem * sizeof(double));
    if (e == hipSuccess) e = fill_sync(d_ball, 0, nelem * sizeof(double), s->stream);
    InitBallFn fn = s->user ? nullptr : init_ball_fn(s->cfg.density);
    const int pieces = s->p2p ? 2 : 1;
    for (int piece = 0; piece < pieces && e == hipSuccess; ++piece) {
        InitBallArgs a{};
        const int64_t rows = s->p2p ? s->h_loc : s->nrows;
        a.pos = d_ball + (size_t)piece * (size_t)s->h_loc * (size_t)s->ld;
        a.logp = s->d_logp + (size_t)piece * (size_t)s->h_loc;
        a.theta0 = d_par; a.radius = d_par + nd;
        a.nrows = rows;
        a.row_walker0 = s->p2p ? (int64_t)piece * s->h + s->active_begin
                               : (s->cfg.deal_count > 0 ? (int64_t)s->cfg.deal_rank * s->nrows : 0);   // rows of ONE global ball
        a.ndim = (int32_t)nd; a.ld = (int32_t)s->ld;
        a.halving_steps = halving_steps; a.ntries = ntries;
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        a.dp = s->dp;
        a.fail = d_fail;
        const unsigned grid = (unsigned)((rows + 255) / 256);
        if (s->user) e = launch_module(s->uk.init_ball, grid, 256u, s->stream, a);
        else { hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, s->stream, a); e = hipGetLastError(); }
    }
    unsigned long long nfail = 0;
    if (s->f32 && e == hipSuccess) {
        hipLaunchKernelGGL(narrow_rows, dim3((unsigned)((nelem + 255) / 256)), dim3(256), 0, s->stream, d_ball, reinterpret_cast<float*>(s->d_pos), (int64_t)nelem);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = copy_sync(&nfail, d_fail, sizeof(nfail), hipMemcpyDeviceToHost, s->stream);
    (void)hipFree(d_par);
    (void)hipFree(d_fail);
    if (s->f32) (void)hipFree(d_ball);
    HIP_TRY(e);
    if (nfail != 0) {
        s->positions_set = false;
        return fail(KMC_ERR_NONFINITE_LOGP, "Could not find suitable initial theta.  PDF is zero in too many places inside ball. (" +
                                            std::to_string(nfail) + " walkers)");
    }
    HIP_TRY(hipMemsetAsync(s->d_naccept, 0, (size_t)s->nrows * sizeof(uint32_t), s->stream));
    HIP_TRY(hipMemsetAsync(s->d_gen, 0, 64, s->stream));
    if (s->p2p) {
        HIP_TRY(fill_sync(s->d_flags, 0, 4096, s->stream));
        HIP_TRY(fill_sync(s->d_err, 0, 64, s->stream));
        if (s->d_done) HIP_TRY(fill_sync(s->d_done, 0, 33 * 64, s->stream));
    }
    s->dev_gen = 0;
    s->moment_base = 0;
    if (s->d_ids) {
        hipLaunchKernelGGL(deal_init_ids, dim3((unsigned)((s->nrows + 255) / 256)), dim3(256), 0, s->stream, s->d_ids, s->nrows,
                           (uint32_t)((uint64_t)s->cfg.deal_rank * (uint64_t)s->nrows));
        HIP_TRY(hipGetLastError());
    }
    return reset_run_state(s, /*eval_logp=*/s->f32, 0, 0u);      // KMC_F32: the log-pdfs of the rows as rounded
}

// Checkpoint / resume: restore (positions, log-pdfs, acceptance counters, generation).  The random
// stream is a pure function of (seed, generation, walker), so the continued run is bit-identical to an
// uninterrupted one.  Moments and the chain restart at the restored generation.
KMC_EXPORT kmc_status kmc_sampler_set_state(kmc_sampler* s, const double* pos_host, const double* logp_host,
                                            const int64_t* naccept_host, int64_t generation)
{
    if (!s || !pos_host || !logp_host || generation < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (s->d_chain || s->d_chain_logp) return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_set_state: not with chain storage (download the chain before checkpointing)");
    if (s->p2p && s->cfg.shard_count > 1)
        return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_set_state: P2P progress flags restart at 0; restore is single-GPU for now");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t nw = (size_t)s->nrows;
    HIP_TRY(upload_rows(s, s->d_pos, pos_host, nw));
    HIP_TRY(copy_sync(s->d_logp, logp_host, nw * sizeof(double), hipMemcpyHostToDevice, s->stream));
    std::vector<uint32_t> na(nw, 0u);
    if (naccept_host) for (size_t i = 0; i < nw; ++i) na[i] = (uint32_t)naccept_host[i];
    HIP_TRY(copy_sync(s->d_naccept, na.data(), nw * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    if (s->p2p) {
        HIP_TRY(fill_sync(s->d_flags, 0, 4096, s->stream));
        HIP_TRY(fill_sync(s->d_err, 0, 64, s->stream));
        if (s->d_done) HIP_TRY(fill_sync(s->d_done, 0, 33 * 64, s->stream));
    }
    s->generation = generation;            // the device counter follows at the next graph replay
    const int64_t done = samples_done(s);
    s->moment_base = done;
    return reset_run_state(s, /*eval_logp=*/false, generation, (uint32_t)done);
}

KMC_EXPORT kmc_status kmc_sampler_set_positions(kmc_sampler* s, const double* theta_host)
{
    if (!s || !theta_host) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (std::getenv("KMC_ABORT_BACKTRACE") || KMC_DIAG_ALWAYS) signal(SIGABRT, abort_backtrace);      // (diagnostics: somebody may have replaced it)
    HIP_TRY(hipSetDevice(s->cfg.device));
    const size_t nw = (size_t)s->nrows, nd = (size_t)s->cfg.ndim;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (!s->p2p) {
        HIP_TRY(upload_rows(s, s->d_pos, theta_host, nw));   // :198 (caller's array untouched)
    } else {
        // theta_host is the GLOBAL ensemble; keep this shard's slice of each half: local rows
        // [0,h_loc) = global [begin, begin+h_loc), local [h_loc,2h_loc) = global [h+begin, ...)
        const size_t hl = (size_t)s->h_loc;
        HIP_TRY(upload_rows(s, s->d_pos, theta_host + (size_t)s->active_begin * nd, hl));
        HIP_TRY(upload_rows(s, s->d_pos + hl * (size_t)s->ld, theta_host + ((size_t)s->h + (size_t)s->active_begin) * nd, hl));
        if (s->push) {      // the local copies of the other shards (block 1 + q = rank q's rows)
            for (int q = 0; q < s->cfg.shard_count; ++q) {
                if (q == s->cfg.shard_rank) continue;
                double* blk = s->d_pos + (size_t)(1 + q) * nw * (size_t)s->ld;
                HIP_TRY(upload_rows(s, blk, theta_host + (size_t)q * hl * nd, hl));
                HIP_TRY(upload_rows(s, blk + hl * (size_t)s->ld, theta_host + ((size_t)s->h + (size_t)q * hl) * nd, hl));
            }
        }
        HIP_TRY(fill_sync(s->d_flags, 0, 4096, s->stream));     // callers barrier across ranks before running
        HIP_TRY(fill_sync(s->d_err, 0, 64, s->stream));
        if (s->d_done) HIP_TRY(fill_sync(s->d_done, 0, 33 * 64, s->stream));
        if (s->lazy) {                               // every shadow is current, nothing has been accepted yet
            const size_t P = (size_t)s->cfg.shard_count, hl = (size_t)s->h_loc;
            HIP_TRY(fill_sync(s->d_lazy, 0, 2 * P * 2 * hl * sizeof(uint32_t) + 16, s->stream));
            HIP_TRY(fill_sync(s->peer_amap_in[s->cfg.shard_rank], 0, P * 4 * hl, s->stream));
        }
    }
    if (s->host_eval) {                                          // :209-210, on the caller's thread
        std::vector<double> lp0(nw);
        if (s->cfg.host_logpdf(theta_host, (int64_t)nw, (int64_t)nd, lp0.data(), s->cfg.host_user) != 0) {
            s->positions_set = false;
            return fail(KMC_ERR_BAD_ARG, "the host log-pdf callback failed on the initial ensemble");
        }
        HIP_TRY(copy_sync(s->d_logp, lp0.data(), nw * sizeof(double), hipMemcpyHostToDevice, s->stream));
    } else {
        KMC_TRY(eval_initial_logp(s));                           // :209-210
    }
    std::vector<double> lp(nw);
    HIP_TRY(copy_sync(lp.data(), s->d_logp, nw * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_naccept, 0, nw * sizeof(uint32_t), s->stream));
    HIP_TRY(hipMemsetAsync(s->d_gen, 0, 64, s->stream));
    if (s->d_msum) {
        HIP_TRY(hipMemsetAsync(s->d_msum, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        HIP_TRY(hipMemsetAsync(s->d_msumsq, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        if (s->d_klast) HIP_TRY(hipMemsetAsync(s->d_klast, 0, nw * sizeof(uint32_t), s->stream));
        if (s->d_mcnt) HIP_TRY(hipMemsetAsync(s->d_mcnt, 0, 2 * (size_t)s->mring_waves * sizeof(uint32_t), s->stream));
        if (s->d_isum) {
            const size_t ne = (size_t)s->nislands * 4 * (size_t)s->island_K;
            HIP_TRY(hipMemsetAsync(s->d_isum, 0, ne * sizeof(double), s->stream));
            HIP_TRY(hipMemsetAsync(s->d_isumsq, 0, ne * sizeof(double), s->stream));
        }
    }
    if (s->d_ids) {                                              // dealt sub-ensembles: slot i holds global walker r S + i
        hipLaunchKernelGGL(deal_init_ids, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s->stream, s->d_ids, (int64_t)nw,
                           (uint32_t)((uint64_t)s->cfg.deal_rank * (uint64_t)nw));
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->generation = 0;
    s->dev_gen = 0;
    s->moment_base = 0;
    s->launches = 0;
    s->have_run_events = false;
    if (s->stream_chain) { HIP_TRY(hipStreamSynchronize(s->copy_stream)); s->blocks_copied = 0; s->blocks_waited = 0; s->flushed_done = -1; }
    for (size_t w = 0; w < nw; ++w)
        if (!std::isfinite(lp[w])) {
            s->positions_set = false;
            return fail(KMC_ERR_NONFINITE_LOGP,
                        "walker " + std::to_string(w) + " has a non-finite initial log-pdf");
        }
    s->positions_set = true;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_set_chain_host(kmc_sampler* s, double* chain_host, double* chain_logp_host)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    if (!s->stream_chain) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STREAM_CHAIN (or stores no samples)");
    if ((s->d_chain && !chain_host) || (s->d_chain_logp && !chain_logp_host))
        return fail(KMC_ERR_BAD_ARG, "KMC_STREAM_CHAIN: a host buffer is needed for every stored quantity (KMC_STORE_CHAIN / KMC_STORE_LOGP)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    chain_unregister(s);
    s->flushed_done = -1;
    s->dst_chain = s->d_chain ? chain_host : nullptr;
    s->dst_logp = s->d_chain_logp ? chain_logp_host : nullptr;
    // page-lock the destination in place: the copies are then direct DMA into their final position and truly
    // asynchronous; without it (registration refused: limits, already registered) they are staged by the runtime
    const size_t nl = (size_t)s->nlocal, ns = (size_t)s->nsamples;
    if (s->dst_chain && std::getenv("KMC_NO_HOST_REGISTER") == nullptr) {
        if (hipHostRegister(s->dst_chain, ns * nl * (size_t)s->cfg.ndim * sizeof(double), hipHostRegisterDefault) == hipSuccess) s->dst_chain_reg = true;
        else (void)hipGetLastError();
    }
    if (s->dst_logp && std::getenv("KMC_NO_HOST_REGISTER") == nullptr) {
        if (hipHostRegister(s->dst_logp, ns * nl * sizeof(double), hipHostRegisterDefault) == hipSuccess) s->dst_logp_reg = true;
        else (void)hipGetLastError();
    }
    s->dev_dst_chain = s->dev_dst_logp = nullptr;
    if (s->stream_by_walker && ((s->dst_chain && !s->dst_chain_reg) || (s->dst_logp && !s->dst_logp_reg))) {
        // the by-walker copies are 2-D windows (or a kernel's stores) into the caller's arrays: page-locked memory only -- a 2-D
        // asynchronous copy into pageable memory is the call that aborted inside the runtime (upload_rows)
        chain_unregister(s);
        s->dst_chain = s->dst_logp = nullptr;
        return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER: the host buffers could not be page-locked (hipHostRegister); use the sample-major stream");
    }
    if (s->stream_by_walker) {
        // the by-walker copy is a kernel that writes into the caller's arrays: they must be mapped into the device's address space
        if (s->dst_chain && !s->bw_scratch && (!s->dst_chain_reg || hipHostGetDevicePointer((void**)&s->dev_dst_chain, s->dst_chain, 0) != hipSuccess)) {
            (void)hipGetLastError();
            chain_unregister(s);
            s->dst_chain = s->dst_logp = nullptr;
            return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER: the chain buffer could not be page-locked (hipHostRegister); use the sample-major stream");
        }
        if (s->dst_logp && !s->bw_scratch_logp && (!s->dst_logp_reg || hipHostGetDevicePointer((void**)&s->dev_dst_logp, s->dst_logp, 0) != hipSuccess)) {
            (void)hipGetLastError();
            chain_unregister(s);
            s->dst_chain = s->dst_logp = nullptr;
            return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER: the log-pdf buffer could not be page-locked (hipHostRegister); use the sample-major stream");
        }
    }
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_run(kmc_sampler* s, int64_t ngen)
{
    if (!s || ngen < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->positions_set) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_set_positions has not succeeded yet");
    if (s->cfg.shard_count != 1 && !s->p2p && !s->comm)
        return fail(KMC_ERR_UNSUPPORTED, "kmc_sampler_run needs shard_count == 1, KMC_P2P or an RCCL communicator (kmc_sampler_rccl_init); "
                                         "other replica-sharded drivers call kmc_sampler_half_step and exchange themselves");
    if (s->p2p && !s->connected) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_p2p_connect has not been called");
    if (s->generation + ngen >= (int64_t)1 << 31) return fail(KMC_ERR_UNSUPPORTED, "at most 2^31 - 1 generations (the step index is 32 bits)");
    if (s->stream_chain && ((s->d_chain && !s->dst_chain) || (s->d_chain_logp && !s->dst_logp)))
        return fail(KMC_ERR_BAD_ARG, "KMC_STREAM_CHAIN: call kmc_sampler_set_chain_host before kmc_sampler_run");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    if (s->resident) {
        // the whole ensemble lives in one workgroup's LDS; a launch carries up to 4096 generations
        while (ngen > 0) {
            const int64_t n = std::min<int64_t>(ngen, 4096);
            ResidentArgs ra{};
            IslandArgs& ia = ra.is;
            ia.pos = s->d_pos; ia.logp = s->d_logp; ia.naccept = s->d_naccept;
            ia.nwalkers = s->cfg.nwalkers; ia.permA = 1; ia.permC = 0;
            ia.gen0 = s->generation; ia.ngen = (int32_t)n;
            ia.ndim = (int32_t)s->cfg.ndim; ia.ld = (int32_t)s->ld;
            ia.nburnin = s->cfg.nburnin; ia.nthin = s->cfg.nthin; ia.nsamples = s->nsamples;
            const HalfStepArgs ha = make_args(s, 0, false, s->generation);
            ia.dc = ha.dc;                       // nhalf = nwalkers / 2, as in the multi-launch kernels
            ia.dp = s->dp;
            ia.msum = s->d_isum; ia.msumsq = s->d_isumsq;
            ra.S = (int32_t)s->cfg.nwalkers;
            ra.chain = s->d_chain; ra.chain_logp = s->d_chain_logp;
            if (s->user) {
                HIP_TRY(launch_module(s->uk.resident, 1u, 256u, s->stream, ra, (unsigned)s->island_lds));
            } else {
                hipLaunchKernelGGL(s->resident_kernel, dim3(1), dim3((unsigned)s->resident_tpb), s->island_lds, s->stream, ra);
                HIP_TRY(hipGetLastError());
            }
            s->generation += n;
            s->launches += 1;
            ngen -= n;
        }
        HIP_TRY(hipEventRecord(s->ev1, s->stream));
        s->have_run_events = true;
        return KMC_OK;
    }
    if (s->islands) {
        // one launch per epoch (or per piece of one, when a run stops inside an epoch)
        while (ngen > 0) {
            const int64_t epoch = s->generation / s->island_gens;
            const int64_t upto = (epoch + 1) * s->island_gens;
            const int64_t n = std::min<int64_t>(ngen, upto - s->generation);
            IslandArgs ia{};
            ia.pos = s->d_pos; ia.logp = s->d_logp; ia.naccept = s->d_naccept;
            ia.nwalkers = s->cfg.nwalkers;
            island_perm(s->cfg.seed, epoch, s->cfg.nwalkers, &ia.permA, &ia.permC);
            ia.gen0 = s->generation; ia.ngen = (int32_t)n;
            ia.ndim = (int32_t)s->cfg.ndim; ia.ld = (int32_t)s->ld;
            ia.nburnin = s->cfg.nburnin; ia.nthin = s->cfg.nthin; ia.nsamples = s->nsamples;
            const HalfStepArgs ha = make_args(s, 0, false, s->generation);
            ia.dc = ha.dc;
            ia.dc.nhalf = (uint32_t)(s->island_size / 2);
            ia.dp = s->dp;
            ia.msum = s->d_isum; ia.msumsq = s->d_isumsq;
            if (s->user) {
                HIP_TRY(launch_module(s->uk.island, (unsigned)s->nislands, (unsigned)s->island_size, s->stream, ia, (unsigned)s->island_lds));
            } else {
                hipLaunchKernelGGL(s->island_kernel, dim3((unsigned)s->nislands), dim3((unsigned)s->island_size), s->island_lds, s->stream, ia);
                HIP_TRY(hipGetLastError());
            }
            s->generation += n;
            s->launches += 1;
            ngen -= n;
        }
        HIP_TRY(hipEventRecord(s->ev1, s->stream));
        s->have_run_events = true;
        return KMC_OK;
    }
    if (s->host_eval) {
        // per half-step: PROPOSE on the device -> proposals to the host -> callback -> log-pdfs back
        // -> ACCEPT on the device (which recomputes the same proposals from the same draws)
        const size_t hh = (size_t)s->h, nd = (size_t)s->cfg.ndim, ld = (size_t)s->ld;
        for (; ngen > 0; --ngen) {
            KMC_TRY(chain_before(s, s->generation + 1));
            for (int half = 0; half < 2; ++half) {
                HalfStepArgs a = make_args(s, half, false, s->generation);
                a.prop_out = s->d_prop;
                HIP_TRY(launch_half_kernel(s, a));
                HIP_TRY(hipMemcpy2DAsync(s->h_prop, nd * sizeof(double), s->d_prop, ld * sizeof(double), nd * sizeof(double), hh,
                                         hipMemcpyDeviceToHost, s->stream));
                HIP_TRY(hipStreamSynchronize(s->stream));
                if (s->cfg.host_logpdf(s->h_prop, (int64_t)hh, (int64_t)nd, s->h_p1, s->cfg.host_user) != 0) {   // :257
                    s->positions_set = false;
                    return fail(KMC_ERR_BAD_ARG, "the host log-pdf callback failed in generation " + std::to_string(s->generation));
                }
                HIP_TRY(hipMemcpyAsync(s->d_p1, s->h_p1, hh * sizeof(double), hipMemcpyHostToDevice, s->stream));
                a.prop_out = nullptr;
                a.p1_in = s->d_p1;
                a.acc_out = s->d_acc;
                HIP_TRY(launch_half_kernel(s, a));
                s->launches += 2;
                if (s->cfg.host_accepted) {
                    HIP_TRY(hipMemcpyAsync(s->h_acc, s->d_acc, hh, hipMemcpyDeviceToHost, s->stream));
                    HIP_TRY(hipStreamSynchronize(s->stream));
                    const int32_t stored = (a.sched_inline.flags & kSample) != 0 ? 1 : 0;
                    if (s->cfg.host_accepted(s->h_acc, (int64_t)hh, (int64_t)half * (int64_t)hh, s->generation, stored, s->cfg.host_user) != 0) {
                        s->positions_set = false;
                        return fail(KMC_ERR_BAD_ARG, "the host accept callback failed in generation " + std::to_string(s->generation));
                    }
                }
            }
            s->generation += 1;
            KMC_TRY(chain_after(s));
        }
        HIP_TRY(hipEventRecord(s->ev1, s->stream));
        s->have_run_events = true;
        return KMC_OK;
    }
    auto eager_generations = [&](int64_t n) -> kmc_status {
        for (; n > 0; --n, --ngen) {
            KMC_TRY(chain_before(s, s->generation + 1));
            for (int half = 0; half < 2; ++half) KMC_TRY(launch_half(s, half, false, s->generation));
            s->generation += 1;
            s->launches += 2;
            if (++s->gens_since_sweep >= kSweepEvery) HIP_TRY(launch_sweep(s));
            KMC_TRY(chain_after(s));
        }
        return KMC_OK;
    };
    auto graph_chunk = [&]() -> kmc_status {
        KMC_TRY(ensure_graph(s));
        if (!s->graph_exec) return eager_generations(kGraphChunk);      // (RCCL all-gather that cannot be captured)
        KMC_TRY(sync_device_counter(s));
        KMC_TRY(chain_before(s, s->generation + kGraphChunk));
        HIP_TRY(hipGraphLaunch(s->graph_exec, s->stream));
        HIP_TRY(launch_sweep(s));
        s->generation += kGraphChunk;
        s->dev_gen += kGraphChunk;
        s->launches += 2 * kGraphChunk;
        ngen -= kGraphChunk;
        return chain_after(s);
    };
    auto updated_chunk = [&]() -> kmc_status {
        KMC_TRY(chain_before(s, s->generation + s->uchunk));
        if (launch_updated_graph(s) != KMC_OK) {       // nothing was enqueued: fall back to the table graph for good
            s->launch_mode = 1;
            return graph_chunk();
        }
        HIP_TRY(launch_sweep(s));
        s->generation += s->uchunk;
        s->launches += 2 * s->uchunk;
        ngen -= s->uchunk;
        return chain_after(s);
    };
    // How to issue the launches?  Same kernels, same results, three ways:
    //   1 table graph   -- hipGraph replay; the kernels read their generation from a device table (one scalar round
    //                      trip in front of Philox: C2 4.32 us per half-step); the host is free after ~16 ms per 10^4
    //                      generations;
    //   2 eager         -- the step travels among the preloaded kernel parameters, Philox starts at wave entry (C2 4.03 us)
    //                      -- as long as the host launches as fast as the GPU drains, which it does not reliably (three
    //                      bench runs: 8.7, 6.5, 8.6 x 10^9 walker-steps/s; C3 3.7-4.8 us against 3.25);
    //   3 updated graph -- the eager form of the kernels inside a graph whose node parameters are rewritten before
    //                      every replay (C2 4.00 us, C3 3.25 us; steady).  But hipGraphExecKernelNodeSetParams leaks ~80
    //                      bytes of host memory per call inside the runtime (1.6 MB per 10^4 generations, not returned
    //                      when the executables are destroyed; scripts/exp/leak_check.py), so the process has a BUDGET
    //                      of such calls (kUpdateBudgetCalls, KMC_UPDATED_BUDGET_MB): beyond it samplers choose
    //                      between 1 and 2.
    // A long run measures 1 against 3 (or 2) once -- four chunks each, HIP events: a starved GPU shows as idle time
    // between the events -- and keeps the faster; KMC_LAUNCH=graph|eager|updated decides without measuring.
    bool use_graph = !(s->cfg.flags & KMC_NO_GRAPH);
    const int64_t calib_min = 11 * std::max<int64_t>(kGraphChunk, s->uchunk) + kGraphChunk;
    auto calibrate = [&](bool with_updated) -> kmc_status {
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventCreate(&e2));
        const int64_t gens = 4 * std::max<int64_t>(kGraphChunk, s->uchunk);
        kmc_status st = graph_chunk();                                    // warm: instantiation, code objects
        if (st == KMC_OK) st = with_updated ? updated_chunk() : eager_generations(kGraphChunk);
        if (st == KMC_OK) st = with_updated ? updated_chunk() : eager_generations(kGraphChunk);
        if (st == KMC_OK && hipEventRecord(e0, s->stream) != hipSuccess) st = KMC_ERR_HIP;
        for (int64_t r = 0; r < gens / kGraphChunk && st == KMC_OK; ++r) st = graph_chunk();
        if (st == KMC_OK && hipEventRecord(e1, s->stream) != hipSuccess) st = KMC_ERR_HIP;
        if (with_updated) { for (int64_t r = 0; r < gens / s->uchunk && st == KMC_OK && s->launch_mode == 0; ++r) st = updated_chunk(); }
        else if (st == KMC_OK) st = eager_generations(gens);
        if (st == KMC_OK && hipEventRecord(e2, s->stream) != hipSuccess) st = KMC_ERR_HIP;
        float tg = 0.f, tu = 0.f;
        if (st == KMC_OK && (hipEventSynchronize(e2) != hipSuccess || hipEventElapsedTime(&tg, e0, e1) != hipSuccess ||
                             hipEventElapsedTime(&tu, e1, e2) != hipSuccess)) st = KMC_ERR_HIP;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
        if (st != KMC_OK) return st == KMC_ERR_HIP ? fail(st, "launch-mode calibration failed") : st;
        if (s->launch_mode == 0) {
            s->launch_mode = tu < 0.98f * tg ? (with_updated ? 3 : 2) : 1;        // the alternative must win clearly
            s->calib_graph_ms = tg * (float)kGraphChunk / (float)gens; s->calib_eager_ms = tu * (float)kGraphChunk / (float)gens;
        }
        return KMC_OK;
    };
    if (use_graph && s->launch_mode == 0) {
        if (!s->uexec[0])
            if (const char* e = std::getenv("KMC_UPD_CHUNK")) { const long v = std::atol(e); if (v >= 16 && v <= 1024) s->uchunk = v; }
        const char* env = std::getenv("KMC_LAUNCH");
        if (env && std::strcmp(env, "graph") == 0) s->launch_mode = 1;
        else if (env && std::strcmp(env, "eager") == 0) s->launch_mode = 2;
        else if (env && std::strcmp(env, "updated") == 0 && updated_graph_possible(s)) { s->launch_mode = 3; s->updated_forced = true; }
        else if (!updated_graph_possible(s)) s->launch_mode = 1;
        else if (ngen >= calib_min) KMC_TRY(calibrate(update_budget_left()));
    }
    while (use_graph && s->launch_mode == 3 && ngen >= s->uchunk) {
        if (!s->updated_forced && !update_budget_left()) {             // the process has used up its leak budget: decide again, between 1 and 2
            s->launch_mode = 0;
            if (ngen >= calib_min) KMC_TRY(calibrate(false));
            break;
        }
        KMC_TRY(updated_chunk());
    }
    if (s->launch_mode == 2) use_graph = false;
    while (use_graph && ngen >= kGraphChunk) KMC_TRY(graph_chunk());
    KMC_TRY(eager_generations(ngen));
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    s->have_run_events = true;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_half_step(kmc_sampler* s, int half)
{
    if (!s || (half != 0 && half != 1)) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->positions_set) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_set_positions has not succeeded yet");
    if (s->p2p && !s->connected) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_p2p_connect has not been called");
    if (s->islands) return fail(KMC_ERR_UNSUPPORTED, "island mode advances whole generations: use kmc_sampler_run");
    if (s->host_eval) return fail(KMC_ERR_UNSUPPORTED, "KMC_HOST_DENSITY: a half-step includes the host callback; use kmc_sampler_run");
    if (s->resident) return fail(KMC_ERR_UNSUPPORTED, "this small ensemble runs in resident mode (whole generations per launch); create it with KMC_NO_GRAPH to step by halves");
    if (s->generation >= ((int64_t)1 << 31) - 1) return fail(KMC_ERR_UNSUPPORTED, "at most 2^31 - 1 generations (the step index is 32 bits)");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (half == 0) KMC_TRY(chain_before(s, s->generation + 1));
    KMC_TRY(launch_half(s, half, false, s->generation));
    s->launches += 1;
    if (half == 1) {
        s->generation += 1;
        if (++s->gens_since_sweep >= kSweepEvery) HIP_TRY(launch_sweep(s));
        KMC_TRY(chain_after(s));
    }
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_sync(kmc_sampler* s)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(chain_flush(s));
    return check_p2p_err(s);
}

KMC_EXPORT kmc_status kmc_sampler_last_run_ms(kmc_sampler* s, double* ms)
{
    if (!s || !ms) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->have_run_events) return fail(KMC_ERR_BAD_ARG, "no kmc_sampler_run yet");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipEventSynchronize(s->ev1));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, s->ev0, s->ev1));
    *ms = (double)f;
    return KMC_OK;
}

KMC_EXPORT int64_t kmc_sampler_generation(const kmc_sampler* s) { return s ? s->generation : -1; }
KMC_EXPORT int64_t kmc_sampler_nsamples(const kmc_sampler* s) { return s ? s->nsamples : -1; }
KMC_EXPORT int64_t kmc_sampler_launch_count(const kmc_sampler* s) { return s ? s->launches : -1; }

// Human-readable description of how this sampler executes (kernel family, geometry, exchange).
KMC_EXPORT kmc_status kmc_sampler_describe(const kmc_sampler* s, char* buf, int64_t buflen)
{
    if (!s || !buf || buflen <= 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    std::ostringstream o;
    if (s->islands)
        o << "island mode: " << s->nislands << " islands of " << s->island_size << " walkers in LDS, " << s->island_gens
          << " generations per launch, rows 2 lanes x " << s->island_K << " chunks";
    else if (s->resident)
        o << "resident mode (exact): whole ensemble in one workgroup's LDS (" << (s->user ? 256 : s->resident_tpb)
          << " threads), up to 4096 generations per launch, rows 2 lanes x " << s->island_K << " chunks";
    else if (s->host_eval)
        o << "host-evaluated density (exact): per half-step propose kernel -> D2H -> callback -> H2D -> accept kernel, grid "
          << s->grid << " x 256";
    else if (s->plan.vec) {
        o << "multi-launch (exact): half_step_vec L=" << s->plan.L << " K=" << s->plan.K << " ITER=" << s->plan.ITER
          << (s->plan.ragged ? " ragged" : " exact-size") << ", grid " << s->grid << " x " << s->tpb
          << (((s->cfg.flags & KMC_NO_GRAPH) || s->launch_mode == 2) ? ", eager launches (step among the preloaded kernel parameters)"
              : s->launch_mode == 3 ? ", hipGraph replay of 64 generations with per-replay parameter updates (step preloaded)"
                                    : ", hipGraph replay of 64 generations");
        if (s->calib_graph_ms > 0.f) {
            char b[128];
            std::snprintf(b, sizeof(b), " (measured per 64 generations: table graph %.3f ms, %s %.3f ms)", s->calib_graph_ms, s->launch_mode == 2 ? "eager launches" : "updated graph", s->calib_eager_ms);
            o << b;
        }
    } else
        o << "multi-launch (exact): " << (s->user && s->uk.staged ? "half_step_staged (one walker per lane, rows staged through LDS)" : "half_step_generic (one walker per lane)")
          << ", grid " << s->grid << " x " << s->tpb;
    if (s->lazy) o << "; lazy pull into local copies (KMC_P2P_LAZY)";
    else if (s->push) o << "; accepted rows pushed into the peers' local copies (KMC_P2P_PUSH)";
    if (s->f32) o << "; rows kept in float (KMC_F32), arithmetic in double";
    if (s->stream_chain) o << "; chain streamed to host memory in blocks of " << s->ring_blk << " samples (device ring of 3 blocks" << (s->dst_chain_reg || s->dst_logp_reg ? ", destination page-locked" : "") << ")";
    if (s->d_ids) o << "; dealt sub-ensemble " << s->cfg.deal_rank << "/" << s->cfg.deal_count << " (walkers re-dealt between epochs)";
    if (s->d_mring) o << "; moments through a ring of " << s->mring_depth << " posted rows per wave";
    if (s->user) o << "; runtime-compiled density";
    if (s->p2p) o << "; P2P shard " << s->cfg.shard_rank << "/" << s->cfg.shard_count << (s->connected ? "" : " (not connected)");
    else if (s->cfg.shard_count > 1 || s->comm)
        o << "; replica shard " << s->cfg.shard_rank << "/" << s->cfg.shard_count
          << (s->comm ? ((s->comm_graph_ok && !(s->cfg.flags & KMC_NO_GRAPH) && s->launch_mode != 2) ? ", RCCL all-gather of the updated half after every half-step (captured in the graph)"
                                                                                                         : ", RCCL all-gather of the updated half after every half-step (enqueued launch by launch)") : "");
    const std::string t = o.str();
    std::snprintf(buf, (size_t)buflen, "%s", t.c_str());
    return KMC_OK;
}

KMC_EXPORT void* kmc_sampler_device_ptr(kmc_sampler* s, int which)
{
    if (!s) return nullptr;
    switch (which) {
    case 0: return s->d_pos;
    case 1: return s->d_logp;
    case 2: return s->d_naccept;
    default: return nullptr;
    }
}

KMC_EXPORT kmc_status kmc_sampler_get_positions(kmc_sampler* s, double* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    HIP_TRY(download_rows(s, host, s->d_pos, (size_t)s->nrows));
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_logp(kmc_sampler* s, double* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    HIP_TRY(copy_sync(host, s->d_logp, (size_t)s->nrows * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_naccept(kmc_sampler* s, int64_t* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    std::vector<uint32_t> tmp((size_t)s->nrows);
    HIP_TRY(copy_sync(tmp.data(), s->d_naccept, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    for (size_t i = 0; i < tmp.size(); ++i) host[i] = (int64_t)tmp[i];
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_accept_ratio(kmc_sampler* s, double* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    std::vector<int64_t> na((size_t)s->nrows);
    KMC_TRY(kmc_sampler_get_naccept(s, na.data()));
    const double denom = (double)(s->generation - s->cfg.nburnin);   // :291 (0 -> inf/nan like the reference)
    for (size_t i = 0; i < na.size(); ++i) host[i] = (double)na[i] / denom;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_moments(kmc_sampler* s, double* sum, double* sumsq, int64_t* n)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    if (!s->d_msum) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_MOMENTS");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (s->islands || s->resident) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        const int64_t nd = s->cfg.ndim;
        const size_t per = 4 * (size_t)s->island_K, ne = (size_t)s->nislands * per;
        std::vector<double> hs(ne), hq(ne);
        HIP_TRY(copy_sync(hs.data(), s->d_isum, ne * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(copy_sync(hq.data(), s->d_isumsq, ne * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        std::vector<double> S((size_t)nd, 0.0), Q((size_t)nd, 0.0);
        for (int64_t b = 0; b < s->nislands; ++b)
            for (size_t e = 0; e < per; ++e)
                if ((int64_t)e < nd) { S[e] += hs[(size_t)b * per + e]; Q[e] += hq[(size_t)b * per + e]; }
        for (int64_t d = 0; d < nd; ++d) {
            if (sum) sum[d] = S[d];
            if (sumsq) sumsq[d] = Q[d];
        }
        if (n) *n = (samples_done(s) - s->moment_base) * s->nlocal;
        return KMC_OK;
    }
    KMC_TRY(flush_moments_now(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    const int64_t nd = s->cfg.ndim;
    std::vector<double> hs((size_t)s->macc_elems), hq((size_t)s->macc_elems);
    HIP_TRY(copy_sync(hs.data(), s->d_msum, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(copy_sync(hq.data(), s->d_msumsq, hq.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    std::vector<double> S((size_t)nd, 0.0), Q((size_t)nd, 0.0);
    if (s->plan.vec && s->plan.K == 2 && (s->plan.L == 8 || s->plan.L == 16 || s->plan.L == 32)) {
        // transposed fold (kmc_kernels.hpp, FoldT): every lane of a wave owns NVL of the wave's 8 L / 64 * 64 sums
        const int L = s->plan.L, NVL = 8 * L / 64;
        const int64_t nwaves = s->macc_stride / 64;
        for (int64_t w = 0; w < nwaves; ++w)
            for (int r = 0; r < NVL; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
                    const int v = L == 8 ? 4 * b3 + 2 * b4 + b5 : L == 16 ? 4 * b4 + 2 * b5 + r : 4 * b5 + r;
                    const int64_t d = 2 * ((int64_t)((v >> 1) & 1) * L + (lane & (L - 1))) + (v & 1);
                    if (d >= nd) continue;
                    const double x = hs[(size_t)((w * NVL + r) * 64 + lane)];
                    if (v >> 2) Q[(size_t)d] += x; else S[(size_t)d] += x;
                }
    } else if (s->plan.vec) {
        const int L = s->plan.L, K = s->plan.K;
        for (int k = 0; k < K; ++k)
            for (int64_t t = 0; t < s->macc_stride; ++t) {
                const int64_t d0 = 2 * ((int64_t)k * L + (t % L));
                const int64_t idx = 2 * ((int64_t)k * s->macc_stride + t);
                if (d0 < nd) { S[d0] += hs[idx]; Q[d0] += hq[idx]; }
                if (d0 + 1 < nd) { S[d0 + 1] += hs[idx + 1]; Q[d0 + 1] += hq[idx + 1]; }
            }
    } else {
        for (int64_t d = 0; d < nd; ++d)
            for (int64_t t = 0; t < s->macc_stride; ++t) {
                S[d] += hs[d * s->macc_stride + t];
                Q[d] += hq[d * s->macc_stride + t];
            }
    }
    for (int64_t d = 0; d < nd; ++d) {
        if (sum) sum[d] = S[d];
        if (sumsq) sumsq[d] = Q[d];
    }
    if (n) *n = (samples_done(s) - s->moment_base) * s->nlocal;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_chain(kmc_sampler* s, double* chain, double* chain_logp)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    const size_t rows = (size_t)samples_done(s) * (size_t)s->nlocal;
    if (s->stream_by_walker)
        return fail(KMC_ERR_UNSUPPORTED, "this sampler streams its chain by walker into the caller's buffers (kmc_sampler_get_chain_by_walker)");
    if (s->stream_chain) {             // the chain is in the caller's host buffers already
        KMC_TRY(chain_flush(s));
        if (chain && !s->d_chain) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_CHAIN");
        if (chain_logp && !s->d_chain_logp) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_LOGP");
        if (chain && chain != s->dst_chain) std::memcpy(chain, s->dst_chain, rows * (size_t)s->cfg.ndim * sizeof(double));
        if (chain_logp && chain_logp != s->dst_logp) std::memcpy(chain_logp, s->dst_logp, rows * sizeof(double));
        return KMC_OK;
    }
    if (chain) {
        if (!s->d_chain && s->nsamples > 0) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_CHAIN");
        HIP_TRY(download_rows(s, chain, s->d_chain, rows));
    }
    if (chain_logp) {
        if (!s->d_chain_logp && s->nsamples > 0) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_LOGP");
        if (rows) HIP_TRY(copy_sync(chain_logp, s->d_chain_logp, rows * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    }
    return KMC_OK;
}

// Device chain [K][nl][ld] (T = float or double) -> host [nl][K][width] doubles: transposed on the device into a scratch
// buffer, a piece of walkers (<= ~256 MiB, KMC_BY_WALKER_PIECE_MB) at a time, each piece one contiguous copy.
kmc_status kmc_host::download_by_walker(const void* src, bool is_float, int64_t nl, int64_t ld, int64_t width, int64_t K, double* dst_host, hipStream_t st)
{
    if (K <= 0 || nl <= 0) return KMC_OK;
    const size_t per_walker = (size_t)K * (size_t)width * sizeof(double);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)256 << 20;
    if (const char* mb = std::getenv("KMC_BY_WALKER_PIECE_MB")) { const double v = std::atof(mb); if (v > 0.0) budget = (size_t)(v * 1048576.0); }
    if (budget > free_b / 2) budget = free_b / 2;
    int64_t wb = (int64_t)(budget / per_walker);
    if (wb < 1) wb = 1;
    if (wb > nl) wb = nl;
    double* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)wb * per_walker));
    hipError_t e = hipSuccess;
    for (int64_t w0 = 0; w0 < nl && e == hipSuccess; w0 += wb) {
        const int64_t n = nl - w0 < wb ? nl - w0 : wb;
        int64_t gy = (K * width + 255) / 256;
        if (gy > 4096) gy = 4096;
        if (is_float)
            hipLaunchKernelGGL(chain_by_walker<float>, dim3((unsigned)n, (unsigned)gy), dim3(256), 0, st, static_cast<const float*>(src), tmp, nl, (int32_t)ld, (int32_t)width, K, w0, n, K * width);
        else
            hipLaunchKernelGGL(chain_by_walker<double>, dim3((unsigned)n, (unsigned)gy), dim3(256), 0, st, static_cast<const double*>(src), tmp, nl, (int32_t)ld, (int32_t)width, K, w0, n, K * width);
        e = hipGetLastError();
        if (e == hipSuccess) e = copy_sync(dst_host + (size_t)w0 * (size_t)K * (size_t)width, tmp, (size_t)n * per_walker, hipMemcpyDeviceToHost, st);   // (waits: one scratch buffer)
    }
    (void)hipFree(tmp);
    HIP_TRY(e);
    return KMC_OK;
}

// The chain in the reference's order: [walker][sample][ndim] and [walker][sample] (thetas[w][k], logdensities[w][k],
// src/samplers.jl:219-221).  Transposed on the device, a block of walkers at a time, and copied out contiguously.
KMC_EXPORT kmc_status kmc_sampler_get_chain_by_walker(kmc_sampler* s, double* chain, double* chain_logp)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    KMC_TRY(check_p2p_err(s));
    if (s->stream_by_walker) {         // the chain is in the caller's buffers already: [nlocal][nsamples][ndim], [nlocal][nsamples]
        KMC_TRY(chain_flush(s));
        const size_t n = (size_t)s->nlocal * (size_t)s->nsamples;
        if (chain && chain != s->dst_chain && s->dst_chain) std::memcpy(chain, s->dst_chain, n * (size_t)s->cfg.ndim * sizeof(double));
        if (chain_logp && chain_logp != s->dst_logp && s->dst_logp) std::memcpy(chain_logp, s->dst_logp, n * sizeof(double));
        return KMC_OK;
    }
    if (s->stream_chain)
        return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN delivers the chain sample-major into the caller's buffers (kmc_sampler_get_chain)");
    const int64_t K = samples_done(s), nl = s->nlocal, nd = s->cfg.ndim;
    if (chain && !s->d_chain && s->nsamples > 0) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_CHAIN");
    if (chain_logp && !s->d_chain_logp && s->nsamples > 0) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_LOGP");
    if (K == 0 || (!chain && !chain_logp)) return KMC_OK;
    if (chain) KMC_TRY(download_by_walker(s->d_chain, s->f32, nl, s->ld, nd, K, chain, s->stream));
    if (chain_logp) KMC_TRY(download_by_walker(s->d_chain_logp, false, nl, 1, 1, K, chain_logp, s->stream));
    return KMC_OK;
}

// ------------------------------------------------------------------------------------------
// dealt sub-ensembles (include/kissmcmc_hip.h: "dealt sub-ensembles")
// ------------------------------------------------------------------------------------------
KMC_EXPORT uint64_t kmc_deal_seed(uint64_t seed, int32_t deal_rank) { return deal_seed(seed, deal_rank); }

KMC_EXPORT kmc_status kmc_deal_perm(uint64_t seed, int64_t epoch, int32_t deal_rank, int64_t S, int64_t* A, int64_t* C)
{
    if (!A || !C || S < 2 || epoch < 0 || deal_rank < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    deal_perm(seed, epoch, deal_rank, S, A, C);
    return KMC_OK;
}

namespace {
DealArgs deal_args(kmc_sampler* s, void* buf)
{
    DealArgs a{};
    a.pos = s->d_pos; a.logp = s->d_logp; a.naccept = s->d_naccept; a.ids = s->d_ids;
    a.buf = static_cast<double*>(buf);
    a.S = s->nrows; a.A = 1; a.C = 0;
    a.ndim = (int32_t)s->cfg.ndim; a.ld = (int32_t)s->ld;
    return a;
}
}  // namespace

KMC_EXPORT kmc_status kmc_sampler_deal_pack(kmc_sampler* s, int64_t epoch, void* send_dev)
{
    if (!s || !send_dev || epoch < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->d_ids) return fail(KMC_ERR_BAD_ARG, "sampler was created without kmc_config.deal_count");
    if (!s->positions_set) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_set_positions has not succeeded yet");
    HIP_TRY(hipSetDevice(s->cfg.device));
    KMC_TRY(flush_moments_now(s));          // the accumulators are per slot: settle them before walkers change slots
    DealArgs a = deal_args(s, send_dev);
    deal_perm(s->user_seed, epoch, s->cfg.deal_rank, s->nrows, &a.A, &a.C);
    const int64_t n = a.S * ((int64_t)a.ndim + 2);
    hipLaunchKernelGGL(deal_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, a);
    HIP_TRY(hipGetLastError());
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_deal_unpack(kmc_sampler* s, const void* recv_dev)
{
    if (!s || !recv_dev) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->d_ids) return fail(KMC_ERR_BAD_ARG, "sampler was created without kmc_config.deal_count");
    HIP_TRY(hipSetDevice(s->cfg.device));
    const DealArgs a = deal_args(s, const_cast<void*>(recv_dev));
    const int64_t n = a.S * ((int64_t)a.ndim + 2);
    hipLaunchKernelGGL(deal_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, a);
    HIP_TRY(hipGetLastError());
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_get_walker_ids(kmc_sampler* s, int64_t* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (!s->d_ids) { for (int64_t i = 0; i < s->nrows; ++i) host[i] = i; return KMC_OK; }
    std::vector<uint32_t> tmp((size_t)s->nrows);
    HIP_TRY(copy_sync(tmp.data(), s->d_ids, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    for (size_t i = 0; i < tmp.size(); ++i) host[i] = (int64_t)tmp[i];
    return KMC_OK;
}

// dealt sub-ensembles, after kmc_sampler_set_state: which global walker each slot holds (a function of the restored generation's
// epoch alone: replay kmc_deal_perm on the host, distributed.deal_slot_ids)
KMC_EXPORT kmc_status kmc_sampler_set_walker_ids(kmc_sampler* s, const int64_t* host)
{
    if (!s || !host) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->d_ids) return fail(KMC_ERR_BAD_ARG, "sampler was created without kmc_config.deal_count");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<uint32_t> tmp((size_t)s->nrows);
    for (size_t i = 0; i < tmp.size(); ++i) {
        if (host[i] < 0 || host[i] >= (int64_t)1 << 32) return fail(KMC_ERR_BAD_ARG, "walker index out of range");
        tmp[i] = (uint32_t)host[i];
    }
    HIP_TRY(copy_sync(s->d_ids, tmp.data(), tmp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    return KMC_OK;
}

KMC_EXPORT int kmc_sizeof_config(void) { return (int)sizeof(kmc_config); }
KMC_EXPORT int kmc_sizeof_metropolis_config(void) { return (int)sizeof(kmc_metropolis_config); }

// ------------------------------------------------------------------------------------------
// one-shot: emcee(), src/samplers.jl:188-216 + :232-293
// ------------------------------------------------------------------------------------------
KMC_EXPORT kmc_status kmc_emcee_run(const kmc_config* cfg, const double* theta0, kmc_outputs* out)
{
    if (!cfg || !theta0 || !out) return fail(KMC_ERR_BAD_ARG, "null argument");
    kmc_config c = *cfg;
    if (out->chain) c.flags |= KMC_STORE_CHAIN;
    if (out->chain_logp) c.flags |= KMC_STORE_LOGP;
    if (out->sum || out->sumsq) c.flags |= KMC_MOMENTS;
    c.shard_rank = 0;
    c.shard_count = 1;
    if ((out->chain || out->chain_logp) && c.dtype == KMC_F64 && !(c.flags & (KMC_ISLANDS | KMC_P2P)) && c.nthin > 0 && c.ngenerations > c.nburnin) {
        // a chain that does not fit the device is streamed to the caller's buffers while sampling (KMC_STREAM_CHAIN)
        size_t free_b = 0, total_b = 0;
        const size_t ns = (size_t)((c.ngenerations - c.nburnin) / c.nthin), nw_ = (size_t)c.nwalkers;
        const size_t need = ns * nw_ * ((out->chain ? (size_t)(c.ndim + (c.ndim & 1)) * sizeof(double) : 0) + (out->chain_logp ? sizeof(double) : 0));
        if (hipSetDevice(c.device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 10 * 8) c.flags |= KMC_STREAM_CHAIN;
        else (void)hipGetLastError();
    }
    kmc_sampler* s = nullptr;
    KMC_TRY(kmc_sampler_create(&c, &s));
    kmc_status st = KMC_OK;
    if (s->stream_chain) st = kmc_sampler_set_chain_host(s, out->chain, out->chain_logp);
    if (st == KMC_OK) st = kmc_sampler_set_positions(s, theta0);
    if (st == KMC_OK) st = kmc_sampler_run(s, c.ngenerations);
    if (st == KMC_OK) st = kmc_sampler_sync(s);
    if (st == KMC_OK) st = kmc_sampler_last_run_ms(s, &out->device_ms);
    if (st == KMC_OK && (out->chain || out->chain_logp))
        st = (c.flags & KMC_CHAIN_BY_WALKER) ? kmc_sampler_get_chain_by_walker(s, out->chain, out->chain_logp) : kmc_sampler_get_chain(s, out->chain, out->chain_logp);
    if (st == KMC_OK && out->accept_ratio) st = kmc_sampler_get_accept_ratio(s, out->accept_ratio);
    if (st == KMC_OK && out->naccept) st = kmc_sampler_get_naccept(s, out->naccept);
    if (st == KMC_OK && out->final_pos) st = kmc_sampler_get_positions(s, out->final_pos);
    if (st == KMC_OK && out->final_logp) st = kmc_sampler_get_logp(s, out->final_logp);
    out->nmoment = 0;
    if (st == KMC_OK && (out->sum || out->sumsq)) st = kmc_sampler_get_moments(s, out->sum, out->sumsq, &out->nmoment);
    out->nsamples = s->nsamples;
    kmc_sampler_destroy(s);
    return st;
}

// ------------------------------------------------------------------------------------------
// stateless op
// ------------------------------------------------------------------------------------------
KMC_EXPORT kmc_status kmc_logpdf_eval(const kmc_config* cfg, const double* pos_dev, double* logp_dev,
                                      int64_t nrows, void* hip_stream)
{
    if (!cfg || !pos_dev || !logp_dev || nrows < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    DensityParams dp;
    KMC_TRY(digest_params(*cfg, &dp));
    if (nrows == 0) return KMC_OK;
    const LogpdfArgs la{pos_dev, logp_dev, nrows, (int32_t)cfg->ndim, (int32_t)cfg->ndim, dp};
    const unsigned grid = (unsigned)((nrows + 255) / 256);
    if (cfg->density == KMC_USER_DENSITY) {
        UserKernels uk;
        KMC_TRY(load_user(static_cast<kmc_user_density*>(cfg->user_density), false, 0, 0, 0, false, &uk, 0, false, 0, false, cfg->ndim));
        const hipError_t e = launch_module(uk.logpdf, grid, 256u, (hipStream_t)hip_stream, la);
        if (e == hipSuccess) (void)hipStreamSynchronize((hipStream_t)hip_stream);   // the module is unloaded below
        (void)hipModuleUnload(uk.mod);
        HIP_TRY(e);
        return KMC_OK;
    }
    if (cfg->density == KMC_HOST_DENSITY) return fail(KMC_ERR_UNSUPPORTED, "KMC_HOST_DENSITY is evaluated by the caller, not on the device");
    HalfStepFn v, g;
    LogpdfFn lp = nullptr;
    if (!lookup(cfg->density, 0, 0, 1, false, false, false, &v, &g, &lp)) return fail(KMC_ERR_BAD_ARG, "unknown density id");
    hipLaunchKernelGGL(lp, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, la);
    HIP_TRY(hipGetLastError());
    return KMC_OK;
}

// Host-buffer convenience: logp[i] = log pdf(pos[i]) for dense host rows (used by the host shims for
// make_theta0s' `pdf(theta) > -Inf` test with runtime-compiled densities).
KMC_EXPORT kmc_status kmc_logpdf_eval_host(const kmc_config* cfg, const double* pos_host, double* logp_host, int64_t nrows)
{
    if (!cfg || !pos_host || !logp_host || nrows < 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (nrows == 0) return KMC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(KMC_ERR_NO_DEVICE, "no HIP device visible");
    }
    HIP_TRY(hipSetDevice(cfg->device));
    ScopedStream ss;
    HIP_TRY(ss.create());
    double *dpos = nullptr, *dlp = nullptr;
    const size_t nb = (size_t)nrows * (size_t)cfg->ndim * sizeof(double);
    HIP_TRY(hipMalloc(&dpos, nb));
    hipError_t e = hipMalloc(&dlp, (size_t)nrows * sizeof(double));
    kmc_status st = KMC_OK;
    if (e == hipSuccess) e = copy_sync(dpos, pos_host, nb, hipMemcpyHostToDevice, ss.st);
    if (e == hipSuccess) st = kmc_logpdf_eval(cfg, dpos, dlp, nrows, ss.st);
    if (e == hipSuccess && st == KMC_OK) e = hipStreamSynchronize(ss.st);
    if (e == hipSuccess && st == KMC_OK) e = copy_sync(logp_host, dlp, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, ss.st);
    (void)hipFree(dpos);
    (void)hipFree(dlp);
    if (st != KMC_OK) return st;
    HIP_TRY(e);
    return KMC_OK;
}

// The same on the chain a sampler holds on the device (KMC_STORE_CHAIN; the samples stored so far): no host round trip.
KMC_EXPORT kmc_status kmc_sampler_int_acorr(kmc_sampler* s, double c, double* tau, double* converged)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null sampler");
    if (!s->d_chain) return fail(KMC_ERR_BAD_ARG, "sampler was created without KMC_STORE_CHAIN");
    if (s->stream_chain) return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN: the chain is on the host; use kmc_int_acorr on it");
    if (s->ld != s->cfg.ndim) return fail(KMC_ERR_UNSUPPORTED, "odd ndim: rows are padded on the device; use kmc_int_acorr on the downloaded chain");
    if (s->f32) return fail(KMC_ERR_UNSUPPORTED, "KMC_F32: the device chain is float; use kmc_int_acorr on the downloaded chain");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int64_t ns = samples_done(s);
    KMC_TRY(int_acorr_check(ns, s->nlocal, s->cfg.ndim, c, tau, converged));
    return int_acorr_device(s->d_chain, ns, s->nlocal, s->cfg.ndim, c, tau, converged);
}
