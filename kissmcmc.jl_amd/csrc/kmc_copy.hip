// kmc_copy.hip -- host <-> device movement: blocking copies staged through page-locked bounce buffers (kmc_host.hpp: copy_sync),
// the walkers' rows in and out, and sample storage (reference src/samplers.jl:268-272): the streamed chain ring
// (KMC_STREAM_CHAIN) and the read-out of a device chain, sample-major or in the reference's thetas[w][k] order.
#include <chrono>
#include <pthread.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <thread>

#include "kmc_sampler.hpp"
#include "kmc_copy_kernels.hpp"

using namespace kmc;
using namespace kmc_host;

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

namespace kmc_host {
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

// the tiled transposition [sample][walker][ld] -> [walker][sample][nd] (kmc_copy_kernels.hpp: chain_by_walker), walkers [w0, w0 + nw), K samples
hipError_t launch_by_walker(const void* src, bool is_float, double* dst, int64_t nl, int64_t ld, int64_t nd, int64_t K, int64_t w0, int64_t nw,
                            int64_t dst_stride, hipStream_t st)
{
    if (K <= 0 || nw <= 0) return hipSuccess;
    const ByWalkerTile t = by_walker_tile((int32_t)nd);
    const int64_t gx = (nw + t.TW - 1) / t.TW, gy = (K + t.TK - 1) / t.TK, gz = (nd + t.NC - 1) / t.NC;
    // (gridDim.y <= 65535: runs of more samples go in slices of samples -- same kernel, the destination shifted)
    for (int64_t y0 = 0; y0 < gy; y0 += 65535) {
        const int64_t ny = std::min<int64_t>(65535, gy - y0), ks = y0 * t.TK, kn = std::min<int64_t>(K - ks, ny * t.TK);
        if (is_float)
            hipLaunchKernelGGL(chain_by_walker<float>, dim3((unsigned)gx, (unsigned)ny, (unsigned)gz), dim3(256), t.lds_bytes, st, static_cast<const float*>(src) + ks * nl * ld,
                               dst + ks * nd, nl, (int32_t)ld, (int32_t)nd, kn, w0, nw, dst_stride, t.TW, t.TK, t.NC);
        else
            hipLaunchKernelGGL(chain_by_walker<double>, dim3((unsigned)gx, (unsigned)ny, (unsigned)gz), dim3(256), t.lds_bytes, st, static_cast<const double*>(src) + ks * nl * ld,
                               dst + ks * nd, nl, (int32_t)ld, (int32_t)nd, kn, w0, nw, dst_stride, t.TW, t.TK, t.NC);
    }
    return hipGetLastError();
}

// ---- KMC_STREAM_CHAIN ---------------------------------------------------------------------------------------
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
        // per walker: transposed into a device scratch block and copied out by the DMA engine as a 2-D window.  (A variant whose
        // kernel wrote straight into the page-locked arrays returned 1 543 wrong elements once in ~15 runs, page-granular, and was
        // never reproduced or explained: removed in round 4, profiles/NOTES.md.)
        const int64_t ns = s->nsamples;
        if (s->d_chain && s->dst_chain) {
            HIP_TRY(launch_by_walker(s->d_chain + slot0 * nl * ld, false, s->bw_scratch, (int64_t)nl, (int64_t)ld, (int64_t)nd, (int64_t)n, 0, (int64_t)nl, (int64_t)(n * nd), s->copy_stream));
            if (s->dst_chain_reg) {
                HIP_TRY(hipMemcpy2DAsync(s->dst_chain + (size_t)k0 * nd, (size_t)ns * nd * sizeof(double), s->bw_scratch, n * nd * sizeof(double),
                                         n * nd * sizeof(double), nl, hipMemcpyDeviceToHost, s->copy_stream));
            } else {
                // pageable destination (hipHostRegister refused: RLIMIT_MEMLOCK in a container is enough): never a 2-D asynchronous
                // copy into it -- the transposed block comes over contiguously through the bounce buffers (blocking), then
                // each walker's run of n samples is put in place by the host
                if (s->bw_host.size() < n * nl * nd) s->bw_host.resize(n * nl * nd);
                HIP_TRY(copy_sync(s->bw_host.data(), s->bw_scratch, n * nl * nd * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
                for (size_t w = 0; w < nl; ++w)
                    std::memcpy(s->dst_chain + (w * (size_t)ns + (size_t)k0) * nd, s->bw_host.data() + w * n * nd, n * nd * sizeof(double));
            }
        }
        if (s->d_chain_logp && s->dst_logp) {
            HIP_TRY(launch_by_walker(s->d_chain_logp + slot0 * nl, false, s->bw_scratch_logp, (int64_t)nl, 1, 1, (int64_t)n, 0, (int64_t)nl, (int64_t)n, s->copy_stream));
            if (s->dst_logp_reg) {
                HIP_TRY(hipMemcpy2DAsync(s->dst_logp + (size_t)k0, (size_t)ns * sizeof(double), s->bw_scratch_logp, n * sizeof(double), n * sizeof(double), nl,
                                         hipMemcpyDeviceToHost, s->copy_stream));
            } else {
                if (s->bw_host.size() < n * nl) s->bw_host.resize(n * nl);
                HIP_TRY(copy_sync(s->bw_host.data(), s->bw_scratch_logp, n * nl * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
                for (size_t w = 0; w < nl; ++w)
                    std::memcpy(s->dst_logp + w * (size_t)ns + (size_t)k0, s->bw_host.data() + w * n, n * sizeof(double));
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

}  // namespace kmc_host
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
    if (s->dst_chain && !debug_opt("no-host-register")) {
        if (hipHostRegister(s->dst_chain, ns * nl * (size_t)s->cfg.ndim * sizeof(double), hipHostRegisterDefault) == hipSuccess) s->dst_chain_reg = true;
        else (void)hipGetLastError();
    }
    if (s->dst_logp && !debug_opt("no-host-register")) {
        if (hipHostRegister(s->dst_logp, ns * nl * sizeof(double), hipHostRegisterDefault) == hipSuccess) s->dst_logp_reg = true;
        else (void)hipGetLastError();
    }
    // by walker into arrays that could NOT be page-locked: the transposed blocks come through the bounce buffers and are put in
    // place by the host (chain_copy_range) -- slower (blocking), never refused
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
// buffer, a piece of walkers (<= ~256 MiB; KMC_DEBUG=by-walker-piece-mb=n for tests) at a time, each piece one contiguous copy.
kmc_status kmc_host::download_by_walker(const void* src, bool is_float, int64_t nl, int64_t ld, int64_t width, int64_t K, double* dst_host, hipStream_t st)
{
    if (K <= 0 || nl <= 0) return KMC_OK;
    const size_t per_walker = (size_t)K * (size_t)width * sizeof(double);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)256 << 20;
    { std::string mb; if (debug_opt("by-walker-piece-mb", &mb)) { const double v = std::atof(mb.c_str()); if (v > 0.0) budget = (size_t)(v * 1048576.0); } }
    if (budget > free_b / 2) budget = free_b / 2;
    int64_t wb = (int64_t)(budget / per_walker);
    if (wb < 1) wb = 1;
    if (wb > nl) wb = nl;
    double* tmp = nullptr;
    const bool stats = debug_opt("readout-stats");
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    HIP_TRY(cache_alloc((void**)&tmp, (size_t)wb * per_walker));
    const auto t1 = now();
    double t_kernel = 0.0, t_copy = 0.0;
    hipError_t e = hipSuccess;
    for (int64_t w0 = 0; w0 < nl && e == hipSuccess; w0 += wb) {
        const auto ta = now();
        const int64_t n = nl - w0 < wb ? nl - w0 : wb;
        e = launch_by_walker(src, is_float, tmp, nl, ld, width, K, w0, n, K * width, st);
        if (stats) { (void)hipStreamSynchronize(st); t_kernel += ms(ta, now()); }
        const auto tb = now();
        if (e == hipSuccess) e = copy_sync(dst_host + (size_t)w0 * (size_t)K * (size_t)width, tmp, (size_t)n * per_walker, hipMemcpyDeviceToHost, st);   // (waits: one scratch buffer)
        t_copy += ms(tb, now());
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(st);        // (every successful piece has waited already)
    const auto t2 = now();
    cache_free(tmp);
    if (stats)
        std::fprintf(stderr, "[kissmcmc_hip] by-walker read-out of %.0f MB: scratch allocation %.2f ms, transposition %.2f ms, device-to-host copy %.2f ms, scratch release %.2f ms\n",
                     (double)nl * (double)per_walker / 1e6, ms(t0, t1), t_kernel, t_copy, ms(t2, now()));
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

