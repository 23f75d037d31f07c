// kmc_p2p.hip -- multi-GPU wiring of a sampler (the join of reference src/samplers.jl:273 across GPUs):
// the RCCL communicator of the replica-sharded all-gather exchange, and the IPC handles / peer pointers of the
// peer-to-peer exchange (KMC_P2P).
#include <cstdio>

#include "kmc_sampler.hpp"
#include "kmc_kernels.hpp"      // load_row_sys: the link probe reads with the pull's own loads

using namespace kmc;
using namespace kmc_host;

namespace kmc_host {
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

}  // namespace kmc_host
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
    // a communicator on an unsharded sampler (shard_count 1: tests, a one-rank job): the all-gather follows every HALF-step, so the
    // sampler goes back to its two-launch kernels (moments credited so far are kept: unfuse)
    if (s->fused) KMC_TRY(unfuse(s));
    return rccl_comm_create(id, s->cfg.shard_rank, s->cfg.shard_count, &s->comm);
}

// Capture the hipGraph chunk (kernels + all-gathers) NOW and report whether this rank got one; kmc_sampler_rccl_set_capture(0)
// then makes a rank launch by launch although its own capture succeeded.  A driver calls the first on every rank, reduces
// the answers (MIN) and calls the second with the result, so that either every rank replays captured all-gathers or every
// rank enqueues them one by one -- not a mixture nobody has ever run (distributed.AllGatherEmcee).
KMC_EXPORT kmc_status kmc_sampler_rccl_capture(kmc_sampler* s, int* captured)
{
    if (!s || !captured) return fail(KMC_ERR_BAD_ARG, "null argument");
    *captured = 0;
    if (!s->comm) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_rccl_init has not been called");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (s->cfg.flags & KMC_NO_GRAPH) return KMC_OK;
    KMC_TRY(ensure_graph(s));
    *captured = s->graph_exec != nullptr ? 1 : 0;
    return KMC_OK;
}

KMC_EXPORT kmc_status kmc_sampler_rccl_set_capture(kmc_sampler* s, int use_captured)
{
    if (!s) return fail(KMC_ERR_BAD_ARG, "null argument");
    if (!s->comm) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_rccl_init has not been called");
    if (use_captured) {
        if (!s->graph_exec && !(s->cfg.flags & KMC_NO_GRAPH)) return fail(KMC_ERR_BAD_ARG, "this rank holds no captured chunk (kmc_sampler_rccl_capture)");
        return KMC_OK;
    }
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
    if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
    s->comm_graph_ok = false;
    return KMC_OK;
}

// librccl.so as this process resolves it: version code (ncclGetVersion: major * 10000 + minor * 100 + patch) and, optionally,
// its path -- KMC_ERR_UNSUPPORTED when the library or one of the entry points used here is missing.  No device needed.
KMC_EXPORT kmc_status kmc_rccl_version(int* version, char* path_buf, int64_t path_buflen)
{
    const char* path = nullptr;
    KMC_TRY(rccl_version(version, &path));
    if (path_buf && path_buflen > 0) std::snprintf(path_buf, (size_t)path_buflen, "%s", path ? path : "");
    return KMC_OK;
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
            o->nrows != s->nrows || o->ld != s->ld || o->push != s->push)
            return fail(KMC_ERR_BAD_ARG, "kmc_sampler_p2p_connect_local: shards[r] must be shard r of the same configuration on the same device");
        s->peer_pos[r] = o->d_pos;
        s->peer_flags[r] = o->d_flags;
    }
    s->connected = true;
    return KMC_OK;
}

// ---- one fabric link, measured (bench.py's `link-probe` rung; DESIGN.md section 7 prices the exact rule by bytes per link) ------------------
// The pull's access pattern: lane groups read WHOLE ROWS of the peer's shard at random row indices with the half-step kernels' own system-scope
// loads (load_row_sys), nothing else in the kernel -- the rate one xGMI link sustains for what `rand(ncos)` (src/samplers.jl:250) makes it serve.
namespace {
__global__ __launch_bounds__(256) void link_probe_gather(const double2* peer, uint32_t rows_peer, uint32_t chunks, int64_t nrows, uint32_t salt, double* sink)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = t / chunks;
    if (row >= nrows) return;
    uint32_t h = ((uint32_t)row + salt) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint64_t src = ((uint64_t)h * (uint64_t)rows_peer) >> 32;                  // uniform over the peer's rows, with replacement (:250)
    const double2 v = kmc::load_row_sys(peer + src * chunks + (uint32_t)(t % chunks));
    if (v.x == 1.2345e300 && v.y == -v.x) sink[0] = v.x;                             // (never: keeps the loads)
}
}  // namespace

KMC_EXPORT kmc_status kmc_sampler_p2p_link_probe(kmc_sampler* s, int peer, int64_t nrows, int reps, double* gather_gbs, double* copy_gbs)
{
    if (!s || !gather_gbs || !copy_gbs || nrows < 1 || reps < 1) return fail(KMC_ERR_BAD_ARG, "bad argument");
    if (!s->p2p || !s->connected || s->f32) return fail(KMC_ERR_BAD_ARG, "kmc_sampler_p2p_link_probe needs a connected KMC_P2P sampler with double rows");
    if (peer < 0 || peer >= s->cfg.shard_count) return fail(KMC_ERR_BAD_ARG, "no such shard");
    HIP_TRY(hipSetDevice(s->cfg.device));
    const double* src = peer == s->cfg.shard_rank ? s->d_pos : s->peer_pos[peer];    // (own rank: the same pattern on local memory, for comparison)
    const size_t shard_bytes = (size_t)s->nrows * (size_t)s->ld * sizeof(double);
    const uint32_t chunks = (uint32_t)(s->ld / 2);
    double* scratch = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    kmc_status st = KMC_OK;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && st == KMC_OK) st = fail(KMC_ERR_HIP, std::string("link probe: ") + hipGetErrorString(e)); return st == KMC_OK; };
    if (ok(hipMalloc(reinterpret_cast<void**>(&scratch), shard_bytes)) && ok(hipEventCreate(&e0)) && ok(hipEventCreate(&e1)) && ok(hipEventCreate(&e2))) {
        const unsigned grid = (unsigned)((nrows * chunks + 255) / 256);
        for (int r = -2; r < reps && st == KMC_OK; ++r) {                            // (two untimed launches first)
            if (r == 0) ok(hipEventRecord(e0, s->stream));
            hipLaunchKernelGGL(link_probe_gather, dim3(grid), dim3(256), 0, s->stream, reinterpret_cast<const double2*>(src), (uint32_t)s->nrows, chunks, nrows,
                               (uint32_t)(r + 7) * 40503u, scratch);
            ok(hipGetLastError());
        }
        ok(hipEventRecord(e1, s->stream));
        for (int r = 0; r < 4 && st == KMC_OK; ++r) ok(hipMemcpyAsync(scratch, src, shard_bytes, hipMemcpyDeviceToDevice, s->stream));    // the runtime's own copy of the whole shard
        ok(hipEventRecord(e2, s->stream));
        ok(hipEventSynchronize(e2));
        float tg = 0.f, tc = 0.f;
        if (ok(hipEventElapsedTime(&tg, e0, e1)) && ok(hipEventElapsedTime(&tc, e1, e2))) {
            *gather_gbs = (double)nrows * (double)s->ld * 8.0 * (double)reps / ((double)tg * 1e-3) / 1e9;
            *copy_gbs = (double)shard_bytes * 4.0 / ((double)tc * 1e-3) / 1e9;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (scratch) (void)hipFree(scratch);
    return st;
}

