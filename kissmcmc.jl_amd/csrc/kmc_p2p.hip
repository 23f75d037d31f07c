// kmc_p2p.hip -- multi-GPU wiring of a sampler (the join of reference src/samplers.jl:273 across GPUs):
// the RCCL communicator of the replica-sharded all-gather exchange, and the IPC handles / peer pointers of the
// peer-to-peer exchange (KMC_P2P).
#include <cstdio>

#include "kmc_sampler.hpp"

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
