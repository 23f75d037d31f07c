// kmc_launch.hip -- the `_emcee` generation loop (reference src/samplers.jl:232-293) as a stream of kernel launches: two dependent
// half-step launches per generation (the kernel boundary is the join of src/samplers.jl:273) -- or, for small states, one launch per
// generation (kmc_generation.hpp) --, replayed from a hipGraph in chunks of kGraphChunk generations -- table-driven, or with per-replay
// parameter updates -- or issued eagerly; the calibration that picks between them and the process-wide budget of the updated-graph mode.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define KMC_DEFINE_LAUNCH_KERNELS
#include "kmc_sampler.hpp"

using namespace kmc;
using namespace kmc_host;

namespace kmc_host {
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
    KMC_LK(64, 8) KMC_LK(16, 8)      // (16 x 8: function bodies evaluated per walker on rows of 129 ... 256 doubles, kmc_plan.hip)
#undef KMC_LK
    return nullptr;
}

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
    a.nranks = pack_ranks(s->cfg.shard_count, s->cfg.shard_rank, s->push);
    for (int r = 0; r < 8; ++r) a.peer_pos[r] = s->peer_pos[r];
    a.flags = s->d_flags;
    a.err = s->d_err;
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
    a.blob = s->d_blob;
    a.chain_blob = s->d_chain_blob;
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
// ragged_vec: for half_step_vec<..., RAGGED = true> (and only for it: kmc_sampler::ragged_vec()) ndim (and with it the row stride) rides in the 16 bits above the log-pdf block's address
HalfStepFront front_of(const HalfStepArgs& a, bool ragged_vec)
{
    HalfStepFront f{};
    f.pos = a.pos;
    f.sched = a.sched_index < 0 ? nullptr : a.sched_table + a.sched_index;
    f.step = (uint32_t)(2ull * (uint64_t)a.sched_inline.gen + (uint64_t)a.half);       // used when sched == nullptr
    f.gw0 = (uint32_t)a.gw0;
    f.logp = ragged_vec ? reinterpret_cast<double*>(reinterpret_cast<uint64_t>(a.logp) | ((uint64_t)a.ndim << 48)) : a.logp;
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
    if (s->p2p && s->cfg.shard_count > 1) {
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
    const HalfStepFront f = front_of(a, s->ragged_vec());
    if (s->user) {
        const HalfStepLaunch la{f, a};
        if (s->uk.staged) return launch_module(s->uk.staged, (unsigned)s->grid, (unsigned)s->tpb, s->stream, la, (unsigned)staged_lds_bytes((int)s->cfg.ndim));
        return launch_module(s->plan.vec ? s->uk.vec : s->uk.generic, (unsigned)s->grid, (unsigned)s->tpb, s->stream, la, s->plan.vec ? s->vec_lds : 0u);
    }
    if (debug_opt("menu-via-module")) {
        // experiment: the compiled-in kernel launched the way runtime-compiled ones are (hipFunction_t + argument buffer) -- is it the
        // launch path or the code object that makes those 7-19 % slower?  (profiles/NOTES.md round 4)
        hipFunction_t fn = nullptr;
        const hipError_t e = hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(s->plan.fn));
        if (e != hipSuccess) return e;
        const HalfStepLaunch la{f, a};
        return launch_module(fn, (unsigned)s->grid, (unsigned)s->tpb, s->stream, la);
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

// ---- one launch per generation (kmc_generation.hpp) ---------------------------------------------------------
// `from`: which copy of the state the generation reads (0: d_pos / d_logp, 1: d_pos2 / d_logp2); it writes the other.
GenerationArgs make_generation_args(const kmc_sampler* s, int from, bool graph_mode, int64_t gen_offset)
{
    GenerationArgs a{};
    double* const pos[2] = {s->d_pos, s->d_pos2};
    double* const lp[2] = {s->d_logp, s->d_logp2};
    a.pin = pos[from]; a.pout = pos[1 - from];
    a.lin = lp[from]; a.lout = lp[1 - from];
    a.naccept = s->d_naccept;
    a.sched = graph_mode ? s->d_sched + gen_offset : nullptr;
    a.sched_inline = make_sched(gen_offset, s->cfg.nburnin, s->cfg.nthin, s->nsamples, s->ring_slots);
    a.dc = make_args(s, 0, false, 0).dc;
    a.dp = s->dp;
    a.h = (uint32_t)s->h;
    const int64_t per_wg = s->fused_L > 0 ? s->fused_tpb / s->fused_L : s->fused_tpb;       // walkers per workgroup
    a.nb = (uint32_t)((s->h + per_wg - 1) / per_wg);
    a.ld = (int32_t)s->ld;
    a.ndim = (int32_t)s->cfg.ndim;
    a.chain = s->d_chain;
    a.chain_logp = s->d_chain_logp;
    a.msum = s->d_isum;
    a.msumsq = s->d_isumsq;
    a.klast = s->d_klast;
    a.glast = s->d_glast;
    return a;
}

static_assert(offsetof(GenerationLaunch, a) == 56 && offsetof(GenerationFront, gen) == 52, "kernarg layout of the generation kernels");

// the head of the chain among the preloaded kernel parameters (GenerationFront)
GenerationFront generation_front_of(const GenerationArgs& a, int tpb)
{
    return GenerationFront{a.sched, a.pin, a.lin, a.pout, a.dc.seed_lo, a.dc.seed_hi, a.h, a.nb | ((uint32_t)(tpb / 64 - 1) << 30), a.ld | (a.ndim << 16), (uint32_t)a.sched_inline.gen};
}

hipError_t launch_generation(const kmc_sampler* s, int from, bool graph_mode, int64_t gen_offset)
{
    const GenerationArgs a = make_generation_args(s, from, graph_mode, gen_offset);
    const GenerationFront f = generation_front_of(a, s->fused_tpb);
    const unsigned tpb = (unsigned)s->fused_tpb;
    if (s->user) {
        const GenerationLaunch la{f, a};
        return launch_module(s->uk.generation, 2u * a.nb, tpb, s->stream, la);
    }
    hipLaunchKernelGGL(s->generation_kernel, dim3(2u * a.nb), dim3(tpb), 0, s->stream, f.sched, f.pin, f.lin, f.pout, f.seed_lo, f.seed_hi, f.h, f.nb, f.ld, f.gen, a);
    return hipGetLastError();
}

// the state back into the sampler's canonical arrays (after an odd number of generations)
hipError_t generation_settle(kmc_sampler* s)
{
    if (s->fused_cur == 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(s->d_pos, s->d_pos2, (size_t)s->nrows * (size_t)s->ld * sizeof(double), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_logp, s->d_logp2, (size_t)s->nrows * sizeof(double), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) s->fused_cur = 0;
    return e;
}

// A sampler that runs one launch per generation goes back to its two-launch kernels, in place and for good (everything they need was set up
// at creation): for a caller that steps by halves (kmc_sampler_half_step) or attaches an RCCL communicator (an all-gather follows every
// half-step).  The moments credited so far -- every walker's current value included, up to now -- are read out once and carried on the host
// (added at every later read-out); the walkers' current values stand for the samples from here on (klast = samples taken).
void drop_updated_graph(kmc_sampler* s)
{
    for (int i = 0; i < kUExec; ++i) {
        if (s->uexec[i]) { (void)hipGraphExecDestroy(s->uexec[i]); s->uexec[i] = nullptr; }
        if (s->udone[i]) { (void)hipEventDestroy(s->udone[i]); s->udone[i] = nullptr; }      // (ensure_updated_graph creates them anew)
        s->uinflight[i] = false;
    }
    s->unext = 0;
    if (s->ugraph) { (void)hipGraphDestroy(s->ugraph); s->ugraph = nullptr; }
}

kmc_status unfuse(kmc_sampler* s)
{
    if (!s->fused) return KMC_OK;
    HIP_TRY(generation_settle(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->d_msum && s->d_isum && s->generation > 0) {
        std::vector<double> S((size_t)s->cfg.ndim), Q((size_t)s->cfg.ndim);
        int64_t n = 0;
        KMC_TRY(kmc_sampler_get_moments(s, S.data(), Q.data(), &n));         // (still the fused read-out: sums + the walkers' current values)
        if (s->carry_sum.empty()) { s->carry_sum.assign(S.size(), 0.0); s->carry_sumsq.assign(Q.size(), 0.0); }
        for (size_t d = 0; d < S.size(); ++d) { s->carry_sum[d] += S[d]; s->carry_sumsq[d] += Q[d]; }
        if (s->d_klast) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)s->d_klast, (int)(uint32_t)samples_done(s), (size_t)s->nrows, s->stream));
    }
    if (s->graph_exec) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
    if (s->graph) { (void)hipGraphDestroy(s->graph); s->graph = nullptr; }
    drop_updated_graph(s);                                  // (its nodes are generation kernels)
    s->fused = false;
    s->launch_mode = 0;
    s->calib_graph_ms = s->calib_eager_ms = 0.f;
    return KMC_OK;
}

kmc_status ensure_graph(kmc_sampler* s)
{
    if (s->graph_exec) return KMC_OK;
    if (s->comm && !s->comm_graph_ok) return KMC_OK;      // RCCL refused the capture once (or a peer's did): launch by launch, for good
    // A caller's stream may be the legacy default stream (torch's current stream usually is): that one cannot be captured.  The capture
    // then runs on a stream of its own -- a graph does not remember the stream it was recorded on; the replays go to the caller's.
    hipStream_t const users = s->stream;
    hipStream_t tmp = nullptr;
    if (users == nullptr || users == hipStreamPerThread) {
        HIP_TRY(hipStreamCreateWithFlags(&tmp, hipStreamNonBlocking));
        s->stream = tmp;
    }
    struct Restore {
        kmc_sampler* s; hipStream_t users, tmp;
        ~Restore() { s->stream = users; if (tmp) (void)hipStreamDestroy(tmp); }
    } restore{s, users, tmp};
    HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
    kmc_status st = KMC_OK;
    launch_advance(s, (int)kGraphChunk, 0);                 // schedule table of this chunk
    static_assert(kGraphChunk % 2 == 0, "a chunk of fused generations ends in the copy of the state it started from");
    for (int64_t g = 0; g < kGraphChunk && st == KMC_OK; ++g) {
        if (s->fused) { if (launch_generation(s, (int)(g & 1), true, g) != hipSuccess) st = KMC_ERR_HIP; continue; }
        for (int half = 0; half < 2 && st == KMC_OK; ++half) st = launch_half(s, half, true, g);
    }
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

struct GenerationParamPack {  // the same for a generation kernel's node (one launch per generation)
    GenerationFront f;
    GenerationArgs a;
    void* ptrs[11];
    void bind()
    {
        ptrs[0] = &f.sched; ptrs[1] = &f.pin; ptrs[2] = &f.lin; ptrs[3] = &f.pout; ptrs[4] = &f.seed_lo; ptrs[5] = &f.seed_hi;
        ptrs[6] = &f.h; ptrs[7] = &f.nb; ptrs[8] = &f.ld; ptrs[9] = &f.gen; ptrs[10] = &a;
    }
};

hipKernelNodeParams generation_node_params(const kmc_sampler* s, GenerationParamPack* pk)
{
    hipKernelNodeParams np{};
    np.func = s->user ? reinterpret_cast<void*>(s->uk.generation) : reinterpret_cast<void*>(s->generation_kernel);
    np.gridDim = dim3(2u * pk->a.nb);
    np.blockDim = dim3((unsigned)s->fused_tpb);
    np.sharedMemBytes = 0u;
    np.kernelParams = pk->ptrs;
    np.extra = nullptr;
    return np;
}

hipKernelNodeParams node_params(const kmc_sampler* s, KernelParamPack* pk)
{
    hipKernelNodeParams np{};
    // (a runtime-compiled density's lane-striped kernel is a module function: the runtime accepts its hipFunction_t as the node's
    //  function -- ensure_updated_graph tries it once and the sampler stays with the table graph if it does not)
    np.func = s->user ? reinterpret_cast<void*>(s->uk.vec) : reinterpret_cast<void*>(s->plan.fn);
    np.gridDim = dim3((unsigned)s->grid);
    np.blockDim = dim3((unsigned)s->tpb);
    np.sharedMemBytes = s->user ? s->vec_lds : 0u;
    np.kernelParams = pk->ptrs;
    np.extra = nullptr;
    return np;
}

bool updated_graph_possible(const kmc_sampler* s)
{
    if (s->p2p || s->host_eval || s->islands || s->resident || s->comm || s->updated_refused) return false;
    if (s->fused) return s->user ? s->uk.generation != nullptr : s->generation_kernel != nullptr;      // (one node per generation: the generation among the preloaded parameters)
    return s->user ? (s->plan.vec && s->uk.vec != nullptr) : s->plan.fn != nullptr;
}

kmc_status ensure_updated_graph(kmc_sampler* s)
{
    if (s->uexec[0]) return KMC_OK;
    HIP_TRY(hipGraphCreate(&s->ugraph, 0));
    s->unodes.assign((size_t)((s->fused ? 1 : 2) * s->uchunk), nullptr);
    hipGraphNode_t prev = nullptr;
    if (s->fused) {
        // one node per generation, reading copy g & 1 of the state and writing the other (uchunk is even: a replay ends in the copy it started from)
        GenerationParamPack gp;
        gp.bind();
        for (int64_t g = 0; g < s->uchunk; ++g) {
            gp.a = make_generation_args(s, (int)(g & 1), false, g);
            gp.f = generation_front_of(gp.a, s->fused_tpb);
            const hipKernelNodeParams np = generation_node_params(s, &gp);
            hipGraphNode_t node = nullptr;
            HIP_TRY(hipGraphAddKernelNode(&node, s->ugraph, prev ? &prev : nullptr, prev ? 1 : 0, &np));
            s->unodes[(size_t)g] = node;
            prev = node;
        }
    }
    KernelParamPack pk;
    pk.bind();
    for (int64_t g = 0; g < s->uchunk && !s->fused; ++g)
        for (int half = 0; half < 2; ++half) {
            pk.a = make_args(s, half, false, g);
            pk.f = front_of(pk.a, s->ragged_vec());
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

// hipGraphExecKernelNodeSetParams keeps host memory inside the runtime, per call and for good (kmc_sampler_run: launch modes): ~80 bytes in the HIP 7.0 runtime (the one
// PyTorch's wheel bundles and a Python process therefore runs on), ~1.4 bytes in 7.2 (/opt/rocm of this image; scripts/probes/updated_graph_rss.py, profiles/r05_updated_graph_rss.txt).
// A process-wide budget of such calls -- 64 MiB worth by default, priced by the runtime this process actually loaded.
double update_bytes_each()
{
    static const double b = [] {
        int v = 0;
        return (hipRuntimeGetVersion(&v) == hipSuccess && v >= 70200000) ? 2.0 : 80.0;
    }();
    return b;
}
std::atomic<int64_t> g_update_calls{0};
std::atomic<int64_t>& update_budget()
{
    static std::atomic<int64_t> budget{[] {
        double mb = 64.0;
        std::string v;
        if (debug_opt("updated-budget-mb", &v) && !v.empty()) mb = std::atof(v.c_str());
        return (int64_t)(mb * 1048576.0 / update_bytes_each());
    }()};
    return budget;
}
// room for `need` more parameter updates (a replay of this sampler: 2 * uchunk)?  A replay that would overshoot the budget is not started.
bool update_budget_left(int64_t need) { return g_update_calls.load(std::memory_order_relaxed) + need <= update_budget().load(std::memory_order_relaxed); }

void note_budget_spent(kmc_sampler* s)
{
    s->budget_fallback = true;
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
        std::fprintf(stderr, "[kissmcmc_hip] the updated-graph launch mode has used up this process's budget (%lld parameter updates, "
                             "kmc_set_updated_budget_mb / KMC_DEBUG=updated-budget-mb=n; this HIP runtime keeps ~%.0f B per update): samplers now "
                             "choose between the table graph and eager launches (up to ~9 %% slower per half-step; results are identical)\n",
                     (long long)g_update_calls.load(std::memory_order_relaxed), update_bytes_each());
}

// one replay of s->uchunk generations starting at s->generation.  *launched tells a failure BEFORE the replay was
// enqueued (nothing ran: the caller may issue these generations another way) from one after it (they are running).
kmc_status launch_updated_graph(kmc_sampler* s, bool* launched)
{
    *launched = false;
    KMC_TRY(ensure_updated_graph(s));
    g_update_calls.fetch_add((s->fused ? 1 : 2) * s->uchunk, std::memory_order_relaxed);
    const int i = s->unext;
    const auto t_wait0 = std::chrono::steady_clock::now();
    if (s->uinflight[i]) { HIP_TRY(hipEventSynchronize(s->udone[i])); s->uinflight[i] = false; }
    const auto t_upd0 = std::chrono::steady_clock::now();
    if (s->fused) {
        GenerationParamPack gp;
        gp.bind();
        for (int64_t g = 0; g < s->uchunk; ++g) {
            gp.a = make_generation_args(s, (int)(g & 1), false, s->generation + g);
            gp.f = generation_front_of(gp.a, s->fused_tpb);
            const hipKernelNodeParams np = generation_node_params(s, &gp);
            HIP_TRY(hipGraphExecKernelNodeSetParams(s->uexec[i], s->unodes[(size_t)g], &np));
        }
    }
    KernelParamPack pk;
    pk.bind();
    for (int64_t g = 0; g < s->uchunk && !s->fused; ++g)
        for (int half = 0; half < 2; ++half) {
            pk.a = make_args(s, half, false, s->generation + g);
            pk.f = front_of(pk.a, s->ragged_vec());
            const hipKernelNodeParams np = node_params(s, &pk);
            HIP_TRY(hipGraphExecKernelNodeSetParams(s->uexec[i], s->unodes[(size_t)(2 * g + half)], &np));
        }
    const auto t_upd1 = std::chrono::steady_clock::now();
    HIP_TRY(hipGraphLaunch(s->uexec[i], s->stream));
    const auto t_launch1 = std::chrono::steady_clock::now();
    // where the feeding thread's time goes (kmc_sampler_describe with KMC_DEBUG=feed-stats): waiting for a free executable = the GPU is
    // the bottleneck; updating + launching = the host's own cost per replay, which must stay below the replay's GPU time
    s->feed_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t_upd0 - t_wait0).count();
    s->feed_update_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t_upd1 - t_upd0).count();
    s->feed_launch_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t_launch1 - t_upd1).count();
    s->feed_replays += 1;
    *launched = true;
    HIP_TRY(hipEventRecord(s->udone[i], s->stream));
    s->uinflight[i] = true;
    s->unext = (i + 1) % kUExec;
    return KMC_OK;
}

// Streaming moments of the multi-launch kernels are sojourn-weighted (a walker's value is credited when it is replaced):
// credit every walker's CURRENT value with the samples it has stood for so far (enqueued on the sampler's stream; after
// it every klast equals the number of samples taken).  Before a read-out, and before walkers change slots (deal).
kmc_status flush_moments_now(kmc_sampler* s)
{
    if (!s->d_msum || s->islands || s->resident || s->fused) return KMC_OK;
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

}  // namespace kmc_host
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
        // the whole ensemble lives in one workgroup's LDS; a launch carries up to 1024 generations, whose draws a wide kernel
        // computes first (draw_table_fill, kmc_islands.hpp)
        while (ngen > 0) {
            int64_t n = std::min<int64_t>(ngen, kDrawTableGens);
            if (s->stream_chain) n = std::min<int64_t>(n, std::max<int64_t>(1, (s->ring_blk - 1) * s->cfg.nthin));   // a launch fills less than a block of the ring
            KMC_TRY(chain_before(s, s->generation + n));
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
            ra.blob = s->d_blob; ra.chain_blob = s->d_chain_blob;
            ra.ring_slots = s->stream_chain ? s->ring_slots : 0;
            {                                    // (d_draws: allocated with every resident sampler)
                DrawTableArgs ta{};
                const int64_t npad = (n + kDrawBatch - 1) / kDrawBatch * kDrawBatch;      // whole batches (<= kDrawTableGens, a multiple)
                ta.dc = ia.dc; ta.gen0 = s->generation; ta.ngen = (int32_t)npad; ta.S = ra.S; ta.out = s->d_draws;
                hipLaunchKernelGGL(draw_table_fill, dim3((unsigned)((npad * ra.S + 255) / 256)), dim3(256), 0, s->stream, ta);
                HIP_TRY(hipGetLastError());
                ra.draws = s->d_draws;
                s->launches += 1;
            }
            if (s->user) {
                HIP_TRY(launch_module(s->uk.resident, 1u, (unsigned)s->resident_tpb, s->stream, ra, (unsigned)s->island_lds));
            } else {
                hipLaunchKernelGGL(s->resident_kernel, dim3(1), dim3((unsigned)s->resident_tpb), s->island_lds, s->stream, ra);
                HIP_TRY(hipGetLastError());
            }
            s->generation += n;
            s->launches += 1;
            ngen -= n;
            KMC_TRY(chain_after(s));
        }
        HIP_TRY(hipEventRecord(s->ev1, s->stream));
        s->have_run_events = true;
        return KMC_OK;
    }
    if (s->fused && s->fused_L > 0 && ngen > 0 && !(s->pos2_current && s->own_pos && !s->pos_exposed)) {
        // One launch per generation reads one copy of the state and writes the other.  The lane-striped form writes a row to the output copy only when that
        // copy does not hold it already: the copies are made equal whenever somebody may have changed d_pos / d_logp since the last run (set_positions,
        // set_state, the initial ball: they write the first pair only; a caller-owned buffer -- kmc_sampler_bind_positions -- may change at any time: every
        // run); between runs the kernels keep them consistent.  (Launch modes: below, with the two-launch kernels' -- whole chunks start in copy 0.)
        HIP_TRY(generation_settle(s));         // (a run that failed after an odd number of generations left the newest state in copy 1: bring it home before copy 0 overwrites it)
        HIP_TRY(hipMemcpyAsync(s->d_pos2, s->d_pos, (size_t)s->nrows * (size_t)s->ld * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_logp2, s->d_logp, (size_t)s->nrows * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        s->pos2_current = true;
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
        // Large batches travel in pieces, each behind its own event: the callback works on piece c while piece c + 1 is still
        // on the link (the D2H of the proposals is the larger half of a half-step's time once the closure is cheap).  Not when
        // accept outcomes go back to the host as well: that caller keeps per-batch state (blobs) between the two callbacks.
        int npiece = (!s->cfg.host_accepted && hh >= 8192) ? 4 : 1;
        { const long v = debug_opt_long("host-pieces", 0); if (v >= 1 && v <= kHostPieces && !s->cfg.host_accepted) npiece = (int)v; }
        for (int i = 0; i < npiece && npiece > 1; ++i)
            if (!s->host_ev[i]) HIP_TRY(hipEventCreateWithFlags(&s->host_ev[i], hipEventDisableTiming));
        // Small batches (the reference's own sizes) skip the copies: the propose kernel writes the dense proposal rows straight into
        // the page-locked host array over the link and the accept kernel reads the log-pdfs straight from theirs -- a half-step is
        // then two launches, one synchronisation and the callback (100 walkers: 32 -> ~15 us with a Python callable).
        // KMC_DEBUG=host-zerocopy=0|1 forces it off / on.
        bool zero_copy = npiece == 1 && s->h_prop_dev != nullptr && hh * nd * sizeof(double) <= ((size_t)256 << 10);
        { std::string e; if (debug_opt("host-zerocopy", &e)) zero_copy = e == "1" && npiece == 1 && s->h_prop_dev != nullptr; }
        for (; ngen > 0; --ngen) {
            KMC_TRY(chain_before(s, s->generation + 1));
            for (int half = 0; half < 2; ++half) {
                HalfStepArgs a = make_args(s, half, false, s->generation);
                a.prop_out = zero_copy ? s->h_prop_dev : s->d_prop;
                a.prop_ld = zero_copy ? (int32_t)nd : (int32_t)ld;
                HIP_TRY(launch_half_kernel(s, a));
                bool cb_failed = false;
                if (zero_copy) {
                    HIP_TRY(hipStreamSynchronize(s->stream));
                    cb_failed = s->cfg.host_logpdf(s->h_prop, (int64_t)hh, (int64_t)nd, s->h_p1, s->cfg.host_user) != 0;   // :257
                } else if (npiece == 1) {
                    HIP_TRY(hipMemcpy2DAsync(s->h_prop, nd * sizeof(double), s->d_prop, ld * sizeof(double), nd * sizeof(double), hh,
                                             hipMemcpyDeviceToHost, s->stream));
                    HIP_TRY(hipStreamSynchronize(s->stream));
                    cb_failed = s->cfg.host_logpdf(s->h_prop, (int64_t)hh, (int64_t)nd, s->h_p1, s->cfg.host_user) != 0;   // :257
                } else {
                    auto r0 = [&](int c) { return hh * (size_t)c / (size_t)npiece; };
                    for (int c = 0; c < npiece; ++c) {
                        HIP_TRY(hipMemcpy2DAsync(s->h_prop + r0(c) * nd, nd * sizeof(double), s->d_prop + r0(c) * ld, ld * sizeof(double), nd * sizeof(double),
                                                 r0(c + 1) - r0(c), hipMemcpyDeviceToHost, s->stream));
                        HIP_TRY(hipEventRecord(s->host_ev[c], s->stream));
                    }
                    for (int c = 0; c < npiece; ++c) {
                        HIP_TRY(hipEventSynchronize(s->host_ev[c]));
                        if (!cb_failed)
                            cb_failed = s->cfg.host_logpdf(s->h_prop + r0(c) * nd, (int64_t)(r0(c + 1) - r0(c)), (int64_t)nd, s->h_p1 + r0(c), s->cfg.host_user) != 0;   // :257
                    }
                }
                if (cb_failed) {
                    s->positions_set = false;
                    return fail(KMC_ERR_BAD_ARG, "the host log-pdf callback failed in generation " + std::to_string(s->generation));
                }
                if (!zero_copy) HIP_TRY(hipMemcpyAsync(s->d_p1, s->h_p1, hh * sizeof(double), hipMemcpyHostToDevice, s->stream));
                a.prop_out = nullptr;
                a.p1_in = zero_copy ? s->h_p1_dev : s->d_p1;
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
    const int64_t lpg = s->fused ? 1 : 2;                    // launches per generation
    auto eager_generations = [&](int64_t n) -> kmc_status {
        for (; n > 0; --n, --ngen) {
            KMC_TRY(chain_before(s, s->generation + 1));
            if (s->fused) {
                HIP_TRY(launch_generation(s, s->fused_cur, false, s->generation));
                s->fused_cur ^= 1;
            } else {
                for (int half = 0; half < 2; ++half) KMC_TRY(launch_half(s, half, false, s->generation));
            }
            s->generation += 1;
            s->launches += lpg;
            if (!s->fused && ++s->gens_since_sweep >= kSweepEvery) HIP_TRY(launch_sweep(s));
            KMC_TRY(chain_after(s));
        }
        return KMC_OK;
    };
    auto graph_chunk = [&]() -> kmc_status {
        if (s->fused) HIP_TRY(generation_settle(s));                    // (a chunk of generation kernels starts in copy 0 and ends there)
        KMC_TRY(ensure_graph(s));
        if (!s->graph_exec) return eager_generations(kGraphChunk);      // (RCCL all-gather that cannot be captured)
        KMC_TRY(sync_device_counter(s));
        KMC_TRY(chain_before(s, s->generation + kGraphChunk));
        HIP_TRY(hipGraphLaunch(s->graph_exec, s->stream));
        if (!s->fused) HIP_TRY(launch_sweep(s));
        s->generation += kGraphChunk;
        s->dev_gen += kGraphChunk;
        s->launches += lpg * kGraphChunk;
        ngen -= kGraphChunk;
        return chain_after(s);
    };
    auto updated_chunk = [&]() -> kmc_status {
        if (s->fused) HIP_TRY(generation_settle(s));
        KMC_TRY(chain_before(s, s->generation + s->uchunk));
        bool launched = false;
        if (launch_updated_graph(s, &launched) != KMC_OK) {
            // this sampler uses the table graph from here on (the loops below go on with graph_chunk while a whole chunk is
            // left, eagerly after that: exactly the generations asked for).  Failed before the replay was enqueued: nothing
            // ran, nothing to account.  Failed after it (the event of this executable): the replay is running -- account
            // it; without its event the executable is not reused, which leaving this mode guarantees.
            s->launch_mode = 1;
            if (debug_opt("feed-stats")) std::fprintf(stderr, "[kissmcmc_hip] the updated-graph replay failed (%s): table graph from here on\n", kmc_last_error());
            if (!launched) return KMC_OK;
            HIP_TRY(hipStreamSynchronize(s->stream));
        }
        if (!s->fused) HIP_TRY(launch_sweep(s));
        s->generation += s->uchunk;
        s->launches += lpg * s->uchunk;
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
    //                      every replay (C2 4.00 us, C3 3.25 us; steady).  But hipGraphExecKernelNodeSetParams keeps host
    //                      memory per call inside the runtime -- ~80 bytes in HIP 7.0 (1.6 MB per 10^4 generations, not returned
    //                      when the executables are destroyed), ~1.4 in 7.2: update_bytes_each above -- so the process has a BUDGET
    //                      of such calls (kmc_set_updated_budget_mb): beyond it samplers choose
    //                      between 1 and 2.
    // A long run (>= 896 generations) measures 1 against 3 (or 2) once -- 256 generations each, HIP events: a starved GPU shows as idle time
    // between the events -- and keeps the faster; KMC_LAUNCH=graph|eager|updated decides without measuring.
    bool use_graph = !(s->cfg.flags & KMC_NO_GRAPH);
    if (s->comm && !s->comm_graph_ok) use_graph = false;    // (decided at the first capture, kmc_sampler_rccl_capture)
    const int64_t calib_gens = std::max<int64_t>(4 * kGraphChunk, 2 * s->uchunk);                               // generations measured in each mode
    const int64_t calib_min = kGraphChunk + 2 * std::max<int64_t>(kGraphChunk, s->uchunk) + 2 * calib_gens + kGraphChunk;    // warm-up pieces + both measurements + a chunk to spare (896)
    auto calibrate = [&](bool with_updated) -> kmc_status {
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventCreate(&e2));
        const int64_t gens = calib_gens;
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
    // The measurement and the updated graph's set-up (six executables of 128 or 256 nodes) cost about a millisecond once: worth it for a job planned long
    // (kmc_config::ngenerations) or a sampler that has come this far anyway, not for a one-shot run of a few thousand generations (4 096 x 4, one run of 1 024
    // generations: 4.3 ms with the measurement, 3.4 ms from the table graph alone; break-even ~2 000 generations at C2, ~10 000 for a short-row ensemble)
    constexpr int64_t kWorthMeasuring = 4096;
    const bool long_job = s->cfg.ngenerations >= kWorthMeasuring || s->generation + ngen >= kWorthMeasuring;
    const char* const launch_env = std::getenv("KMC_LAUNCH");
    const bool updated_asked = launch_env && (std::strcmp(launch_env, "updated") == 0 || std::strcmp(launch_env, "updated,budget") == 0);
    if (use_graph && s->launch_mode == 0 && s->user && updated_graph_possible(s) && !s->uexec[0] && (updated_asked || (long_job && ngen >= calib_min))) {
        // module functions as graph kernel nodes: build the graph now; a runtime that refuses leaves this sampler with the table graph
        if (ensure_updated_graph(s) != KMC_OK) {
            (void)hipGetLastError();
            drop_updated_graph(s);
            s->updated_refused = true;
        }
    }
    if (use_graph && s->launch_mode == 0) {
        const char* env = std::getenv("KMC_LAUNCH");
        if (env && std::strcmp(env, "graph") == 0) s->launch_mode = 1;
        else if (env && std::strcmp(env, "eager") == 0) s->launch_mode = 2;
        else if (env && std::strcmp(env, "updated") == 0 && updated_graph_possible(s)) { s->launch_mode = 3; s->updated_forced = true; }
        else if (env && std::strcmp(env, "updated,budget") == 0 && updated_graph_possible(s) && update_budget_left(lpg * s->uchunk)) s->launch_mode = 3;   // (as if measured: the budget applies)
        else if (!updated_graph_possible(s)) s->launch_mode = 1;
        else if (ngen >= calib_min && long_job) {
            const bool left = update_budget_left(lpg * s->uchunk);      // (the measurement itself may overshoot by its six replays; the run after it may not)
            if (!left) note_budget_spent(s);
            KMC_TRY(calibrate(left));
        }
    }
    while (use_graph && s->launch_mode == 3 && ngen >= s->uchunk) {
        if (!s->updated_forced && !update_budget_left(lpg * s->uchunk)) {             // the process has used up its leak budget: decide again, between 1 and 2
            note_budget_spent(s);
            s->launch_mode = 0;
            if (ngen >= calib_min) KMC_TRY(calibrate(false));
            break;
        }
        KMC_TRY(updated_chunk());
    }
    if (s->launch_mode == 2) use_graph = false;
    while (use_graph && ngen >= kGraphChunk) KMC_TRY(graph_chunk());
    KMC_TRY(eager_generations(ngen));
    if (s->fused) HIP_TRY(generation_settle(s));                        // the state back in d_pos / d_logp (after an odd number of generations)
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
    HIP_TRY(hipSetDevice(s->cfg.device));             // (before unfuse: its copies, graph destruction and moment read-out belong to this sampler's device)
    if (s->fused) KMC_TRY(unfuse(s));                 // stepping by halves: the two-launch kernels from here on (same chain, bit for bit)
    if (s->generation >= ((int64_t)1 << 31) - 1) return fail(KMC_ERR_UNSUPPORTED, "at most 2^31 - 1 generations (the step index is 32 bits)");
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

// How kmc_sampler_run issues this sampler's launches: KMC_LAUNCH_UNDECIDED (short runs so far: table graph for whole chunks),
// _TABLE_GRAPH, _EAGER, _UPDATED_GRAPH; _SINGLE for the modes that are one launch per many generations (resident, islands)
// and the host-evaluated density.  *budget_fallback (may be NULL): 1 when the sampler is not in the updated-graph mode
// because the process-wide budget of parameter updates was spent.
KMC_EXPORT int kmc_sampler_launch_mode(const kmc_sampler* s, int* budget_fallback)
{
    if (!s) return -1;
    if (budget_fallback) *budget_fallback = s->budget_fallback ? 1 : 0;
    if (s->resident || s->islands || s->host_eval) return KMC_LAUNCH_SINGLE;
    if ((s->cfg.flags & KMC_NO_GRAPH) || (s->comm && !s->comm_graph_ok)) return KMC_LAUNCH_EAGER;
    return s->launch_mode;
}

KMC_EXPORT void kmc_updated_budget(int64_t* calls_used, int64_t* calls_budget)
{
    if (calls_used) *calls_used = g_update_calls.load(std::memory_order_relaxed);
    if (calls_budget) *calls_budget = update_budget().load(std::memory_order_relaxed);
}

KMC_EXPORT void kmc_set_updated_budget_mb(double mb)
{
    update_budget().store(mb <= 0.0 ? 0 : (int64_t)(mb * 1048576.0 / update_bytes_each()), std::memory_order_relaxed);
}

