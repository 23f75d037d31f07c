// kmc_sampler.hpp -- what the sampler's translation units share (kmc_sampler.hip: lifecycle; kmc_launch.hip: the generation
// loop; kmc_state.hip: state in and out; kmc_copy.hip: copies and the chain ring; kmc_rtc.hip: runtime-compiled densities;
// kmc_p2p.hip: multi-GPU wiring; kmc_diag.hip: diagnostics): the handle behind kmc_sampler* and the internal entry points
// that cross those files.  Internal.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "kmc_host.hpp"

KMC_EXPORT int kmc_version(void);

namespace kmc_host {

static_assert(1024 % kmc::kDrawBatch == 0, "the draw table holds whole batches");
constexpr int64_t kDrawTableGens = 1024;   // resident mode with a draw table: generations per launch (table = 32 B x nwalkers x this)
constexpr int64_t kGraphChunk = 64;   // generations per hipGraph replay (128 kernel nodes + 1)
constexpr int kUExec = 6;             // executables of the "updated graph" launch mode (kmc_sampler::uexec)
constexpr size_t kGuardBytes = 4096;  // KMC_DEBUG=poison: guard band behind every device allocation of a sampler
constexpr int kHostPieces = 8;        // KMC_HOST_DENSITY: most pieces a half-step's proposals travel to the host in

struct Plan {
    kmc::HalfStepFn fn = nullptr;
    bool vec = false;
    bool ragged = false;
    int L = 1, K = 1, ITER = 1;
};

// kernels of a runtime-compiled density, loaded from its code object (kmc_rtc.hip)
struct UserKernels {
    hipModule_t mod = nullptr;          // (owned by `keep`, shared with the density's cache and other samplers)
    std::shared_ptr<void> keep;
    hipFunction_t vec = nullptr, generic = nullptr, logpdf = nullptr, resident = nullptr, island = nullptr, init_ball = nullptr;
    hipFunction_t generation = nullptr; // one launch per generation (kmc_generation.hpp), when the sampler asked for it
    hipFunction_t staged = nullptr;     // body densities, double rows, ndim <= kStagedMaxDim: half_step_staged_body
    hipFunction_t logpdf_sep = nullptr; // a body recognised as a sum over elements: the generated form, row by row (check_sum_form)
};

}  // namespace kmc_host

struct kmc_sampler {
    kmc_config cfg{};
    int64_t h = 0, h_loc = 0, active_begin = 0, nlocal = 0, nsamples = 0;
    int64_t ld = 0;                    // device row stride in doubles (ndim rounded up to even)
    kmc::DensityParams dp{};
    kmc_host::Plan plan{};
    bool ragged_vec() const { return plan.vec && plan.ragged; }   // the half-step kernel is half_step_vec<..., RAGGED = true> (menu or runtime-compiled): ndim among its preloaded parameters
    kmc::LogpdfFn logpdf_fn = nullptr;
    kmc_user_density* user = nullptr;     // KMC_USER_DENSITY: kernels come from a runtime-compiled module
    kmc_host::UserKernels uk{};
    int grid = 0;
    int tpb = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool foreign_stream_seen = false;                    // a caller's stream was bound at some point: destroy waits for the whole device before recycling buffers
    double* d_pos = nullptr;           // rows [nrows][ld]; float elements when f32 (KMC_F32)
    bool f32 = false;
    bool own_pos = true;
    double* d_logp = nullptr;
    uint32_t* d_naccept = nullptr;
    int64_t* d_gen = nullptr;
    kmc::SchedEntry* d_sched = nullptr;
    double* d_chain = nullptr;
    double* d_chain_logp = nullptr;
    int nblob = 0;                     // body density with blobs: doubles per evaluation (kmc_user_density::nblob)
    double* d_blob = nullptr;          //   [nrows][nblob]: blob of every walker's current position (blob0s, src/samplers.jl:210, :264)
    double* d_chain_blob = nullptr;    //   [nsamples][nlocal][nblob]: KMC_STORE_BLOBS (:270)
    double* d_msum = nullptr;
    double* d_msumsq = nullptr;
    double2* d_mring = nullptr;       // moment ring (HalfStepArgs::mring): [waves][mring_depth][K][64] rows
    double* d_mring_w = nullptr;      //   and [waves][mring_depth] weights
    uint32_t* d_mcnt = nullptr;       // [waves] entries posted, then [waves] entries swept (one allocation)
    int mring_depth = 0;
    int64_t mring_waves = 0;
    int64_t gens_since_sweep = 0;
    uint32_t* d_klast = nullptr;      // vec kernels: samples already credited per walker (d_logp, d_naccept, d_klast: one block)
    double2* d_draws = nullptr;       // resident mode, one walker per thread: the draws of a launch's generations, [kDrawTableGens][nwalkers] x 32 B
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
    int launch_mode = 0;      // 0: not decided, 1: table graph, 2: eager launches, 3: updated graph (kmc_sampler_run; the two-launch kernels and, since round 5, the generation kernels)
    float calib_graph_ms = 0.f, calib_eager_ms = 0.f;   // one chunk each, when measured
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    // "updated graph": a chain of kGraphChunk * 2 kernel nodes (generation kernels: kGraphChunk) launched in the eager form (step among the preloaded
    // parameters, schedule entry in the args), their parameters rewritten before every replay; kUExec executables
    // take turns, so the host updates up to kUExec - 1 replays ahead of the one that is running (two were enough for
    // a quiet host -- the update of 128 nodes takes about as long as their replay -- but left a single replay of
    // slack: one scheduling hiccup of the host process starved the GPU, 4.27 instead of 3.8 us per launch in one run)
    hipGraph_t ugraph = nullptr;
    hipGraphExec_t uexec[kmc_host::kUExec] = {};
    hipEvent_t udone[kmc_host::kUExec] = {};
    bool uinflight[kmc_host::kUExec] = {};
    int unext = 0;
    std::vector<hipGraphNode_t> unodes;
    int64_t uchunk = 128;     // generations per replay of the updated graph (consecutive replays are ~10 us apart on the GPU: 64 -> 128 is 1-3 % of a launch-bound period; profiles/NOTES.md round 5)
    bool updated_forced = false;   // KMC_LAUNCH=updated: no budget
    int64_t feed_wait_ns = 0, feed_update_ns = 0, feed_launch_ns = 0, feed_replays = 0;   // updated-graph mode: the feeding thread's time per phase
    unsigned vec_lds = 0;          // dynamic LDS of the vector kernel (a function body evaluated per walker: one tile of proposals per wave)
    bool updated_refused = false;  // a runtime-compiled kernel the runtime would not take as a graph kernel node: table graph / eager only
    bool budget_fallback = false;  // this sampler left (or never entered) the updated-graph mode because the process budget was spent
    std::vector<std::pair<char*, size_t>> guards;      // KMC_DEBUG=poison: (guard address, size of the allocation in front of it)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool have_run_events = false;
    bool positions_set = false;
    // host-evaluated density (KMC_HOST_DENSITY)
    bool host_eval = false;
    double* d_prop = nullptr;          // [h][ld] proposals of the current half-step
    double* d_p1 = nullptr;            // [h] their log-pdfs, as returned by the callback
    double* h_prop = nullptr;          // pinned, dense [h][ndim]
    double* h_p1 = nullptr;            // pinned [h]
    double* h_prop_dev = nullptr;      // the same two arrays as the device addresses them (small batches: the kernels write the proposals
    double* h_p1_dev = nullptr;        //   straight into h_prop and read the log-pdfs straight from h_p1 -- no copies, one synchronisation)
    uint8_t* d_acc = nullptr;          // [h] accept outcomes of the current half-step (host_accepted only)
    uint8_t* h_acc = nullptr;          // pinned [h]
    hipEvent_t host_ev[kmc_host::kHostPieces] = {};   // proposals of a large half-step reach the host in pieces, each behind its event (kmc_sampler_run)
    // resident mode: exact sampler, whole (small) ensemble in one workgroup's LDS, many generations per launch
    bool resident = false;
    kmc::ResidentFn resident_kernel = nullptr;
    int resident_tpb = 256;
    bool resident_lane2 = false;       // ... two walkers per thread (1026 .. 2048 walkers: resident_lane2_body)
    bool resident_lane = false;        // ... one walker per thread (kmc_islands.hpp: resident_lane_body) instead of two lanes per walker
    // one launch per generation (kmc_generation.hpp): mid-size ensembles with short double rows, exact.  The state ping-pongs between
    // (d_pos, d_logp) and (d_pos2, d_logp2) generation by generation; kmc_sampler_run leaves it in the first pair.
    bool fused = false;
    kmc::GenerationFn generation_kernel = nullptr;
    double* d_pos2 = nullptr;
    double* d_logp2 = nullptr;
    bool sep_off = false;              // a body density recognised as a sum over elements whose check was blind at THIS sampler's ndim / parameters: run as written
    std::string sep_off_note;
    bool pos_exposed = false;          // kmc_sampler_device_ptr handed d_pos / d_logp out
    bool pos2_current = false;         // lane-striped form: the second copy of the state is consistent with the first (nobody has written d_pos / d_logp since the last run)
    bool fused_fold = false;           // lane-striped form with K == 2, L = 8 / 16 / 32: moment accumulators per WAVE in the vector kernels' transposed layout (d_isum)
    std::vector<double> carry_sum, carry_sumsq;   // a sampler that left this mode after it had run (unfuse): the moments credited until then, per dimension
    uint32_t* d_glast = nullptr;       // lane-striped form: 1 + the generation of every walker's last accepted move (GenerationArgs::glast)
    int fused_cur = 0;                 // which pair holds the state at the tail of the stream (0 between kmc_sampler_run calls)
    int fused_L = 0;                   // 0: one walker per lane (generation_lane, ndim <= 8); else rows striped over L lanes (generation_group<L, plan.K>)
    int fused_tpb = 64;                //   threads per workgroup of the lane-striped form
    // island mode (KMC_ISLANDS)
    bool islands = false;
    kmc::IslandFn island_kernel = nullptr;
    int island_K = 0;
    bool island_ragged = false;
    int64_t island_gens = 32, nislands = 0, island_size = kmc::kIslandSizeDefault;
    size_t island_lds = 0;
    double* d_isum = nullptr;                            // [nislands][4K] per-island moment sums
    double* d_isumsq = nullptr;
    // peer-to-peer sharding (KMC_P2P)
    bool p2p = false;
    bool connected = false;
    int64_t nrows = 0;                                   // rows held by this sampler (nwalkers, or nlocal for P2P)
    unsigned long long* d_flags = nullptr;               // fine-grained progress flags [shard_count]
    unsigned long long* d_err = nullptr;
    bool stream_by_walker = false;                       // KMC_STREAM_CHAIN | KMC_CHAIN_BY_WALKER: host buffers are [walker][nsamples][..]
    double *bw_scratch = nullptr, *bw_scratch_logp = nullptr;   // ... one transposed block on the device, copied out as a 2-D window (also: rows_compact of odd ndim)
    std::vector<double> bw_host;                         // ... (host buffers that could not be page-locked: a block lands here, then is scattered by memcpy)
    int64_t flushed_done = -1;                           // samples_done at the last flush of an incomplete block (nothing new: skip it)
    bool push = false;                                   // KMC_P2P_PUSH: d_pos = (1 + shard_count) blocks, see HalfStepArgs::push
    double* peer_pos[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long* peer_flags[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace kmc_host {

inline int64_t samples_done(const kmc_sampler* s)
{
    const int64_t post = s->generation - s->cfg.nburnin;
    if (post <= 0) return 0;
    const int64_t k = post / s->cfg.nthin;
    return k < s->nsamples ? k : s->nsamples;
}

// kmc_plan.hip
Plan make_plan(const kmc_config& c, int64_t n_active);
kmc::IslandFn island_fn(int density, int S, int K, bool ragged);
kmc::ResidentFn resident_fn(int density, int tpb, int K, bool ragged);
kmc::ResidentFn resident_lane_fn(int density, int ndim, bool f32);
kmc::ResidentFn resident_lane2_fn(int density, int ndim);
kmc::GenerationFn generation_fn(int density, int ndim);
kmc::GenerationFn generation_group_fn(int density, int L, int K);
bool resident_lane_wanted(int64_t ndim);
int lane_nd(int64_t ndim);
void island_perm(uint64_t seed, int64_t epoch, int64_t N, int64_t* A, int64_t* C);
uint64_t deal_seed(uint64_t seed, int32_t rank);
void deal_perm(uint64_t seed, int64_t epoch, int32_t rank, int64_t S, int64_t* A, int64_t* C);
kmc::InitBallFn init_ball_fn(int density);

// kmc_launch.hip
kmc_status ensure_graph(kmc_sampler* s);                 // capture + instantiate the table-graph chunk (no-op when it exists)
kmc_status flush_moments_now(kmc_sampler* s);            // credit every walker's current value with the samples it stood for

// kmc_copy.hip
hipError_t upload_rows(const kmc_sampler* s, double* dst_dev, const double* src_host, size_t rows);
hipError_t download_rows(const kmc_sampler* s, double* dst_host, const double* src_dev, size_t rows);
kmc_status chain_before(kmc_sampler* s, int64_t g_end);  // KMC_STREAM_CHAIN: before enqueueing generations [.., g_end)
kmc_status chain_after(kmc_sampler* s);                  //   after enqueueing up to s->generation
kmc_status chain_flush(kmc_sampler* s);                  //   at a synchronisation point
void chain_unregister(kmc_sampler* s);

// kmc_rtc.hip
kmc_status load_user(kmc_user_density* ud, bool with_vec, int L, int K, int iter, bool ragged, UserKernels* uk,
                     int resident_K = 0, bool resident_ragged = false, int island_S = 0, bool f32 = false, int64_t ndim = 0, bool p2p = false,
                     int generation_nd = 0);
void drop_updated_graph(kmc_sampler* s);                               // the updated-graph mode's executables, events and template graph (kmc_launch.hip)
kmc_status unfuse(kmc_sampler* s);                                     // back to the two-launch kernels, in place (kmc_launch.hip)
void set_offline_compiler_hint(bool wanted);                          // runtime-compiled kernels of this thread: hipcc as a child process instead of hiprtc (kmc_rtc.hip)
bool body_vec_possible(const kmc_user_density* ud, int64_t ndim);     // a function body inside the vector kernels, evaluated per walker (kmc_rtc.hip)

// kmc_p2p.hip
kmc_status check_p2p_err(kmc_sampler* s);                // a peer wait that timed out invalidates everything after it

// kmc_diag.hip
void check_guards(kmc_sampler* s);                       // KMC_DEBUG=poison: abort when a guard band was overwritten
void reinstall_abort_backtrace();                        // KMC_DEBUG=abort-backtrace: (re)install the SIGABRT handler

}  // namespace kmc_host
