/*
 * kissmcmc_hip.h -- C ABI of the MI355X-native emcee (affine-invariant ensemble sampler) hot path.
 *
 * Drop-in boundary.  The reference (mauro3/KissMCMC.jl, pure Julia) has no FFI; the path sits
 * behind three Julia methods, and this header is what a `ccall` binding for that path binds:
 *
 *   emcee(pdf, theta0s; niter, nburnin, nthin, a_scale, ...)     reference src/samplers.jl:188-216
 *   _emcee(...)  (generation loop, stretch move, accept/reject)  reference src/samplers.jl:232-293
 *   g_pdf / cdf_g_inv / sample_g                                 reference src/samplers.jl:224-230
 *
 * `make_theta0s` (src/samplers.jl:311-349) and `squash_walkers` (src/samplers.jl:372-428) are
 * host-side pre/post-processing and stay in the host language.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller owns every host buffer.
 *   - Every function returns a kmc_status; kmc_last_error() gives the message of the last
 *     failure on the calling thread.  Nothing aborts the process.
 *   - Threads: any number of handles may be used at once, each by one host thread at a time (the reference's own
 *     call is a blocking function, src/samplers.jl:188).  The library never uses the legacy (null) stream, so one
 *     thread's blocking copies do not disturb the hipGraph capture of another.
 *   - Ensembles are dense row-major [nwalkers][ndim] arrays of double, walker order = the
 *     reference's (walkers 0..nwalkers/2-1 are the first half of src/samplers.jl:247).
 *   - "generation" = one pass of src/samplers.jl:245 (two half-steps, every walker proposes
 *     once).  The reference's `niter`/`nburnin` count log-pdf evaluations; the host shim
 *     converts with the reference's own integer divisions (src/samplers.jl:203-204).
 *   - The user closure `pdf` of src/samplers.jl:257 is replaced by a fixed menu of analytic
 *     log-densities (kmc_density) evaluated on the device.
 *   - Random stream: Philox4x32-10, counter {step_lo, step_hi, walker_lo, walker_hi},
 *     key {seed_lo, seed_hi}, step = 2*generation + half, walker = global walker index: the
 *     block rocRAND's rocrand4() yields after rocrand_init(seed, walker, 4*step).  Results
 *     are a pure function of (seed, inputs), independent of sharding and launch geometry.
 */
#ifndef KISSMCMC_HIP_H
#define KISSMCMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMC_VERSION 100 /* 0.1.0 */

typedef enum kmc_status {
    KMC_OK = 0,
    KMC_ERR_A_SCALE = 1,         /* a_scale <= 1                    src/samplers.jl:200 */
    KMC_ERR_ODD_WALKERS = 2,     /* "Use an even number of walkers." src/samplers.jl:202 */
    KMC_ERR_TOO_FEW_WALKERS = 3, /* "Use more walkers: at least DOF+2, but better many more." src/samplers.jl:205 */
    KMC_ERR_BAD_ARG = 4,
    KMC_ERR_NONFINITE_LOGP = 5,  /* an initial walker has log-pdf -inf/nan (reference: make_theta0s guarantees > -Inf, src/samplers.jl:338) */
    KMC_ERR_HIP = 6,
    KMC_ERR_OOM = 7,
    KMC_ERR_NO_DEVICE = 8,
    KMC_ERR_UNSUPPORTED = 9
} kmc_status;

/* Log-density menu (all drop normalisation constants).  p[] = kmc_config.params */
typedef enum kmc_density {
    KMC_GAUSSIAN_ISO = 0, /* -1/2 sum_i ((x_i - p0)/p1)^2                       p = {mu, sigma}      */
    KMC_EXPONENTIAL  = 1, /* any x_i < 0 ? -inf : -p0 * sum_i x_i                p = {rate}           (reference README.md:15) */
    KMC_ROSENBROCK   = 2, /* -sum_{i<N-1} [p1 (x_{i+1}-x_i^2)^2 + (p0-x_i)^2]/p2 p = {a, b, scale}    (reference test/runtests.jl:68 at N=2, {1,100,20}) */
    KMC_LOGNORMAL    = 3, /* any x_i <= 0 ? -inf : sum_i [-log x_i - (log x_i - p0)^2/(2 p1^2)]  p = {mu, sigma} */
    KMC_MVNORMAL2    = 4, /* ndim == 2: -1/2 (d' P d), d = x - {p0,p1}, P = [[p2,p3],[p3,p4]] (precision matrix) */
    KMC_USER_DENSITY = 100, /* runtime-compiled: sum_d term(x_d) + sum_{d<n-1} pair(x_d, x_{d+1}); kmc_config.user_density
                              holds the handle made by kmc_user_density_create; params[0..5] are passed to it as p[] */
    KMC_HOST_DENSITY = 101  /* ANY log-density, evaluated by the caller: per half-step the device writes the batch of
                              proposals, kmc_config.host_logpdf evaluates it on the host, the device accepts/rejects.
                              Keeps the reference's arbitrary `pdf` closure (src/samplers.jl:257); bound by the callback
                              and one PCIe round trip per half-step.  Single GPU, no island mode. */
} kmc_density;

/* KMC_HOST_DENSITY callback: rows = dense [nrows][ndim]; write the log-pdf of every row to logp_out[nrows].
   Return 0, or non-zero to abort the run (kmc_sampler_run then returns KMC_ERR_BAD_ARG).  Called on the
   thread that called kmc_sampler_set_positions / kmc_sampler_run / kmc_emcee_run.  A large half-step's proposals may arrive
   in several calls (consecutive pieces of the batch, evaluated while the next piece is still crossing PCIe); with
   kmc_config.host_accepted set it is exactly one call per half-step, so that batch state (blobs) lines up with the outcomes. */
typedef int (*kmc_host_logpdf_fn)(const double* rows, int64_t nrows, int64_t ndim, double* logp_out, void* user);
/* KMC_HOST_DENSITY, optional: called after the accept test of every half-step with its outcome, so the caller can
   carry per-walker side data of its density (the reference's blobs: `blob0s[nc] = blob1` on accept :264,
   `reduce_blob!(blobs[nc], blob0s[nc])` when the state is stored :270).  accepted[i] = 1 when row i of the batch
   host_logpdf has just evaluated replaced walker row0 + i; stored = 1 when this generation's states are stored
   (after burn-in, every nthin-th).  Return 0, or non-zero to abort the run. */
typedef int (*kmc_host_accepted_fn)(const uint8_t* accepted, int64_t nrows, int64_t row0, int64_t generation,
                                    int32_t stored, void* user);

/* Many-chain Metropolis, optional: proposals from the host.  rows = current states, dense [nrows][ndim]; write
   sample_ppdf(row) for every row to proposals_out [nrows][ndim] (a SYMMETRIC proposal, as the reference requires,
   src/samplers.jl:41).  Return 0, or non-zero to abort the run. */
typedef int (*kmc_host_propose_fn)(const double* rows, int64_t nrows, int64_t ndim, double* proposals_out, void* user);

enum {
    KMC_F64 = 0, /* state and arithmetic in IEEE double, as the reference (Float64) */
    KMC_F32 = 1  /* throughput option: walker rows and the stored chain are kept in IEEE single ON THE DEVICE (half the
                    row bytes); a proposal is rounded to single before its log-density is evaluated, so a stored row and
                    its log-pdf belong together; draws, log-densities, the accept test, counters and moments stay double,
                    and so does every HOST buffer of this interface.  Densities evaluated on the device (built-in or
                    runtime-compiled), one GPU (no KMC_P2P / KMC_ISLANDS / sharding / kmc_sampler_bind_positions). */
};

/* kmc_config.flags */
enum {
    KMC_STORE_CHAIN = 1u << 0, /* keep thetas      (src/samplers.jl:269) on the device: [nsamples][nwalkers][ndim] */
    KMC_STORE_LOGP  = 1u << 1, /* keep logdensities (src/samplers.jl:271): [nsamples][nwalkers] */
    KMC_MOMENTS     = 1u << 2, /* streaming sum x, sum x^2 per dimension over the samples that would be stored */
    KMC_NO_GRAPH    = 1u << 3, /* launch every half-step eagerly instead of replaying a hipGraph */
    KMC_ISLANDS     = 1u << 6, /* ISLAND MODE (opt-in, not the reference's partner rule): islands of `island_size` walkers live in LDS
                                  for `island_gens` generations per launch and draw partners from their own complementary
                                  half; walkers are re-dealt to islands between launches.  Same target distribution, far
                                  fewer kernel boundaries and no HBM traffic inside an epoch.  Needs nwalkers % island_size == 0,
                                  island_size >= ndim + 2, ndim <= 32, shard_count == 1; no chain storage.  Works with user densities
                                  (their island's rows fit the 160 KiB of LDS for every permitted island_size and ndim). */
    KMC_STREAM_CHAIN = 1u << 11, /* with KMC_STORE_CHAIN / KMC_STORE_LOGP: the chain does NOT live in HBM.  The device keeps a ring of three
                                    blocks of sample slots; every completed block is copied to the caller's host buffers
                                    (kmc_sampler_set_chain_host) by a second stream while sampling goes on, so the number of stored
                                    samples is bounded by host memory, not by the 288 GB of HBM (the reference grows per-walker
                                    vectors without bound, src/samplers.jl:268-272; C2 with nthin = 1 is 84 GB, C5 336 GB).  The
                                    host buffers are page-locked in place (hipHostRegister), so the copies are direct DMA into
                                    their final position; if that fails the copies are blocking and staged through the library's own page-locked
                                    buffers (the sampling then waits for them).  KMC_F64, one GPU
                                    (no KMC_P2P / sharding / KMC_ISLANDS); small ensembles stay in resident mode (their launches are cut to less than a block).
                                    kmc_emcee_run switches it on by itself when the chain would not fit the device. */
    KMC_CHAIN_BY_WALKER = 1u << 12, /* kmc_emcee_run and kmc_metropolis_run: kmc_outputs.chain is [nwalkers][nsamples][ndim] and chain_logp
                                    [nwalkers][nsamples] -- the reference's own order, thetas[w][k] (src/samplers.jl:219-221,
                                    :268-272) -- instead of sample-major.  Transposed on the device before the copy
                                    (kmc_sampler_get_chain_by_walker).  Together with KMC_STREAM_CHAIN (also in kmc_sampler_create):
                                    the host buffers of kmc_sampler_set_chain_host are [nwalkers][nsamples][ndim] and
                                    [nwalkers][nsamples], and a completed block is transposed into a device scratch block and
                                    copied into them as a 2-D window by the copy stream. */
    KMC_STORE_BLOBS = 1u << 13, /* a body density with blobs (kmc_user_density_create_body_blob; the reference's hasblob=true,
                                   src/samplers.jl:194-196): keep the blob of every stored sample, [nsamples][nwalkers][nblob]
                                   (reduce_blob! with the default push!, :196, :270); read with kmc_sampler_get_blobs.  The
                                   blob of every walker's CURRENT position (blob0s, :210, :264) is kept regardless. */
    KMC_P2P_FINEGRAINED = 1u << 7, /* with KMC_P2P: keep the rows in fine-grained (coherent, uncached-for-peers) device memory */
    /* The default exchange under KMC_P2P is the pull of the drawn rows with system-scope loads, ordered by a separate signal kernel. */
    KMC_P2P_PUSH    = 1u << 9, /* with KMC_P2P: every rank keeps local copies of all the other shards and reads its partner rows
                                  from them -- with system-scope loads, like the pull (never from the reader's L2: peers write
                                  that memory); a rank that accepts a move writes the new row into its copy on every peer as
                                  well (write-through stores over xGMI).  Only accepted rows cross the fabric (C2: 23 %), once
                                  per peer, instead of every drawn row once: less per link while the acceptance is below
                                  1 / shard_count (2 ranks: half the bytes; 8 ranks: 1.9 x).  Menu densities in the vector
                                  kernels (anything else keeps the pull: kmc_sampler_describe).  Not with KMC_P2P_FINEGRAINED /
                                  kmc_sampler_init_ball.  In the default library since round 5. */
    /* (bits 8 and 10 were KMC_P2P_FOLD_SIGNAL and KMC_P2P_LAZY of rounds 1-4: removed in round 5, refused with KMC_ERR_UNSUPPORTED) */
    KMC_P2P         = 1u << 4  /* walker sharding with peer-to-peer partner reads over xGMI: the sampler holds only
                                  its shard ([2][nwalkers/2/shard_count][ndim], halves back to back), reads partner
                                  rows straight from the owning rank's HBM and synchronises half-steps with
                                  per-rank progress flags; needs kmc_sampler_p2p_export/_connect (<= 8 shards) */
};

#define KMC_P2P_HANDLE_BYTES 128
#define KMC_RCCL_ID_BYTES 128

typedef struct kmc_config {
    int32_t  dtype;         /* KMC_F64 or KMC_F32 (device storage of the rows) */
    int32_t  density;       /* kmc_density */
    double   params[8];
    int64_t  nwalkers;      /* GLOBAL ensemble size (all shards) */
    int64_t  ndim;
    int64_t  ngenerations;  /* niter_walker   = niter  / nwalkers   src/samplers.jl:203 */
    int64_t  nburnin;       /* nburnin_walker = nburnin / nwalkers  src/samplers.jl:204 */
    int64_t  nthin;         /*                                      src/samplers.jl:190 */
    double   a_scale;       /*                                      src/samplers.jl:192 */
    uint64_t seed;
    uint32_t flags;
    int32_t  device;        /* HIP device ordinal */
    int32_t  shard_rank;    /* walker sharding: this sampler updates slice shard_rank ...          */
    int32_t  shard_count;   /* ... of shard_count of EACH half; 1 = the whole ensemble (default 0 -> 1) */
    void*    user_density;  /* kmc_user_density* when density == KMC_USER_DENSITY, else NULL */
    int32_t  island_gens;   /* KMC_ISLANDS: generations per epoch (launch); 0 -> 32 */
    int32_t  island_size;   /* KMC_ISLANDS: walkers per island: 64, 128 or 256; 0 -> 256 */
    kmc_host_logpdf_fn host_logpdf; /* KMC_HOST_DENSITY: the callback, else NULL */
    void*    host_user;     /* passed through to host_logpdf / host_accepted */
    kmc_host_accepted_fn host_accepted; /* KMC_HOST_DENSITY: per-half-step accept outcomes, or NULL */
    int32_t  deal_rank;     /* DEALT SUB-ENSEMBLES (opt-in, not the reference's partner rule; see kmc_sampler_deal_pack): this sampler is */
    int32_t  deal_count;    /* sub-ensemble deal_rank of deal_count; 0 = off.  nwalkers is then THIS sub-ensemble's size */
} kmc_config;

/* Host output buffers of the one-shot call; any pointer may be NULL. */
typedef struct kmc_outputs {
    double*  chain;         /* [nsamples][nwalkers][ndim]  (needs KMC_STORE_CHAIN) */
    double*  chain_logp;    /* [nsamples][nwalkers]        (needs KMC_STORE_LOGP)  */
    double*  accept_ratio;  /* [nwalkers]                  src/samplers.jl:291 */
    int64_t* naccept;       /* [nwalkers] */
    double*  final_pos;     /* [nwalkers][ndim] */
    double*  final_logp;    /* [nwalkers] */
    double*  sum;           /* [ndim]  (needs KMC_MOMENTS) */
    double*  sumsq;         /* [ndim] */
    int64_t  nmoment;       /* out: number of (sample, walker) pairs accumulated */
    int64_t  nsamples;      /* out: (ngenerations - nburnin) / nthin  src/samplers.jl:234 */
    double   device_ms;     /* out: generation loop only, HIP events on the sampler's stream */
    double*  blobs;         /* [nsamples][nwalkers][nblob] ([nwalkers][nsamples][nblob] with KMC_CHAIN_BY_WALKER): the blobs of a
                               body density created with kmc_user_density_create_body_blob (src/samplers.jl:270) */
} kmc_outputs;

typedef struct kmc_sampler kmc_sampler; /* opaque */
typedef struct kmc_user_density kmc_user_density; /* opaque */

/* ---- library ---- */
int         kmc_version(void);
/* sizeof(kmc_config) / sizeof(kmc_metropolis_config) / sizeof(kmc_outputs) / sizeof(kmc_metropolis_outputs) as this library was
   built: a binding checks them against its own mirror of the structs when it loads the library, so layout drift fails loudly
   before the first real call (a SHORTER stale kmc_outputs would have the library read -- and write through -- `blobs` past its end). */
int         kmc_sizeof_config(void);
int         kmc_sizeof_metropolis_config(void);
int         kmc_sizeof_outputs(void);
int         kmc_sizeof_metropolis_outputs(void);
int         kmc_device_count(void);
/* hipMemGetInfo of a device (a caller deciding between a device-resident chain and KMC_STREAM_CHAIN; reference
 * src/samplers.jl:268-272 grows the chain without bound). */
kmc_status  kmc_device_free_bytes(int device, uint64_t* free_bytes, uint64_t* total_bytes);
/* Small device buffers of samplers are recycled through a per-device cache (blocks of up to 8 MiB, at most 128 MiB held;
 * KMC_DEBUG=poison turns it off): a sampler of the reference's own sizes otherwise spends more time in hipMalloc / hipFree
 * than sampling (the README call: 1.7 ms -> 1.0 ms).  This returns every block the cache holds to the device; kmc_device_free_bytes
 * counts held blocks as free. */
void        kmc_device_cache_release(void);
/* Touch every page of a host buffer (one read-modify-write of a byte per 4 KiB, up to `nthreads` threads; contents unchanged): the
 * reference returns per-walker vectors that grew during the run (src/samplers.jl:269-271); a drop-in host allocates the dense output
 * arrays BEFORE the run and has them faulted in by a helper thread while the device samples, so that the chain read-out afterwards
 * copies into resident pages (4 096 walkers x 4 doubles x 1 000 samples: read-out 15.6 -> ~5 ms).  No device involved. */
void        kmc_host_prefault(void* buffer, uint64_t nbytes, int nthreads);
const char* kmc_last_error(void);
const char* kmc_status_string(kmc_status st);

/* Validation only: src/samplers.jl:200-205 plus argument sanity. No device needed. */
kmc_status  kmc_validate(const kmc_config* cfg);

/* Stretch-factor helpers (host): src/samplers.jl:224, :227. */
double      kmc_g_pdf(double z, double a);
double      kmc_cdf_g_inv(double u, double a);

/* ---- user-supplied log-densities: the device-side stand-in for the arbitrary `pdf` closure of
 *      src/samplers.jl:257.  Two C expressions are compiled at run time (hiprtc, gfx950) into the
 *      same kernels:  log p(x) = sum_d TERM + sum_{d<n-1} PAIR, where
 *        term_expr may use  x (= x_d), d, n (= ndim), p (const double*, = params[0..5]);
 *        pair_expr may use  x (= x_d), y (= x_{d+1}), d, n, p;   NULL/"" = no pair term.
 *      Compiled code objects are cached on disk ($KMC_CACHE_DIR, else ~/.cache/kissmcmc_hip; keyed by the program, the kernel
 *      headers, the options and the hiprtc version; KMC_CACHE_DIR=off disables), so later processes skip the compiler.
 *      A term may evaluate to -INFINITY to reject a proposal.  Works in the multi-launch, resident and
 *      island modes and under KMC_P2P (the plain pull; the push is for menu densities only). */
kmc_status  kmc_user_density_create(const char* term_expr, const char* pair_expr, kmc_user_density** out);
/* The general form: the BODY of a C++ function
 *     double logpdf(const double* x, int n, const double* p) { BODY }
 * over the whole proposal x[0..n-1] (n = ndim, p = params[0..5]); any coupling between the dimensions, loops, locals;
 * return -INFINITY to reject.  The emcee samplers keep the rows lane-striped like a menu density's and evaluate the body once
 * per walker on the whole proposal (collected through LDS); with KMC_DEBUG=no-body-vec, and beyond 1024 dimensions, it runs in the
 * one-walker-per-lane kernels (the proposal is collected per lane, ndim <= 1024; for double rows of ndim <= 256 the rows are staged
 * through LDS so that memory is still read in whole rows):
 * emcee (multi-launch), initial log-pdfs, kmc_sampler_init_ball, many-chain Metropolis -- slower than a menu or term / pair
 * density of the same form (those stripe a row over lanes), far faster than a host callback.  Not with KMC_ISLANDS; under KMC_P2P the unstaged kernel.
 * A body that IS a sum over elements --
 *     double s = 0; for (int i = 0; i < n; ++i) s += f(x[i]);  return g(s);       (or: i + 1 < n, reading x[i] and x[i + 1])
 * with the loop body any statements that read the proposal only as x[i] (x[i + 1]) and change s only by `s +=` (up to four such
 * accumulators, `return g(s, t, ...)`) -- is recognised as
 * such (kmc_user_density_is_separable) and the emcee samplers run it in the lane-striped kernels of the menu densities, the loop
 * body as the per-element function and g as the finish: same operations per element, the sum in lane order instead of index
 * order (log-pdfs equal to rounding, like a menu density's).  Early returns, other indices, state carried between elements are
 * not recognised and are evaluated per walker as above.  The recognition reads text, so the first sampler over the density
 * evaluates the generated form next to the body itself on 256 test rows and keeps the route only if they agree (else: per walker,
 * as written; kmc_user_density_is_separable then returns 0 and kmc_sampler_describe says why).  Reference: the arbitrary closure pdf(theta), src/samplers.jl:257. */
kmc_status  kmc_user_density_create_body(const char* body, kmc_user_density** out);
int         kmc_user_density_is_separable(const kmc_user_density* ud);
/* ... returning a BLOB with the log-density -- the reference's `pdf(theta) -> (p, blob)` under hasblob=true
 * (src/samplers.jl:150-151, :194-196, :257) for device densities -- the body of
 *     double logpdf(const double* x, int n, const double* p, double* blob) { BODY }
 * which fills blob[0 .. nblob) (1 <= nblob <= 1024 doubles per evaluation; zero on entry).  A sampler over such a density
 * carries, next to every walker's log-pdf, the blob of its current position (blob0s[nc] = blob1 on accept, :264; initial
 * blobs from the initial evaluations, :209-210) and, with KMC_STORE_BLOBS, the blob of every stored sample (:270).
 * One GPU, double rows, no KMC_ISLANDS / KMC_P2P / KMC_STREAM_CHAIN / dealt sub-ensembles (KMC_ERR_UNSUPPORTED). */
kmc_status  kmc_user_density_create_body_blob(const char* body, int nblob, kmc_user_density** out);
int         kmc_user_density_nblob(const kmc_user_density* ud);   /* 0 for densities without blobs */
/* log-pdfs AND blobs of dense host rows pos_host [nrows][ndim] (evaluated on the device): the reference's `pdf.(theta0s)` under
 * hasblob=true (src/samplers.jl:209-210) -- e.g. the blob0 a caller's init_blobs(blob0, nsamples) receives (:238). */
kmc_status  kmc_logpdf_blob_eval_host(const kmc_config* cfg, const double* pos_host, double* logp_host, double* blob_host /* [nrows][nblob] */, int64_t nrows);
void        kmc_user_density_destroy(kmc_user_density* ud);

/* ---- one-shot: emcee + _emcee, src/samplers.jl:188-293 ---- */
kmc_status  kmc_emcee_run(const kmc_config* cfg, const double* theta0 /* host [nwalkers][ndim] */,
                          kmc_outputs* out);

/* ---- stateful sampler (device-resident state; what bench.py and the distributed driver use) ---- */
kmc_status  kmc_sampler_create(const kmc_config* cfg, kmc_sampler** out);
/* Waits for the sampler's own streams; small device buffers then go back to a per-device cache WITHOUT the device-wide wait a
 * hipFree implies.  When a caller's stream was ever bound (kmc_sampler_set_stream) or the sampler is a shard (shard_count > 1, an RCCL
 * communicator attached) the whole device is waited for first, so work other streams still have in flight on the sampler's buffers
 * (a framework's collectives on the rows, ...) cannot race the next owner of a recycled block.  Work on a buffer handed out by
 * kmc_sampler_device_ptr that the library cannot know of must have completed before this call. */
void        kmc_sampler_destroy(kmc_sampler* s);
/* Run on a caller-owned HIP stream (hipStream_t) instead of the sampler's own. */
kmc_status  kmc_sampler_set_stream(kmc_sampler* s, void* hip_stream);
/* Use a caller-owned device buffer (double [nwalkers][ndim], e.g. a torch tensor's data_ptr) for
 * the ensemble instead of the sampler's own allocation, so a collective library can gather
 * into it in place.  The caller keeps it alive; call before kmc_sampler_set_positions. */
kmc_status  kmc_sampler_bind_positions(kmc_sampler* s, void* pos_dev);
/* KMC_P2P: export this sampler's IPC handles (KMC_P2P_HANDLE_BYTES), to be all-gathered by the
 * caller (one process per GPU), then connect with the blobs of all shard_count ranks in rank order.
 * With KMC_P2P, kmc_sampler_set_positions still takes the GLOBAL ensemble (each rank keeps its
 * slices; barrier across ranks before the first run), while get_positions/get_logp/get_naccept/
 * get_accept_ratio return this shard's nwalkers/shard_count rows (first-half slice, then
 * second-half slice). */
kmc_status  kmc_sampler_p2p_export(kmc_sampler* s, void* handle_out);
kmc_status  kmc_sampler_p2p_connect(kmc_sampler* s, const void* handles /* [shard_count][KMC_P2P_HANDLE_BYTES] */);
/* The all-gather exchange of the exact partner rule, native: a sampler created with shard_rank / shard_count (no KMC_P2P)
 * holds a full replica of the ensemble and updates its slice of each half; with an RCCL communicator attached,
 * kmc_sampler_run enqueues, per half-step, the kernel and an in-place ncclAllGather of the updated half on the sampler's
 * stream (reference src/samplers.jl:273: the join, across GPUs) -- captured into the same hipGraph chunks as the
 * single-GPU run where RCCL allows capture, enqueued launch by launch otherwise; no host involvement inside a run.
 * kmc_rccl_unique_id: rank 0 creates the id (KMC_RCCL_ID_BYTES), the caller distributes it (any transport), every
 * rank calls kmc_sampler_rccl_init (collective: ncclCommInitRank).  librccl.so is loaded on first use. */
kmc_status  kmc_rccl_unique_id(void* id_out);
kmc_status  kmc_sampler_rccl_init(kmc_sampler* s, const void* id /* KMC_RCCL_ID_BYTES */);
/* Whether the chunk (kernels + all-gathers) is replayed from a captured hipGraph or enqueued launch by launch must be the SAME on
 * every rank: kmc_sampler_rccl_capture captures now and reports this rank's outcome (1 / 0; 0 also with KMC_NO_GRAPH); the
 * driver reduces the answers over the ranks (MIN) and passes the result to kmc_sampler_rccl_set_capture on every rank
 * (0: drop the captured chunk, launch by launch from now on).  Results are identical either way. */
kmc_status  kmc_sampler_rccl_capture(kmc_sampler* s, int* captured);
kmc_status  kmc_sampler_rccl_set_capture(kmc_sampler* s, int use_captured);
/* librccl.so as this process resolves it (dlopen): ncclGetVersion's code (major*10000 + minor*100 + patch) and its path;
 * KMC_ERR_UNSUPPORTED when the library or one of the entry points used here cannot be resolved.  Needs no device. */
kmc_status  kmc_rccl_version(int* version, char* path_buf /* may be NULL */, int64_t path_buflen);
/* The same wiring for shards that live in ONE process on one device (no IPC): shards[r] = the sampler of shard r.
   They run concurrently on their own streams like ranks on separate GPUs (single-process tests, timing, profiling). */
kmc_status  kmc_sampler_p2p_connect_local(kmc_sampler* s, kmc_sampler* const* shards /* [shard_count] */);
/* One fabric link measured with the pull's own access pattern (a connected KMC_P2P sampler with double rows; the peers must be idle): `nrows` whole rows of shard
 * `peer` read at random row indices -- uniform with replacement, what src/samplers.jl:250 makes a link serve -- with the kernels' system-scope loads, `reps`
 * launches -> *gather_gbs; the same shard copied whole by the runtime's device-to-device copy -> *copy_gbs.  Read-only.  peer == own rank: local memory. */
kmc_status  kmc_sampler_p2p_link_probe(kmc_sampler* s, int peer, int64_t nrows, int reps, double* gather_gbs, double* copy_gbs);
/* Upload the ensemble (host, [nwalkers][ndim], global order), evaluate the initial log-pdfs on
 * the device (src/samplers.jl:209-210), reset generation/counters.  Fails with
 * KMC_ERR_NONFINITE_LOGP if any is not finite. */
kmc_status  kmc_sampler_set_positions(kmc_sampler* s, const double* theta_host);
/* Device-side make_theta0s (src/samplers.jl:311-349, intended behaviour): every walker gets
 * theta0 + N(0, diag(ball_radius^2)), redrawn while its log-pdf is -inf (:336-341), up to `ntries` draws per ball
 * size (:327), the ball shrinking WITHIN A WALKER by the reference's compounding factors 1, 1/2, 1/8, 1/64, ...
 * (:326) over `halving_steps` sizes (any value >= 1; the factor underflows to 0, i.e. theta0 itself, near 47).
 * Two deliberate differences from the reference's loop: the shrink factor restarts at 1 for every walker (the
 * reference never resets ball_radius, so one unlucky walker would shrink the ball of all later ones; walkers here
 * are independent and drawn in parallel -- the host-side make_theta0s of the shims keeps the reference's sequential
 * behaviour), and a walker that finds no admissible point is an error (KMC_ERR_NONFINITE_LOGP with the
 * reference's message; its own error() at :345 is unreachable).
 * Random stream: Philox4x32-10, key {seed_lo ^ 0x42414c4c, seed_hi}, counter {attempt, d / 2, walker_lo, walker_hi}
 * (attempt = 0-based try index of that walker over all ball sizes, walker = GLOBAL walker index); words (w0,w1,w2)
 * -> u1 = (((w0 << 20) | (w1 >> 12)) + 1/2) 2^-52, u2 = (w2 + 1/2) 2^-32, Box-Muller pair sqrt(-2 log u1) {cos, sin}(2 pi u2)
 * for dimensions d, d + 1.  A pure function of (seed, walker): sharded samplers draw their own rows of the same ball.
 * Afterwards the sampler is ready to run, as after kmc_sampler_set_positions. */
kmc_status  kmc_sampler_init_ball(kmc_sampler* s, const double* theta0 /* [ndim] */, const double* ball_radius /* [ndim] */,
                                  uint64_t seed, int halving_steps /* 7 */, int ntries /* 100 */);
/* Checkpoint / resume: restore positions [rows][ndim], log-pdfs, acceptance counters (may be NULL = 0)
 * and the generation counter of a previous sampler with the same config.  The random stream is a pure
 * function of (seed, generation, walker), so the continued run is bit-identical to an uninterrupted
 * one; moments restart at the restored generation.  Not with chain storage, single GPU. */
kmc_status  kmc_sampler_set_state(kmc_sampler* s, const double* pos_host, const double* logp_host,
                                  const int64_t* naccept_host, int64_t generation);
/* KMC_STREAM_CHAIN: where the chain goes -- host buffers chain_host [nsamples][nwalkers][ndim] (KMC_STORE_CHAIN) and
 * chain_logp_host [nsamples][nwalkers] (KMC_STORE_LOGP), caller-owned, alive until the sampler is destroyed (or this is
 * called again).  Call before kmc_sampler_run; sample k of a run is complete in these buffers after the kmc_sampler_sync
 * that follows the generation which stored it.  kmc_sampler_get_chain then copies from them (or is a no-op for the same
 * pointers).  A sampler created with KMC_CHAIN_BY_WALKER as well takes [nwalkers][nsamples][ndim] and [nwalkers][nsamples]
 * (the stride between walkers is the whole run's nsamples) and answers kmc_sampler_get_chain_by_walker instead.
 * The buffers are page-locked in place (hipHostRegister) so that blocks arrive by DMA behind the sampling; where that is
 * refused (a container's RLIMIT_MEMLOCK) the blocks come through bounce buffers on the caller's thread -- slower, same result. */
kmc_status  kmc_sampler_set_chain_host(kmc_sampler* s, double* chain_host, double* chain_logp_host);
/* Enqueue `ngenerations` generations (asynchronous).  shard_count must be 1. */
kmc_status  kmc_sampler_run(kmc_sampler* s, int64_t ngenerations);
/* Enqueue ONE half-step (src/samplers.jl:248-273) of the current generation over this shard's
 * slice; half = 1 also advances the generation counter.  For walker-sharded drivers that
 * exchange the updated slice between half-steps.  A sampler that runs one launch per generation (small states:
 * kmc_sampler_describe) goes back to its two-launch kernels at the first such call, in place and for good (same chain, moments
 * credited so far kept); a resident-mode sampler (whole ensemble in one workgroup's LDS) answers KMC_ERR_UNSUPPORTED unless it
 * was created with KMC_NO_GRAPH, which keeps the half-step kernels. */
kmc_status  kmc_sampler_half_step(kmc_sampler* s, int half);
kmc_status  kmc_sampler_sync(kmc_sampler* s);
/* Milliseconds between the start of the first and the end of the last generation enqueued by
 * the most recent kmc_sampler_run (HIP events); synchronises. */
kmc_status  kmc_sampler_last_run_ms(kmc_sampler* s, double* ms);
int64_t     kmc_sampler_generation(const kmc_sampler* s);
int64_t     kmc_sampler_nsamples(const kmc_sampler* s);
/* Number of half-step kernel launches enqueued so far and the algorithmic bytes each moves
 * (read (2 ndim + 1) * 8, write (ndim + 1) * 8 per walker-step).  Resident mode (<= 1024 walkers, <= 2048 for menu densities with ndim <= 8): a launch
 * carries up to 1024 whole generations and is preceded by the kernel that computes its draws; both are counted. */
int64_t     kmc_sampler_launch_count(const kmc_sampler* s);
/* How kmc_sampler_run issues this sampler's launches (same kernels, same results in every mode; reference
 * src/samplers.jl:245-247 -- the generation x half-step loop -- is what the modes enqueue).
 * A sampler measures the table graph against the updated graph (or eager launches) once, at its first kmc_sampler_run of >= 896 generations -- if the
 * job was planned long (kmc_config::ngenerations >= 4096) or the sampler has come that far; until then, and for short jobs, whole chunks replay the table graph
 * (up to ~9 % slower per half-step at C2's size): a caller that drives a long job in pieces -- a progress or checkpoint loop -- should make them >= 896 generations
 * (the Python front end's progress loop does), or decide with KMC_LAUNCH=updated,budget in the environment.
 * *budget_fallback (may be NULL)
 * becomes 1 when the sampler is not in the updated-graph mode because the PROCESS-WIDE budget of graph parameter updates was
 * spent (the HIP runtime keeps host memory per update, ~80 B in HIP 7.0 -- the runtime a PyTorch wheel brings -- and ~1.4 B in 7.2; 64 MiB worth by default, priced by hipRuntimeGetVersion, KMC_DEBUG=updated-budget-mb=n in the
 * environment or kmc_set_updated_budget_mb): said once on stderr, in kmc_sampler_describe, and here. */
#define KMC_LAUNCH_UNDECIDED     0   /* only short runs so far: whole chunks from the table graph, the rest eagerly */
#define KMC_LAUNCH_TABLE_GRAPH   1   /* hipGraph replay of 64 generations, schedule read from a device table */
#define KMC_LAUNCH_EAGER         2   /* one launch per half-step from the host */
#define KMC_LAUNCH_UPDATED_GRAPH 3   /* hipGraph replay (128 generations) with per-replay kernel-node parameter updates (two launches per generation or one: both kinds of kernel) */
#define KMC_LAUNCH_SINGLE        4   /* resident / island kernels (many generations per launch), host-evaluated density */
int         kmc_sampler_launch_mode(const kmc_sampler* s, int* budget_fallback);
void        kmc_updated_budget(int64_t* calls_used, int64_t* calls_budget);   /* parameter updates so far / allowed, this process */
void        kmc_set_updated_budget_mb(double mb);                              /* priced at 80 B (HIP < 7.2) or 2 B per update; <= 0: no updated-graph mode from now on */

/* One line describing how this sampler executes (kernel family and geometry, exchange scheme). */
kmc_status  kmc_sampler_describe(const kmc_sampler* s, char* buf, int64_t buflen);

/* Device pointers for zero-copy exchange (torch / RCCL): which = 0 positions [nwalkers][ndim],
 * 1 logp [nwalkers], 2 naccept (uint32 [nwalkers]). */
void*       kmc_sampler_device_ptr(kmc_sampler* s, int which);

/* Downloads (synchronise the sampler's stream first). */
kmc_status  kmc_sampler_get_positions(kmc_sampler* s, double* host /* [nwalkers][ndim] */);
kmc_status  kmc_sampler_get_logp(kmc_sampler* s, double* host /* [nwalkers] */);
kmc_status  kmc_sampler_get_naccept(kmc_sampler* s, int64_t* host /* [nwalkers] */);
kmc_status  kmc_sampler_get_accept_ratio(kmc_sampler* s, double* host /* [nwalkers] */);
kmc_status  kmc_sampler_get_moments(kmc_sampler* s, double* sum, double* sumsq /* [ndim] */, int64_t* n);
/* Chain of this shard: [nsamples][nlocal][ndim] and [nsamples][nlocal]; nlocal = nwalkers /
 * shard_count, local order = (first-half slice, second-half slice).  With shard_count == 1
 * this is the global walker order. */
kmc_status  kmc_sampler_get_chain(kmc_sampler* s, double* chain, double* chain_logp);
/* The same samples in the reference's order (thetas[w][k], logdensities[w][k], src/samplers.jl:219-221, :268-272):
 * chain [nlocal][k][ndim], chain_logp [nlocal][k], k = samples stored so far (= nsamples after a complete run), dense.
 * Transposed on the device in pieces of walkers and copied out contiguously, so the host never reorders gigabytes
 * (what squash_walkers' default, walker-major, concatenation wants; src/samplers.jl:395-413).  With KMC_STREAM_CHAIN only
 * for a sampler created with KMC_CHAIN_BY_WALKER (then a copy of / no-op on the streamed buffers, stride nsamples). */
kmc_status  kmc_sampler_get_chain_by_walker(kmc_sampler* s, double* chain, double* chain_logp);
/* Blobs of a body density with blobs (any pointer may be NULL): current [nwalkers][nblob] = the blob of every walker's present
 * position (blob0s, src/samplers.jl:210, :264); stored (needs KMC_STORE_BLOBS) = the blobs of the samples stored so far,
 * [k][nwalkers][nblob] sample-major, or with by_walker != 0 [nwalkers][k][nblob]: blobs[w][k] in the reference's order (:238, :270). */
kmc_status  kmc_sampler_get_blobs(kmc_sampler* s, double* current, double* stored, int by_walker);

/* ---- dealt sub-ensembles: the multi-GPU mode WITHOUT a per-half-step exchange (opt-in extension) ----
 *
 * The reference draws partners from the whole complementary half (src/samplers.jl:250), which across GPUs costs one
 * exchange of walker rows per half-step (KMC_P2P, or an all-gather): fabric-bound on point-to-point xGMI.  Here each
 * GPU instead runs the reference's algorithm UNCHANGED on its own sub-ensemble of S = kmc_config.nwalkers walkers
 * (partners from that sub-ensemble's complementary half; an ordinary sampler: hipGraph replay, moments, ...) for an
 * epoch of E generations, and between epochs the walkers are RE-DEALT across the deal_count sub-ensembles by a
 * state-independent permutation: one all-to-all of S (ndim + 2) doubles per GPU per epoch (RCCL all_to_all_single in
 * the Python driver) instead of 2 E exchanges.  Every sub-ensemble update is a valid emcee move for its walkers and
 * the deal ignores the state, so the target distribution is unchanged; the partner pool is what differs from the
 * reference, which is why this is opt-in and never what bench.py reports as `value`.
 *
 * Contract (restated by the oracle, kmco_emcee_dealt):
 *   - sub-ensemble r of P initially holds global walkers [r S, (r+1) S) (kmc_sampler_set_positions takes ITS rows);
 *     its Philox key is kmc_deal_seed(seed, r) = seed + (r + 1) * 0x9E3779B97F4A7C15 (mod 2^64), walker index =
 *     local slot, so the draws of a slot do not depend on which walker sits in it;
 *   - after generation g with (g + 1) % E == 0 the deal of epoch e = (g + 1) / E - 1 happens: slot j of
 *     sub-ensemble r goes to send position t = (A j + C) mod S, (A, C) = kmc_deal_perm(seed, e, r, S); chunk
 *     q = t / (S / P) of the send buffer goes to sub-ensemble q and lands at its slots [r S / P, (r + 1) S / P)
 *     in order -- exactly what all_to_all_single(recv, send) with equal splits does;
 *   - a walker carries its position, log-pdf, acceptance counter and global index (kmc_sampler_get_walker_ids).
 * kmc_sampler_deal_pack first credits every walker's current value to the streaming moments (they are per slot).
 * A stored chain (KMC_STORE_CHAIN / KMC_STORE_LOGP) is BY SLOT, as the kernels write it; the sample of a generation is taken
 * before the deal that follows it, and which walker a slot held during an epoch follows from replaying kmc_deal_perm on
 * the host (distributed.deal_slot_ids; DealtEmcee.chain / gather_chain re-file the samples by walker) -- pooling all
 * samples, what squash_walkers does, needs no identities.
 * Needs S % (2 P) == 0, KMC_F64, a device density, no KMC_STREAM_CHAIN / KMC_P2P / KMC_ISLANDS / sharding. */
uint64_t    kmc_deal_seed(uint64_t seed, int32_t deal_rank);
kmc_status  kmc_deal_perm(uint64_t seed, int64_t epoch, int32_t deal_rank, int64_t S, int64_t* A, int64_t* C);
/* Enqueue (on the sampler's stream) the packing of this sub-ensemble's S rows of ndim + 2 doubles into send_dev
 * (device, S * (ndim + 2) doubles), shuffled for the deal of `epoch`; and the taking-over of a received buffer. */
kmc_status  kmc_sampler_deal_pack(kmc_sampler* s, int64_t epoch, void* send_dev);
kmc_status  kmc_sampler_deal_unpack(kmc_sampler* s, const void* recv_dev);
/* Global walker index held by each local slot (host, [nwalkers]); row order of get_positions / get_naccept. */
kmc_status  kmc_sampler_get_walker_ids(kmc_sampler* s, int64_t* host);
/* Checkpoint / resume of a sub-ensemble: kmc_sampler_set_state restores the slots' contents and the generation, this the
 * slot -> walker map that goes with that generation's epoch (host, [nwalkers]; replayed from kmc_deal_perm). */
kmc_status  kmc_sampler_set_walker_ids(kmc_sampler* s, const int64_t* host);

/* ---- stateless device ops on caller-owned device memory ---- */
/* logp[i] = log pdf(pos[i]) for nrows rows, src/samplers.jl:209. */
kmc_status  kmc_logpdf_eval(const kmc_config* cfg, const double* pos_dev, double* logp_dev,
                            int64_t nrows, void* hip_stream);

/* Same on dense host rows [nrows][ndim] (allocates, copies, evaluates, copies back). */
kmc_status  kmc_logpdf_eval_host(const kmc_config* cfg, const double* pos_host, double* logp_host, int64_t nrows);

/* ---- many-chain Metropolis: metropolis / _metropolis, reference src/samplers.jl:59-128 ----
 *
 * The reference's `metropolis(pdf, sample_ppdf, theta0; niter, nburnin, nthin)` is one serial Markov
 * chain; the device form runs `nchains` independent chains at once, one per lane, each the reference's
 * loop (src/samplers.jl:96-126): propose theta1 = sample_ppdf(theta0), accept iff
 * p1 - p0 > log(rand()) (strict, :101), store the current state every nthin-th step after burn-in
 * (:108-116), count accepted steps after burn-in (:105, :122-125), accept_ratio = naccept /
 * (niter - nburnin) (:127).  `niter`, `nburnin` count steps PER CHAIN, as in the reference.
 * `sample_ppdf` is the symmetric Gaussian step of the reference's tests, theta + step .* randn(ndim)
 * (test/runtests.jl:54,59,64,75), `pdf` a menu or runtime-compiled density.
 * Random stream: Philox4x32-10, key {seed_lo ^ 0x4d455452, seed_hi}, counter {it_lo, it_hi, chain, block};
 * block 0 = words (w0,w1) -> Box-Muller pair for dimensions 0,1 and (w2<<20 | w3>>12) -> the accept uniform;
 * block b >= 1 = (w0,w1) -> dimensions 4b-2,4b-1 and (w2,w3) -> dimensions 4b,4b+1.  A chain's result is
 * a pure function of (seed, chain index, inputs). */
typedef struct kmc_metropolis_config {
    int32_t  dtype;         /* KMC_F64 */
    int32_t  density;       /* kmc_density (menu or KMC_USER_DENSITY) */
    double   params[8];
    int64_t  nchains;       /* independent chains (the reference: 1) */
    int64_t  ndim;
    int64_t  niter;         /* steps per chain                      src/samplers.jl:62 */
    int64_t  nburnin;       /* discarded initial steps per chain    src/samplers.jl:63 */
    int64_t  nthin;         /*                                      src/samplers.jl:64 */
    const double* step;     /* host [ndim]: proposal scale per dimension (theta + step .* randn) */
    uint64_t seed;
    uint32_t flags;         /* KMC_STORE_CHAIN | KMC_STORE_LOGP | KMC_MOMENTS | KMC_STORE_BLOBS | KMC_CHAIN_BY_WALKER (chain [nchains][nsamples][ndim],
                               chain_logp [nchains][nsamples]: thetas[chain][sample], reordered on the device) */
    int32_t  device;
    void*    user_density;  /* kmc_user_density* when density == KMC_USER_DENSITY */
    /* Host route: ANY closure for `pdf` and / or `sample_ppdf` (the reference takes both as arbitrary functions,
       src/samplers.jl:59-61).  One iteration of all chains per round trip: proposals on the device (the Gaussian step) or
       from host_propose, their log-pdfs on the device (menu / runtime-compiled density) or from host_logpdf
       (density == KMC_HOST_DENSITY), the accept test (:101), counters and storage always on the device.  Launch- and
       PCIe-bound (tens of microseconds per iteration): the general route, not the fast one. */
    kmc_host_logpdf_fn   host_logpdf;   /* density == KMC_HOST_DENSITY: log-pdfs of a batch of rows, else NULL */
    void*                host_user;     /* passed to the three callbacks */
    kmc_host_accepted_fn host_accepted; /* optional: accept outcomes per iteration (blobs: :100-103, :116-118); row0 = 0, generation = iteration */
    kmc_host_propose_fn  host_propose;  /* optional: theta1 = sample_ppdf(theta0) for a batch of rows; `step` may then be NULL */
} kmc_metropolis_config;

typedef struct kmc_metropolis_outputs {
    double*  chain;         /* [nsamples][nchains][ndim]   thetas      (needs KMC_STORE_CHAIN)  :113 */
    double*  chain_logp;    /* [nsamples][nchains]         logdensities (needs KMC_STORE_LOGP)  :115 */
    double*  accept_ratio;  /* [nchains]                                                        :127 */
    int64_t* naccept;       /* [nchains] */
    double*  final_pos;     /* [nchains][ndim] */
    double*  final_logp;    /* [nchains] */
    double*  chain_sum;     /* [nchains][ndim] per-chain sum over the stored samples (needs KMC_MOMENTS) */
    double*  chain_sumsq;   /* [nchains][ndim] */
    int64_t  nsamples;      /* out: (niter - nburnin) / nthin   src/samplers.jl:88 */
    double   device_ms;     /* out: the sampling kernels only (HIP events) */
    double*  blobs;         /* [nsamples][nchains][nblob] ([nchains][nsamples][nblob] with KMC_CHAIN_BY_WALKER): the blob of every stored
                               sample (hasblob=true, src/samplers.jl:70-72, :103, :117) of a body density created with
                               kmc_user_density_create_body_blob; in-kernel chains only (no host_propose) */
    double*  final_blob;    /* [nchains][nblob]: blob0 of every chain at the end */
} kmc_metropolis_outputs;

/* Argument sanity only (the reference asserts nothing for metropolis).  No device needed. */
kmc_status  kmc_metropolis_validate(const kmc_metropolis_config* cfg);
kmc_status  kmc_metropolis_run(const kmc_metropolis_config* cfg, const double* theta0 /* host [nchains][ndim] */,
                               kmc_metropolis_outputs* out);

/* ---- convergence diagnostics: int_acorr / acor1d / auto_window, reference src/analysis.jl:140-167, :252-273, :280-285 ----
 * (that file is entirely commented out in the reference: the code is followed as written, there is no live behaviour to
 * match).  Integrated autocorrelation time per dimension of a chain as the samplers produce it, chain_host
 * [nsamples][nwalkers][ndim]: circular FFT autocorrelation of every walker's series normalised by its lag-0 value, first
 * nsamples/2 lags, averaged over walkers; tau = 2 cumsum(rho) - 1 at the first window M >= c tau(M); converged =
 * nsamples / tau (the reference suggests > 50).  All -1 if any value is NaN (:161-165).  Needs libhipfft.so at run time. */
kmc_status  kmc_int_acorr(const double* chain_host, int64_t nsamples, int64_t nwalkers, int64_t ndim, double c /* 5 */,
                          int device, double* tau /* [ndim] */, double* converged /* [ndim] */);
/* The same on the chain a sampler holds on the device (KMC_STORE_CHAIN, the samples stored so far; even ndim). */
kmc_status  kmc_sampler_int_acorr(kmc_sampler* s, double c, double* tau, double* converged);

/* ---- diagnostics ----
 * The random side of the accept test of reference src/samplers.jl:260, "(N-1)*log(z) + p1 - p0 >= log(rand())", exactly as
 * the half-step kernels compute it, for walkers walker0 .. walker0 + n - 1 of one step (= 2 * generation + half): the partner
 * index (:250; may be NULL), z (:252), t1 = (ndim - 1) * log z and lu = log u, into host arrays.  The kernels take the two
 * logarithms from their own < 1 ulp routine, a CPU implementation from its libm: this export lets a test measure that gap
 * and the probability that it flips an accept decision (tests/test_gpu_accept_margin.py). */
kmc_status  kmc_debug_accept_terms(uint64_t seed, uint64_t step, uint64_t walker0, int64_t n, int64_t nhalf, double a_scale,
                                   int64_t ndim, int device, int64_t* partner_host, double* z_host, double* t1_host, double* lu_host);

#ifdef __cplusplus
}
#endif
#endif /* KISSMCMC_HIP_H */
