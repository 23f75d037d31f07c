/*
 * kmc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, fp64 restatement of KissMCMC.jl's `emcee` affine-invariant ensemble sampler
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (kissmcmc.jl_amd/) never links, imports or calls it.
 *
 * Parity status: the reference (pure Julia) cannot be compiled or imported in this image
 * (no julia binary), it holds no golden vectors and never seeds its RNG
 * (reference src/samplers.jl:248-260 draws from the implicit default RNG inside
 * Threads.@threads), so bit-level parity with the reference is undefined.  This oracle is
 * pinned against every known-answer and statistical test the reference's own suite holds
 * for this path (reference test/emcee.jl:2-14 g-dist; test/emcee.jl:17-48 over the cases of
 * test/runtests.jl:52-107) -- see tests/test_oracle_pins.py -- and its Philox generator
 * against the published Random123 known-answer vectors.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference
 * repository root).
 *
 * Random stream (the build's own contract; the reference has none): Philox4x32-10
 * (Salmon, Moraes, Dror, Shaw, SC'11), counter = {step_lo, step_hi, walker_lo, walker_hi},
 * key = {seed_lo, seed_hi}, step = 2*generation + half, walker = GLOBAL walker index.
 * This is the block rocRAND returns from rocrand4() after
 * rocrand_init(seed, /subsequence/ walker, /offset/ 4*step, &state).
 * The four output words w0..w3 give, per walker-step:
 *   partner = floor(w0 * nhalf / 2^32)                      (src/samplers.jl:250)
 *   u_z     = (w1 + 0.5) * 2^-32              in (0,1)      (src/samplers.jl:230)
 *   u_acc   = (((w2 << 20) | (w3 >> 12)) + 0.5) * 2^-52     in (0,1)   (src/samplers.jl:260)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KMCO_API __attribute__((visibility("default")))

/* ---- status codes (same numbering as include/kissmcmc_hip.h, restated here) ---- */
enum {
    KMCO_OK = 0,
    KMCO_ERR_A_SCALE = 1,         /* src/samplers.jl:200  @assert a_scale>1 */
    KMCO_ERR_ODD_WALKERS = 2,     /* src/samplers.jl:202  "Use an even number of walkers." */
    KMCO_ERR_TOO_FEW_WALKERS = 3, /* src/samplers.jl:205  "Use more walkers: at least DOF+2..." */
    KMCO_ERR_BAD_ARG = 4,
    KMCO_ERR_NONFINITE_LOGP = 5
};

/* ---- density menu ---- */
enum {
    KMCO_GAUSSIAN_ISO = 0, /* -1/2 sum(((x-mu)/sigma)^2); params {mu, sigma}. cf. docstring example src/samplers.jl:186 and test/runtests.jl:80 */
    KMCO_EXPONENTIAL = 1,  /* sum(x<0 ? -inf : -rate*x); params {rate}. README.md:15 */
    KMCO_ROSENBROCK = 2,   /* -sum_{i<N-1}[b(x_{i+1}-x_i^2)^2 + (a-x_i)^2]/scale; params {a,b,scale}. test/runtests.jl:68 at N=2,(1,100,20) */
    KMCO_LOGNORMAL = 3,    /* sum(x>0 ? -log x - (log x - mu)^2/(2 sigma^2) : -inf); params {mu, sigma}. test/runtests.jl:57 up to a constant */
    KMCO_MVNORMAL2 = 4     /* 2-D normal, params {m1, m2, P11, P12, P22} (precision matrix). test/runtests.jl:62 up to a constant */
};

typedef struct {
    int32_t  density;
    int32_t  nthreads;      /* OpenMP threads for the walker loop; <=1: serial */
    double   params[8];
    int64_t  nwalkers;
    int64_t  ndim;
    int64_t  ngenerations;  /* niter_walker      src/samplers.jl:203 */
    int64_t  nburnin;       /* nburnin_walker    src/samplers.jl:204 */
    int64_t  nthin;         /*                   src/samplers.jl:190 */
    double   a_scale;       /*                   src/samplers.jl:192 */
    uint64_t seed;
    int32_t  state_f32;     /* the build's KMC_F32 option (not a reference feature): the walkers are kept in IEEE single --
                               the initial ensemble and every proposal are rounded to float before the log-density is
                               evaluated; all arithmetic stays double */
    int32_t  pad_;
} kmco_config;

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10, Random123 (Salmon et al. 2011).  Published algorithm, restated.
 * ---------------------------------------------------------------------------------------- */
KMCO_API void kmco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------------------------------
 * Stretch-factor distribution g(z) ~ 1/sqrt(z) on [1/a, a].
 * ---------------------------------------------------------------------------------------- */
/* src/samplers.jl:224 */
KMCO_API double kmco_g_pdf(double z, double a)
{
    return (1.0 / a <= z && z <= a) ? 1.0 / sqrt(z) * 1.0 / (2.0 * (sqrt(a) - sqrt(1.0 / a))) : 0.0;
}

/* src/samplers.jl:227.  Written as t = u*c1 + c0 (one fused multiply-add), z = t*t, with
 * c1 = sqrt(a)-sqrt(1/a), c0 = sqrt(1/a) hoisted, so the device kernel can match it bit for bit. */
static inline double g_c0(double a) { return sqrt(1.0 / a); }
static inline double g_c1(double a) { return sqrt(a) - sqrt(1.0 / a); }
KMCO_API double kmco_cdf_g_inv(double u, double a)
{
    double t = fma(u, g_c1(a), g_c0(a));
    return t * t;
}

/* One walker-step's random draws.  src/samplers.jl:250 (partner), :252/:230 (z), :260 (rand()). */
KMCO_API void kmco_draw(uint64_t seed, uint64_t step, uint64_t walker, int64_t nhalf,
                        int64_t* partner, double* u_z, double* u_acc)
{
    uint32_t ctr[4] = {(uint32_t)step, (uint32_t)(step >> 32), (uint32_t)walker, (uint32_t)(walker >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    kmco_philox4x32_10(ctr, key, w);
    *partner = (int64_t)(((uint64_t)w[0] * (uint64_t)nhalf) >> 32);
    *u_z = ((double)w[1] + 0.5) * 0x1.0p-32;
    uint64_t k = ((uint64_t)w[2] << 20) | (uint64_t)(w[3] >> 12);
    *u_acc = ((double)k + 0.5) * 0x1.0p-52;
}

/* The random side of the accept test, src/samplers.jl:260 "(N-1)*log(z) + p1 - p0 >= log(rand())", for walkers
 * walker0 .. walker0 + n - 1 of one step, with THIS file's arithmetic (glibc log; the same expressions as kmco_half_step):
 * z (:252), t1 = nm1 * log(z), lu = log(u).  tests/test_gpu_accept_margin.py compares them with the device's. */
KMCO_API void kmco_accept_terms(uint64_t seed, uint64_t step, uint64_t walker0, int64_t n, int64_t nhalf, double a, double nm1,
                                int64_t* partner, double* z_out, double* t1_out, double* lu_out)
{
    const double c0 = g_c0(a), c1 = g_c1(a);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int64_t p; double uz, ua;
        kmco_draw(seed, step, walker0 + (uint64_t)i, nhalf, &p, &uz, &ua);
        const double t = fma(uz, c1, c0);
        const double z = t * t;
        if (partner) partner[i] = p;
        z_out[i] = z;
        t1_out[i] = nm1 * log(z);
        lu_out[i] = log(ua);
    }
}

/* sample_g for a seeded stream: src/samplers.jl:230 */
KMCO_API double kmco_sample_g(uint64_t seed, uint64_t step, uint64_t walker, double a)
{
    int64_t p; double uz, ua;
    kmco_draw(seed, step, walker, 2, &p, &uz, &ua);
    return kmco_cdf_g_inv(uz, a);
}

/* ------------------------------------------------------------------------------------------
 * Density menu (stands in for the user closure `pdf(theta)` of src/samplers.jl:257).
 * ---------------------------------------------------------------------------------------- */
KMCO_API double kmco_logpdf(int32_t density, const double* params, const double* x, int64_t ndim)
{
    switch (density) {
    case KMCO_GAUSSIAN_ISO: {
        const double mu = params[0], inv_sigma = 1.0 / params[1];
        double s = 0.0;
        for (int64_t i = 0; i < ndim; ++i) {
            double t = (x[i] - mu) * inv_sigma;
            s += t * t;
        }
        return -0.5 * s;
    }
    case KMCO_EXPONENTIAL: { /* README.md:15: x<0 ? -Inf : -x */
        const double rate = params[0];
        double s = 0.0;
        for (int64_t i = 0; i < ndim; ++i) {
            if (x[i] < 0.0) return -INFINITY;
            s += x[i];
        }
        return -(rate * s);
    }
    case KMCO_ROSENBROCK: { /* test/runtests.jl:68 at ndim=2: -(100*(x2-x1^2)^2 + (1-x1)^2)/20 */
        const double a = params[0], b = params[1], inv_scale = 1.0 / params[2];
        double s = 0.0;
        for (int64_t i = 0; i + 1 < ndim; ++i) {
            double d = x[i + 1] - x[i] * x[i];
            double e = a - x[i];
            s += b * (d * d) + e * e;
        }
        return -(s * inv_scale);
    }
    case KMCO_LOGNORMAL: {
        const double mu = params[0], sigma = params[1];
        double s = 0.0;
        for (int64_t i = 0; i < ndim; ++i) {
            if (!(x[i] > 0.0)) return -INFINITY;
            double lx = log(x[i]);
            double t = (lx - mu) / sigma;
            s += -lx - 0.5 * t * t;
        }
        return s;
    }
    case KMCO_MVNORMAL2: {
        if (ndim != 2) return NAN;
        double d0 = x[0] - params[0], d1 = x[1] - params[1];
        return -0.5 * (params[2] * d0 * d0 + 2.0 * params[3] * d0 * d1 + params[4] * d1 * d1);
    }
    default:
        return NAN;
    }
}

/* ------------------------------------------------------------------------------------------
 * Validation: src/samplers.jl:200-205.
 * ---------------------------------------------------------------------------------------- */
KMCO_API int kmco_validate(const kmco_config* c)
{
    if (!c || c->nwalkers <= 0 || c->ndim <= 0 || c->nthin <= 0 || c->ngenerations < 0 || c->nburnin < 0)
        return KMCO_ERR_BAD_ARG;
    if (!(c->a_scale > 1.0)) return KMCO_ERR_A_SCALE;
    if (c->nwalkers % 2 != 0) return KMCO_ERR_ODD_WALKERS;
    if (c->nwalkers < c->ndim + 2) return KMCO_ERR_TOO_FEW_WALKERS;
    if (c->density == KMCO_ROSENBROCK && c->ndim < 2) return KMCO_ERR_BAD_ARG;
    if (c->density == KMCO_MVNORMAL2 && c->ndim != 2) return KMCO_ERR_BAD_ARG;
    return KMCO_OK;
}

/* ------------------------------------------------------------------------------------------
 * One half-step over a contiguous range of active walkers -- the body of the
 * Threads.@threads loop, src/samplers.jl:248-273 (without storage, which the caller does).
 *
 *   pos     [nwalkers][ndim] row-major, global walker order
 *   logp    [nwalkers]
 *   naccept [nwalkers]
 *   half    0: update walkers [0,h) with partners from [h,2h); 1: swapped  (src/samplers.jl:247)
 *   active_begin, n_active: sub-range of the active half handled by this call (walker sharding)
 * ---------------------------------------------------------------------------------------- */
KMCO_API void kmco_half_step(const kmco_config* c, double* pos, double* logp, int64_t* naccept,
                             int64_t generation, int half, int64_t active_begin, int64_t n_active,
                             int count_accept)
{
    const int64_t nd = c->ndim, h = c->nwalkers / 2;
    const double c0 = g_c0(c->a_scale), c1 = g_c1(c->a_scale);
    const double nm1 = (double)(nd - 1);
    const uint64_t step = 2ull * (uint64_t)generation + (uint64_t)half;
    const int64_t act0 = (int64_t)half * h, oth0 = (int64_t)(1 - half) * h;

#pragma omp parallel num_threads(c->nthreads > 1 ? c->nthreads : 1)
    {
        double* y = (double*)malloc(sizeof(double) * (size_t)nd);
        /* static partition of the active walkers over the team, as `omp for schedule(static)` would cut it; done by
         * hand so that a thread can look ahead inside its own range: the draws of walker i + KMCO_PF are computed
         * KMCO_PF iterations early (each still computed exactly once) and the partner row they name is prefetched --
         * with the team spread over several L3 domains the random partner row is otherwise a serial cache miss per
         * walker-step.  Purely a memory-latency measure: every walker's arithmetic is unchanged. */
        enum { KMCO_PF = 8 };
        int nth = 1, tid = 0;
#ifdef _OPENMP
        nth = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
        const int64_t per = n_active / nth, rem = n_active % nth;
        const int64_t lo = tid * per + (tid < rem ? tid : rem), hi = lo + per + (tid < rem ? 1 : 0);
        int64_t q_no[KMCO_PF]; double q_uz[KMCO_PF], q_ua[KMCO_PF];
        for (int64_t i = lo; i < hi && i < lo + KMCO_PF; ++i) {
            kmco_draw(c->seed, step, (uint64_t)(act0 + active_begin + i), h, &q_no[i % KMCO_PF], &q_uz[i % KMCO_PF], &q_ua[i % KMCO_PF]);
            const char* pr = (const char*)(pos + (oth0 + q_no[i % KMCO_PF]) * nd);
            for (int64_t b = 0; b < nd * 8; b += 64) __builtin_prefetch(pr + b, 0, 1);
        }
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t nc = act0 + active_begin + i;           /* :248 */
            const int64_t no_rel = q_no[i % KMCO_PF];
            const double uz = q_uz[i % KMCO_PF], ua = q_ua[i % KMCO_PF];
            if (i + KMCO_PF < hi) {                               /* the slot is free now: look ahead */
                const int64_t s2 = (i + KMCO_PF) % KMCO_PF;
                kmco_draw(c->seed, step, (uint64_t)(nc + KMCO_PF), h, &q_no[s2], &q_uz[s2], &q_ua[s2]);
                const char* pr = (const char*)(pos + (oth0 + q_no[s2]) * nd);
                for (int64_t b = 0; b < nd * 8; b += 64) __builtin_prefetch(pr + b, 0, 1);
            }
            const int64_t no = oth0 + no_rel;                      /* :250 rand(ncos) */
            const double t = fma(uz, c1, c0);
            const double z = t * t;                                /* :252 sample_g */
            const double* xc = pos + nc * nd;
            const double* xo = pos + no * nd;
            for (int64_t d = 0; d < nd; ++d)                       /* :255 theta0s[no] .+ z .* (theta0s[nc] .- theta0s[no]) */
                y[d] = fma(z, xc[d] - xo[d], xo[d]);
            if (c->state_f32)
                for (int64_t d = 0; d < nd; ++d) y[d] = (double)(float)y[d];
            const double p1 = kmco_logpdf(c->density, c->params, y, nd);   /* :257 */
            const double lhs = (nm1 * log(z) + p1) - logp[nc];     /* :260, left to right */
            if (lhs >= log(ua)) {                                  /* :260 note >= */
                memcpy(pos + nc * nd, y, sizeof(double) * (size_t)nd);     /* :261 */
                logp[nc] = p1;                                     /* :262 */
                if (count_accept) naccept[nc] += 1;                /* :265 (counts before n==0 are zeroed at :285-288) */
            }
        }
        free(y);
    }
}

/* Streaming moments of one stored generation (the build's stand-in for a reduce_blob! that sums,
 * src/samplers.jl:270 / test/runtests.jl:102-105): msum[d] += sum_w x[w][d], msumsq[d] += sum_w x[w][d]^2.
 * Threaded like the walker loop: fixed blocks of KMCO_MBLK walkers are summed into per-block partials
 * (parallel for), the partials are then added in block order -- the result does not depend on the
 * thread count.  part: scratch [nblk][2][nd]. */
#define KMCO_MBLK 256
static void moments_add(const double* pos, int64_t nw, int64_t nd, double* msum, double* msumsq,
                        double* part, int64_t nblk, int nthreads)
{
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
    for (int64_t b = 0; b < nblk; ++b) {
        double* ps = part + b * 2 * nd;
        double* pq = ps + nd;
        for (int64_t d = 0; d < nd; ++d) { ps[d] = 0.0; pq[d] = 0.0; }
        const int64_t w1 = (b + 1) * KMCO_MBLK < nw ? (b + 1) * KMCO_MBLK : nw;
        for (int64_t w = b * KMCO_MBLK; w < w1; ++w)
            for (int64_t d = 0; d < nd; ++d) {
                const double v = pos[w * nd + d];
                ps[d] += v;
                pq[d] += v * v;
            }
    }
    for (int64_t b = 0; b < nblk; ++b)
        for (int64_t d = 0; d < nd; ++d) {
            if (msum) msum[d] += part[b * 2 * nd + d];
            if (msumsq) msumsq[d] += part[b * 2 * nd + nd + d];
        }
}

/* ------------------------------------------------------------------------------------------
 * emcee + _emcee on dense arrays: src/samplers.jl:188-216 and :232-293.
 *
 * Outputs (any may be NULL):
 *   chain        [nsamples][nwalkers][ndim]   sample k of walker w = thetas[w][k] of :269
 *   chain_logp   [nsamples][nwalkers]                              logdensities[w][k] of :271
 *   accept_ratio [nwalkers]                                        :291
 *   naccept      [nwalkers]
 *   final_pos    [nwalkers][ndim], final_logp [nwalkers]
 *   msum, msumsq [ndim]  sum over stored samples and walkers of x, x^2;  *nmoment = count
 * nsamples = (ngenerations - nburnin) / nthin   (:234)
 * ---------------------------------------------------------------------------------------- */
KMCO_API int kmco_emcee(const kmco_config* c, const double* theta0,
                        double* chain, double* chain_logp, double* accept_ratio, int64_t* naccept_out,
                        double* final_pos, double* final_logp,
                        double* msum, double* msumsq, int64_t* nmoment)
{
    int st = kmco_validate(c);
    if (st != KMCO_OK) return st;
    const int64_t nw = c->nwalkers, nd = c->ndim, h = nw / 2;

    double* pos = (double*)malloc(sizeof(double) * (size_t)(nw * nd));
    double* logp = (double*)malloc(sizeof(double) * (size_t)nw);
    int64_t* nacc = (int64_t*)calloc((size_t)nw, sizeof(int64_t));
    memcpy(pos, theta0, sizeof(double) * (size_t)(nw * nd));   /* :198 deepcopy */
    if (c->state_f32)
        for (int64_t i = 0; i < nw * nd; ++i) pos[i] = (double)(float)pos[i];
    for (int64_t w = 0; w < nw; ++w) {                          /* :209-210 initial log-pdfs */
        logp[w] = kmco_logpdf(c->density, c->params, pos + w * nd, nd);
        if (!isfinite(logp[w])) { free(pos); free(logp); free(nacc); return KMCO_ERR_NONFINITE_LOGP; }
    }
    if (msum) memset(msum, 0, sizeof(double) * (size_t)nd);
    if (msumsq) memset(msumsq, 0, sizeof(double) * (size_t)nd);
    int64_t nmom = 0;
    const int64_t nblk = (nw + KMCO_MBLK - 1) / KMCO_MBLK;
    double* mpart = (msum || msumsq) ? (double*)malloc(sizeof(double) * (size_t)(nblk * 2 * nd)) : NULL;

    /* generation g in [0,G)  <->  reference n = g + 1 - nburnin  (:245) */
    for (int64_t g = 0; g < c->ngenerations; ++g) {
        const int64_t n = g + 1 - c->nburnin;
        for (int half = 0; half < 2; ++half)                    /* :246-247 */
            kmco_half_step(c, pos, logp, nacc, g, half, 0, h, /*count_accept=*/n > 0);
        if (n > 0 && n % c->nthin == 0) {                       /* :268 */
            const int64_t k = n / c->nthin - 1;
            if (k < (c->ngenerations - c->nburnin) / c->nthin) {
                if (chain) memcpy(chain + k * nw * nd, pos, sizeof(double) * (size_t)(nw * nd));
                if (chain_logp) memcpy(chain_logp + k * nw, logp, sizeof(double) * (size_t)nw);
                if (msum || msumsq) moments_add(pos, nw, nd, msum, msumsq, mpart, nblk, c->nthreads);
                nmom += nw;
            }
        }
    }
    const double denom = (double)(c->ngenerations - c->nburnin);   /* :291 (may be 0 -> inf/nan, as in the reference) */
    for (int64_t w = 0; w < nw; ++w) {
        if (accept_ratio) accept_ratio[w] = (double)nacc[w] / denom;
        if (naccept_out) naccept_out[w] = nacc[w];
    }
    if (final_pos) memcpy(final_pos, pos, sizeof(double) * (size_t)(nw * nd));
    if (final_logp) memcpy(final_logp, logp, sizeof(double) * (size_t)nw);
    if (nmoment) *nmoment = nmom;
    free(pos); free(logp); free(nacc); free(mpart);
    return KMCO_OK;
}

KMCO_API int kmco_sizeof_config(void) { return (int)sizeof(kmco_config); }

/* ------------------------------------------------------------------------------------------
 * Seeded make_theta0s on dense arrays: src/samplers.jl:311-349 -- the restatement of the product's
 * DEVICE-side initial ball (kmc_sampler_init_ball); the host-side make_theta0s, which draws from a
 * caller-supplied generator in the reference's sequential order, is restated in oracle/host.py.
 *
 * Reference loop (per walker i, :323): for k in 1:ball_radius_halfing_steps (:324) the radius is scaled by
 * 1/2^(k-1) (:326), then up to ntries (:327) candidates theta0 .+ randn(npara) .* ball_radius (:328-332)
 * are tried and the first with pdf > -Inf is kept (:336-341).  Followed as INTENDED where the
 * reference's own code is order-dependent or unreachable:
 *   - the shrink factor restarts at 1 for every walker (the reference never resets ball_radius, so
 *     the shrinkage of one unlucky walker would carry over to all later ones, :326) -- walkers are
 *     then independent of each other, which is what lets the device draw them in parallel;
 *   - a walker that finds no admissible point is reported (return value = number of such walkers;
 *     the reference's error(...) at :345 is unreachable).
 * Within a walker the compounding 1, 1/2, 1/8, 1/64, ... of :326 is kept.
 *
 * Random stream (the build's contract; the reference draws from an unseeded global generator):
 * Philox4x32-10, key = {seed_lo ^ 0x42414c4c ("BALL"), seed_hi}, counter = {attempt, pair, walker_lo,
 * walker_hi}; attempt = 0-based index of the try over ALL ball sizes of this walker, pair = d / 2.
 * Words (w0, w1, w2): u1 = (((w0 << 20) | (w1 >> 12)) + 1/2) 2^-52, u2 = (w2 + 1/2) 2^-32,
 * r = sqrt(-2 log u1); dimension 2 pair gets r cos(2 pi u2), dimension 2 pair + 1 gets r sin(2 pi u2).
 * log / sin / cos come from the platform libm: device and host agree to rounding, not bit for bit.
 *
 *   theta0, radius [ndim];  pos [nrows][ndim], logp [nrows], attempts [nrows] (tries used, or -1) -- any
 *   output may be NULL;  row r is GLOBAL walker walker0 + r.
 * ---------------------------------------------------------------------------------------- */
KMCO_API int64_t kmco_init_ball(int32_t density, const double* params, const double* theta0, const double* radius,
                                int64_t nrows, int64_t walker0, int64_t ndim, int32_t halving_steps, int32_t ntries,
                                uint64_t seed, double* pos, double* logp, int64_t* attempts)
{
    if (!params || !theta0 || !radius || nrows < 0 || ndim <= 0 || halving_steps < 1 || ntries < 1) return -1;
    const uint32_t key[2] = {(uint32_t)seed ^ 0x42414c4cu, (uint32_t)(seed >> 32)};
    int64_t nfail = 0;
    double* x = (double*)malloc(sizeof(double) * (size_t)ndim);
    for (int64_t r = 0; r < nrows; ++r) {                                      /* :323 */
        const uint64_t walker = (uint64_t)(walker0 + r);
        double shrink = 1.0;
        uint32_t attempt = 0;
        int found = 0;
        double p = -INFINITY;
        for (int k = 1; k <= halving_steps && !found; ++k) {                   /* :324 */
            shrink *= ldexp(1.0, -(k - 1));                                    /* :326 ball_radius *= 1/2^(k-1) */
            for (int t = 0; t < ntries && !found; ++t, ++attempt) {            /* :327 */
                for (int64_t d = 0; d < ndim; d += 2) {                        /* :328-332 theta0 .+ randn(npara) .* ball_radius */
                    const uint32_t ctr[4] = {attempt, (uint32_t)(d >> 1), (uint32_t)walker, (uint32_t)(walker >> 32)};
                    uint32_t w[4];
                    kmco_philox4x32_10(ctr, key, w);
                    const double u1 = ((double)(((uint64_t)w[0] << 20) | (uint64_t)(w[1] >> 12)) + 0.5) * 0x1.0p-52;
                    const double u2 = ((double)w[2] + 0.5) * 0x1.0p-32;
                    const double rad = sqrt(-2.0 * log(u1));
                    const double ang = 6.283185307179586476925286766559 * u2;
                    x[d] = theta0[d] + (rad * cos(ang)) * (radius[d] * shrink);
                    if (d + 1 < ndim) x[d + 1] = theta0[d + 1] + (rad * sin(ang)) * (radius[d + 1] * shrink);
                }
                p = kmco_logpdf(density, params, x, ndim);                     /* :336 */
                if (p > -INFINITY && p == p) found = 1;                        /* :338 */
            }
        }
        if (!found) { ++nfail; p = -INFINITY; }                                /* :345 (intended) */
        if (pos) memcpy(pos + r * ndim, x, sizeof(double) * (size_t)ndim);     /* :339 */
        if (logp) logp[r] = p;
        if (attempts) attempts[r] = found ? (int64_t)attempt : -1;
    }
    free(x);
    return nfail;
}

/* ------------------------------------------------------------------------------------------
 * ISLAND MODE (an extension of the build, NOT a reference feature; opt-in).
 *
 * The ensemble is cut into islands of S walkers; for `epoch_gens` generations every island runs
 * the reference's half-split stretch move (src/samplers.jl:245-274) on ITS OWN walkers only
 * (partner drawn from the island's complementary half instead of the ensemble's), then the walkers
 * are re-dealt to islands by the permutation slot s -> walker (A*s + C) mod N of that epoch.
 * Each island update is a valid emcee kernel for its S walkers and the re-deal does not depend on
 * the state, so the target distribution is unchanged; the partner-selection rule is what differs.
 *
 * Random stream: the same Philox block as the exact mode, keyed by step = 2*generation + half and
 * by the SLOT index (island * S + local index) instead of the walker index.
 * ---------------------------------------------------------------------------------------- */
static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

/* (A, C) of epoch e: from Philox(ctr = {e_lo, e_hi, 0x49534c41 "ISLA", 0}, key = seed); A is made
 * coprime to N by stepping upwards; epoch 0 is the identity deal. */
KMCO_API void kmco_island_perm(uint64_t seed, int64_t epoch, int64_t N, int64_t* A, int64_t* C)
{
    if (epoch == 0 || N <= 2) { *A = 1; *C = 0; return; }
    uint32_t ctr[4] = {(uint32_t)epoch, (uint32_t)((uint64_t)epoch >> 32), 0x49534c41u, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    kmco_philox4x32_10(ctr, key, w);
    int64_t a = (int64_t)((((uint64_t)w[0] << 32) | w[1]) % (uint64_t)N);
    if (a < 1) a = 1;
    while (gcd64(a, N) != 1) a = a % N + 1 >= N ? 1 : a + 1;
    *A = a;
    *C = (int64_t)((((uint64_t)w[2] << 32) | w[3]) % (uint64_t)N);
}

KMCO_API int kmco_emcee_islands(const kmco_config* c, int64_t S, int64_t epoch_gens, const double* theta0,
                                double* accept_ratio, int64_t* naccept_out, double* final_pos, double* final_logp,
                                double* msum, double* msumsq, int64_t* nmoment)
{
    int st = kmco_validate(c);
    if (st != KMCO_OK) return st;
    const int64_t nw = c->nwalkers, nd = c->ndim;
    if (S <= 0 || S % 2 != 0 || nw % S != 0 || S < nd + 2 || epoch_gens <= 0) return KMCO_ERR_BAD_ARG;
    const int64_t B = nw / S, hs = S / 2;
    const double c0 = g_c0(c->a_scale), c1 = g_c1(c->a_scale), nm1 = (double)(nd - 1);

    double* pos = (double*)malloc(sizeof(double) * (size_t)(nw * nd));
    double* logp = (double*)malloc(sizeof(double) * (size_t)nw);
    int64_t* nacc = (int64_t*)calloc((size_t)nw, sizeof(int64_t));
    memcpy(pos, theta0, sizeof(double) * (size_t)(nw * nd));
    for (int64_t w = 0; w < nw; ++w) {
        logp[w] = kmco_logpdf(c->density, c->params, pos + w * nd, nd);
        if (!isfinite(logp[w])) { free(pos); free(logp); free(nacc); return KMCO_ERR_NONFINITE_LOGP; }
    }
    if (msum) memset(msum, 0, sizeof(double) * (size_t)nd);
    if (msumsq) memset(msumsq, 0, sizeof(double) * (size_t)nd);
    int64_t nmom = 0;
    const int64_t nsamples = c->ngenerations > c->nburnin ? (c->ngenerations - c->nburnin) / c->nthin : 0;

    for (int64_t g0 = 0, epoch = 0; g0 < c->ngenerations; g0 += epoch_gens, ++epoch) {
        int64_t A, C;
        kmco_island_perm(c->seed, epoch, nw, &A, &C);
        const int64_t g1 = g0 + epoch_gens < c->ngenerations ? g0 + epoch_gens : c->ngenerations;
#pragma omp parallel for schedule(static) num_threads(c->nthreads > 1 ? c->nthreads : 1)
        for (int64_t b = 0; b < B; ++b) {
            double* y = (double*)malloc(sizeof(double) * (size_t)nd);
            for (int64_t g = g0; g < g1; ++g) {
                const int64_t n = g + 1 - c->nburnin;
                for (int half = 0; half < 2; ++half) {
                    const uint64_t step = 2ull * (uint64_t)g + (uint64_t)half;
                    for (int64_t i = 0; i < hs; ++i) {
                        const int64_t slot = b * S + (int64_t)half * hs + i;        /* active slot */
                        const int64_t nc = (int64_t)(((__int128)A * slot + C) % nw); /* its walker  */
                        int64_t prel; double uz, ua;
                        kmco_draw(c->seed, step, (uint64_t)slot, hs, &prel, &uz, &ua);
                        const int64_t pslot = b * S + (int64_t)(1 - half) * hs + prel;
                        const int64_t no = (int64_t)(((__int128)A * pslot + C) % nw);
                        const double t = fma(uz, c1, c0);
                        const double z = t * t;
                        const double* xc = pos + nc * nd;
                        const double* xo = pos + no * nd;
                        for (int64_t d = 0; d < nd; ++d) y[d] = fma(z, xc[d] - xo[d], xo[d]);
                        const double p1 = kmco_logpdf(c->density, c->params, y, nd);
                        const double lhs = (nm1 * log(z) + p1) - logp[nc];
                        if (lhs >= log(ua)) {
                            memcpy(pos + nc * nd, y, sizeof(double) * (size_t)nd);
                            logp[nc] = p1;
                            if (n > 0) nacc[nc] += 1;
                        }
                    }
                }
            }
            free(y);
        }
        /* samples of this epoch: the state after each kept generation is needed, so redo the
         * bookkeeping generation by generation only for the moments (cheap form: moments are
         * accumulated inside the island loop on the device; here, recompute serially) */
        (void)nsamples;
    }
    /* moments need per-generation states: second pass, serial and generation-major */
    if (msum || msumsq) {
        memcpy(pos, theta0, sizeof(double) * (size_t)(nw * nd));
        for (int64_t w = 0; w < nw; ++w) logp[w] = kmco_logpdf(c->density, c->params, pos + w * nd, nd);
        double* y = (double*)malloc(sizeof(double) * (size_t)nd);
        for (int64_t g = 0; g < c->ngenerations; ++g) {
            const int64_t epoch = g / epoch_gens;
            int64_t A, C;
            kmco_island_perm(c->seed, epoch, nw, &A, &C);
            const int64_t n = g + 1 - c->nburnin;
            for (int half = 0; half < 2; ++half) {
                const uint64_t step = 2ull * (uint64_t)g + (uint64_t)half;
                for (int64_t b = 0; b < B; ++b)
                    for (int64_t i = 0; i < hs; ++i) {
                        const int64_t slot = b * S + (int64_t)half * hs + i;
                        const int64_t nc = (int64_t)(((__int128)A * slot + C) % nw);
                        int64_t prel; double uz, ua;
                        kmco_draw(c->seed, step, (uint64_t)slot, hs, &prel, &uz, &ua);
                        const int64_t pslot = b * S + (int64_t)(1 - half) * hs + prel;
                        const int64_t no = (int64_t)(((__int128)A * pslot + C) % nw);
                        const double t = fma(uz, c1, c0);
                        const double z = t * t;
                        const double* xc = pos + nc * nd;
                        const double* xo = pos + no * nd;
                        for (int64_t d = 0; d < nd; ++d) y[d] = fma(z, xc[d] - xo[d], xo[d]);
                        const double p1 = kmco_logpdf(c->density, c->params, y, nd);
                        const double lhs = (nm1 * log(z) + p1) - logp[nc];
                        if (lhs >= log(ua)) { memcpy(pos + nc * nd, y, sizeof(double) * (size_t)nd); logp[nc] = p1; }
                    }
            }
            if (n > 0 && n % c->nthin == 0 && n / c->nthin - 1 < nsamples) {
                for (int64_t w = 0; w < nw; ++w)
                    for (int64_t d = 0; d < nd; ++d) {
                        const double v = pos[w * nd + d];
                        if (msum) msum[d] += v;
                        if (msumsq) msumsq[d] += v * v;
                    }
                nmom += nw;
            }
        }
        free(y);
    }
    const double denom = (double)(c->ngenerations - c->nburnin);
    for (int64_t w = 0; w < nw; ++w) {
        if (accept_ratio) accept_ratio[w] = (double)nacc[w] / denom;
        if (naccept_out) naccept_out[w] = nacc[w];
    }
    if (final_pos) memcpy(final_pos, pos, sizeof(double) * (size_t)(nw * nd));
    if (final_logp) memcpy(final_logp, logp, sizeof(double) * (size_t)nw);
    if (nmoment) *nmoment = nmom;
    free(pos); free(logp); free(nacc);
    return KMCO_OK;
}

/* ------------------------------------------------------------------------------------------
 * DEALT SUB-ENSEMBLES (an extension of the build, NOT a reference feature; opt-in) -- the multi-GPU mode
 * without a per-half-step exchange.  Restated independently of the product code.
 *
 * The N = P*S walkers are held by P sub-ensembles of S slots.  For an epoch of `epoch_gens` generations
 * sub-ensemble r runs the reference's algorithm unchanged (kmco_half_step: src/samplers.jl:245-274) on the S
 * walkers in its slots -- partners from ITS complementary half -- with its own Philox key
 * seed + (r + 1) * 0x9E3779B97F4A7C15 and the slot index as walker index.  After generation g with
 * (g + 1) % epoch_gens == 0 the walkers are re-dealt (epoch e = (g + 1) / epoch_gens - 1): slot j of
 * sub-ensemble r goes to send position t = (A j + C) mod S with (A, C) from
 * Philox(ctr = {e_lo, e_hi, 0x4445414c "DEAL", r}, key = seed), A made coprime to S by stepping upwards; with
 * c = S / P, chunk q = t / c goes to sub-ensemble q and lands in its slot r c + t % c.  A walker carries its
 * position, log-pdf, acceptance counter and global index; sub-ensemble r starts with walkers [r S, (r+1) S).
 * Each sub-ensemble update is a valid emcee move for its walkers and the deal does not look at the state, so
 * the target distribution is unchanged; the partner-selection rule is what differs from the reference.
 * ---------------------------------------------------------------------------------------- */
KMCO_API uint64_t kmco_deal_seed(uint64_t seed, int32_t rank) { return seed + (uint64_t)(rank + 1) * 0x9E3779B97F4A7C15ull; }

KMCO_API void kmco_deal_perm(uint64_t seed, int64_t epoch, int32_t rank, int64_t S, int64_t* A, int64_t* C)
{
    uint32_t ctr[4] = {(uint32_t)epoch, (uint32_t)((uint64_t)epoch >> 32), 0x4445414cu, (uint32_t)rank};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    kmco_philox4x32_10(ctr, key, w);
    int64_t a = (int64_t)((((uint64_t)w[0] << 32) | w[1]) % (uint64_t)S);
    if (a < 1) a = 1;
    while (gcd64(a, S) != 1) a = a + 1 >= S ? 1 : a + 1;
    *A = a;
    *C = (int64_t)((((uint64_t)w[2] << 32) | w[3]) % (uint64_t)S);
}

/* c->nwalkers = N (all sub-ensembles).  Outputs (any may be NULL) are indexed by GLOBAL WALKER (the index a walker
 * had in theta0), except slot_ids [N]: the walker each slot holds at the end (slot = r S + j). */
/* chain [nsamples][N][ndim], chain_logp [nsamples][N]: by GLOBAL WALKER as well -- a walker's stored samples (:268-272) follow
 * it through the deals (the sample of a generation is taken before the deal that follows it). */
KMCO_API int kmco_emcee_dealt_chain(const kmco_config* c, int32_t P, int64_t epoch_gens, const double* theta0,
                                    double* accept_ratio, int64_t* naccept_out, double* final_pos, double* final_logp,
                                    int64_t* slot_ids, double* msum, double* msumsq, int64_t* nmoment,
                                    double* chain, double* chain_logp)
{
    if (!c || P < 1 || epoch_gens < 1 || c->nwalkers % P != 0) return KMCO_ERR_BAD_ARG;
    const int64_t N = c->nwalkers, nd = c->ndim, S = N / P;
    kmco_config sub = *c;
    sub.nwalkers = S;
    int st = kmco_validate(&sub);                                   /* every sub-ensemble is an emcee ensemble: :200-205 */
    if (st != KMCO_OK) return st;
    if (S % P != 0) return KMCO_ERR_BAD_ARG;
    const int64_t chunk = S / P;

    double* pos = (double*)malloc(sizeof(double) * (size_t)(N * nd));
    double* pos2 = (double*)malloc(sizeof(double) * (size_t)(N * nd));
    double* logp = (double*)malloc(sizeof(double) * (size_t)N);
    double* logp2 = (double*)malloc(sizeof(double) * (size_t)N);
    int64_t* nacc = (int64_t*)calloc((size_t)N, sizeof(int64_t));
    int64_t* nacc2 = (int64_t*)calloc((size_t)N, sizeof(int64_t));
    int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
    int64_t* ids2 = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
    memcpy(pos, theta0, sizeof(double) * (size_t)(N * nd));
    int bad = 0;
    for (int64_t w = 0; w < N; ++w) {
        ids[w] = w;
        logp[w] = kmco_logpdf(c->density, c->params, pos + w * nd, nd);    /* :209-210 */
        if (!isfinite(logp[w])) bad = 1;
    }
    if (msum) memset(msum, 0, sizeof(double) * (size_t)nd);
    if (msumsq) memset(msumsq, 0, sizeof(double) * (size_t)nd);
    int64_t nmom = 0;
    const int64_t nblk = (N + KMCO_MBLK - 1) / KMCO_MBLK;
    double* mpart = (msum || msumsq) ? (double*)malloc(sizeof(double) * (size_t)(nblk * 2 * nd)) : NULL;
    const int64_t nsamples = c->ngenerations > c->nburnin ? (c->ngenerations - c->nburnin) / c->nthin : 0;

    for (int64_t g = 0; g < c->ngenerations && !bad; ++g) {
        const int64_t n = g + 1 - c->nburnin;                              /* :245 */
        for (int32_t r = 0; r < P; ++r) {
            sub.seed = kmco_deal_seed(c->seed, r);
            for (int half = 0; half < 2; ++half)                           /* :246-247, inside sub-ensemble r */
                kmco_half_step(&sub, pos + r * S * nd, logp + r * S, nacc + r * S, g, half, 0, S / 2, n > 0);
        }
        if (n > 0 && n % c->nthin == 0 && n / c->nthin - 1 < nsamples) {   /* :268 */
            if (msum || msumsq) moments_add(pos, N, nd, msum, msumsq, mpart, nblk, c->nthreads);
            nmom += N;
            const int64_t k = n / c->nthin - 1;
            for (int64_t sl = 0; sl < N && (chain || chain_logp); ++sl) {  /* :269-271, filed under the walker the slot holds */
                const int64_t w = ids[sl];
                if (chain) memcpy(chain + (k * N + w) * nd, pos + sl * nd, sizeof(double) * (size_t)nd);
                if (chain_logp) chain_logp[k * N + w] = logp[sl];
            }
        }
        if ((g + 1) % epoch_gens == 0) {                                   /* the deal */
            const int64_t e = (g + 1) / epoch_gens - 1;
            for (int32_t r = 0; r < P; ++r) {
                int64_t A, C;
                kmco_deal_perm(c->seed, e, r, S, &A, &C);
                for (int64_t j = 0; j < S; ++j) {
                    const int64_t t = (int64_t)(((__int128)A * j + C) % S);
                    const int64_t from = r * S + j, to = (t / chunk) * S + r * chunk + t % chunk;
                    memcpy(pos2 + to * nd, pos + from * nd, sizeof(double) * (size_t)nd);
                    logp2[to] = logp[from]; nacc2[to] = nacc[from]; ids2[to] = ids[from];
                }
            }
            double* tp = pos; pos = pos2; pos2 = tp;
            tp = logp; logp = logp2; logp2 = tp;
            int64_t* ti = nacc; nacc = nacc2; nacc2 = ti;
            ti = ids; ids = ids2; ids2 = ti;
        }
    }
    const double denom = (double)(c->ngenerations - c->nburnin);           /* :291 */
    for (int64_t s = 0; s < N && !bad; ++s) {
        const int64_t w = ids[s];
        if (accept_ratio) accept_ratio[w] = (double)nacc[s] / denom;
        if (naccept_out) naccept_out[w] = nacc[s];
        if (final_pos) memcpy(final_pos + w * nd, pos + s * nd, sizeof(double) * (size_t)nd);
        if (final_logp) final_logp[w] = logp[s];
        if (slot_ids) slot_ids[s] = w;
    }
    if (nmoment) *nmoment = nmom;
    free(pos); free(pos2); free(logp); free(logp2); free(nacc); free(nacc2); free(ids); free(ids2); free(mpart);
    return bad ? KMCO_ERR_NONFINITE_LOGP : KMCO_OK;
}

KMCO_API int kmco_emcee_dealt(const kmco_config* c, int32_t P, int64_t epoch_gens, const double* theta0,
                              double* accept_ratio, int64_t* naccept_out, double* final_pos, double* final_logp,
                              int64_t* slot_ids, double* msum, double* msumsq, int64_t* nmoment)
{
    return kmco_emcee_dealt_chain(c, P, epoch_gens, theta0, accept_ratio, naccept_out, final_pos, final_logp, slot_ids, msum, msumsq,
                                  nmoment, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------
 * MANY-CHAIN METROPOLIS: metropolis / _metropolis, src/samplers.jl:59-128, restated for `nchains`
 * independent chains (the reference runs one; chain c here is exactly one run of the reference's
 * loop with its own random stream).
 *
 * Proposal (the reference takes any symmetric `sample_ppdf`; its tests all use this one,
 * test/runtests.jl:54,59,64,75):  theta1 = theta0 .+ step .* randn(ndim).
 *
 * Random stream (the build's contract, restated independently of the product code):
 * Philox4x32-10, key = {seed_lo ^ 0x4d455452, seed_hi}, counter = {it_lo, it_hi, chain, block},
 * it = 0-based iteration.  Block 0: words (w0,w1) -> Box-Muller pair for dimensions 0,1;
 * ((w2 << 20) | (w3 >> 12) + 1/2) 2^-52 -> the accept uniform.  Block b >= 1: (w0,w1) -> dimensions
 * 4b-2, 4b-1; (w2,w3) -> dimensions 4b, 4b+1.  Pair from words (a,b): u1 = (a+1/2) 2^-32,
 * u2 = (b+1/2) 2^-32, r = sqrt(-2 log u1), n0 = r cos(2 pi u2), n1 = r sin(2 pi u2).
 * log/sin/cos come from the platform libm, so device and host agree to rounding, not bit for bit.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t  density;
    int32_t  nthreads;      /* OpenMP threads over the chains; <=1: serial */
    double   params[8];
    int64_t  nchains;
    int64_t  ndim;
    int64_t  niter;         /* src/samplers.jl:62 */
    int64_t  nburnin;       /* src/samplers.jl:63 */
    int64_t  nthin;         /* src/samplers.jl:64 */
    uint64_t seed;
} kmco_metropolis_config;

static void kmco_normal_pair(uint32_t a, uint32_t b, double* n0, double* n1)
{
    const double u1 = ((double)a + 0.5) * 0x1.0p-32;
    const double u2 = ((double)b + 0.5) * 0x1.0p-32;
    const double r = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    *n0 = r * cos(ang);
    *n1 = r * sin(ang);
}

/* the ndim standard normals and the accept uniform of (iteration it, chain) */
KMCO_API void kmco_metropolis_draw(uint64_t seed, uint64_t it, uint64_t chain, int64_t ndim, double* normals, double* u_acc)
{
    const uint32_t key[2] = {(uint32_t)seed ^ 0x4d455452u, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)it, (uint32_t)(it >> 32), (uint32_t)chain, 0u};
    uint32_t w[4];
    kmco_philox4x32_10(ctr, key, w);
    const uint64_t k = ((uint64_t)w[2] << 20) | (uint64_t)(w[3] >> 12);
    *u_acc = ((double)k + 0.5) * 0x1.0p-52;
    double n0, n1;
    kmco_normal_pair(w[0], w[1], &n0, &n1);
    if (ndim > 0) normals[0] = n0;
    if (ndim > 1) normals[1] = n1;
    for (int64_t b = 1; 4 * b - 2 < ndim; ++b) {
        ctr[3] = (uint32_t)b;
        kmco_philox4x32_10(ctr, key, w);
        double m[4];
        kmco_normal_pair(w[0], w[1], &m[0], &m[1]);
        kmco_normal_pair(w[2], w[3], &m[2], &m[3]);
        for (int q = 0; q < 4; ++q)
            if (4 * b - 2 + q < ndim) normals[4 * b - 2 + q] = m[q];
    }
}

/* Outputs (any may be NULL): chain [nsamples][nchains][ndim] (thetas, :113), chain_logp [nsamples][nchains]
 * (logdensities, :115), accept_ratio [nchains] (:127), naccept [nchains], final_pos, final_logp, chain_sum /
 * chain_sumsq [nchains][ndim] (per-chain sums over the stored samples).  nsamples = (niter - nburnin) / nthin (:88). */
KMCO_API int kmco_metropolis(const kmco_metropolis_config* c, const double* theta0, const double* step,
                             double* chain, double* chain_logp, double* accept_ratio, int64_t* naccept_out,
                             double* final_pos, double* final_logp, double* chain_sum, double* chain_sumsq)
{
    if (!c || !theta0 || !step || c->nchains <= 0 || c->ndim <= 0 || c->nthin <= 0 || c->niter < 0 || c->nburnin < 0)
        return KMCO_ERR_BAD_ARG;
    if (c->density == KMCO_ROSENBROCK && c->ndim < 2) return KMCO_ERR_BAD_ARG;
    if (c->density == KMCO_MVNORMAL2 && c->ndim != 2) return KMCO_ERR_BAD_ARG;
    const int64_t nc = c->nchains, nd = c->ndim;
    const int64_t nsamples = c->niter > c->nburnin ? (c->niter - c->nburnin) / c->nthin : 0;   /* :88 */

#pragma omp parallel for schedule(static) num_threads(c->nthreads > 1 ? c->nthreads : 1)
    for (int64_t ch = 0; ch < nc; ++ch) {
        double* th0 = (double*)malloc(sizeof(double) * (size_t)nd * 3);
        double* th1 = th0 + nd;
        double* nrm = th1 + nd;
        memcpy(th0, theta0 + ch * nd, sizeof(double) * (size_t)nd);                 /* :68 deepcopy */
        double p0 = kmco_logpdf(c->density, c->params, th0, nd);                     /* :70 */
        int64_t naccept = 0;                                                         /* :94 */
        int64_t k = 0;
        if (chain_sum) for (int64_t d = 0; d < nd; ++d) chain_sum[ch * nd + d] = 0.0;
        if (chain_sumsq) for (int64_t d = 0; d < nd; ++d) chain_sumsq[ch * nd + d] = 0.0;
        for (int64_t n = 1 - c->nburnin; n <= c->niter - c->nburnin; ++n) {         /* :96 */
            const uint64_t it = (uint64_t)(n + c->nburnin - 1);
            double ua;
            kmco_metropolis_draw(c->seed, it, (uint64_t)ch, nd, nrm, &ua);
            for (int64_t d = 0; d < nd; ++d) th1[d] = fma(step[d], nrm[d], th0[d]);  /* :98 theta1 = sample_ppdf(theta0) */
            const double p1 = kmco_logpdf(c->density, c->params, th1, nd);          /* :99 */
            if (p1 - p0 > log(ua)) {                                                 /* :101 strict > */
                memcpy(th0, th1, sizeof(double) * (size_t)nd);                      /* :102 */
                p0 = p1;                                                             /* :104 */
                naccept += 1;                                                        /* :105 */
            }
            if (n % c->nthin == 0) {                                                 /* :108 rem(n, nthin) == 0 */
                if (n > 0 && k < nsamples) {                                         /* :112 after burn-in */
                    if (chain) memcpy(chain + (k * nc + ch) * nd, th0, sizeof(double) * (size_t)nd);   /* :113 */
                    if (chain_logp) chain_logp[k * nc + ch] = p0;                    /* :115 */
                    for (int64_t d = 0; d < nd; ++d) {
                        if (chain_sum) chain_sum[ch * nd + d] += th0[d];
                        if (chain_sumsq) chain_sumsq[ch * nd + d] += th0[d] * th0[d];
                    }
                    ++k;
                }
            }
            if (n == 0) naccept = 0;                                                 /* :122-125 */
        }
        if (accept_ratio) accept_ratio[ch] = (double)naccept / (double)(c->niter - c->nburnin);   /* :127 */
        if (naccept_out) naccept_out[ch] = naccept;
        if (final_pos) memcpy(final_pos + ch * nd, th0, sizeof(double) * (size_t)nd);
        if (final_logp) final_logp[ch] = p0;
        free(th0);
    }
    return KMCO_OK;
}

KMCO_API int kmco_sizeof_metropolis_config(void) { return (int)sizeof(kmco_metropolis_config); }
