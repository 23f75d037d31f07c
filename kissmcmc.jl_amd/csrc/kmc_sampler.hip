// kmc_sampler.hip -- the sampler's lifecycle behind the C ABI (include/kissmcmc_hip.h): validation of a configuration
// (the reference's asserts, src/samplers.jl:200-205), creation / destruction of the device state of the `emcee` front-end
// (src/samplers.jl:188-216), and the description of how a sampler executes.  Kernel tables and the launch plan per ndim are kmc_plan.hip.
// The generation loop (src/samplers.jl:232-293) is kmc_launch.hip; state in and out is kmc_state.hip; copies and the chain
// ring are kmc_copy.hip; runtime-compiled densities kmc_rtc.hip; multi-GPU wiring kmc_p2p.hip; diagnostics kmc_diag.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <sstream>

#include "kmc_sampler.hpp"

using namespace kmc;
using namespace kmc_host;

thread_local std::string kmc_host::g_err;

KMC_EXPORT int kmc_version(void) { return KMC_VERSION; }

KMC_EXPORT int kmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

KMC_EXPORT const char* kmc_last_error(void) { return g_err.c_str(); }

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
    if (c->flags & ((1u << 8) | (1u << 10)))                 // (KMC_P2P_FOLD_SIGNAL, KMC_P2P_LAZY of rounds 1-4)
        return fail(KMC_ERR_UNSUPPORTED, "the lazy-pull and folded-signal exchange variants were removed in round 5 (they read peer-written memory through the local L2 and could "
                                         "not move the fabric bound): the exchanges are the pull of drawn rows and the push of accepted rows (KMC_P2P_PUSH), both read with "
                                         "system-scope loads");
    if ((c->flags & KMC_P2P_PUSH) && !(c->flags & KMC_P2P)) return fail(KMC_ERR_BAD_ARG, "KMC_P2P_PUSH needs KMC_P2P");
    if (c->flags & KMC_ISLANDS) {
        const int64_t S = c->island_size > 0 ? c->island_size : kIslandSizeDefault;
        if ((S != 64 && S != 128 && S != 256) || c->nwalkers % S != 0 || c->ndim > 32 || c->ndim + 2 > S || P != 1 ||
            (c->flags & (KMC_P2P | KMC_STORE_CHAIN | KMC_STORE_LOGP)) || c->island_gens < 0)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_ISLANDS needs island_size in {64,128,256} >= ndim+2 dividing nwalkers, ndim <= 32, one shard and no chain storage");
        if (c->density == KMC_USER_DENSITY && (size_t)S * (size_t)(2 * c->ndim + 9) * sizeof(double) > 156 * 1024)
            return fail(KMC_ERR_UNSUPPORTED, "KMC_ISLANDS with a user density: island_size * (2 ndim + 9) * 8 must stay below 156 KiB");
    }
    if (c->flags & KMC_STREAM_CHAIN) {
        if (!(c->flags & (KMC_STORE_CHAIN | KMC_STORE_LOGP))) return fail(KMC_ERR_BAD_ARG, "KMC_STREAM_CHAIN needs KMC_STORE_CHAIN and / or KMC_STORE_LOGP");
        if (c->dtype != KMC_F64 || P != 1 || (c->flags & (KMC_P2P | KMC_ISLANDS)))
            return fail(KMC_ERR_UNSUPPORTED, "KMC_STREAM_CHAIN: KMC_F64, one GPU, without KMC_P2P / KMC_ISLANDS / sharding");
    }
    {
        const int nb = c->density == KMC_USER_DENSITY && c->user_density ? static_cast<const kmc_user_density*>(c->user_density)->nblob : 0;
        if ((c->flags & KMC_STORE_BLOBS) && nb == 0)
            return fail(KMC_ERR_BAD_ARG, "KMC_STORE_BLOBS needs a body density with blobs (kmc_user_density_create_body_blob)");
        if (nb > 0 && (c->dtype != KMC_F64 || P != 1 || c->deal_count > 0 || (c->flags & (KMC_P2P | KMC_ISLANDS | KMC_STREAM_CHAIN))))
            return fail(KMC_ERR_UNSUPPORTED, "a density with blobs: KMC_F64, one GPU, without KMC_P2P / KMC_ISLANDS / KMC_STREAM_CHAIN / sharding / dealt sub-ensembles");
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

// hipMalloc does not clear memory.  KMC_DEBUG=poison (diagnostics): every allocation of a sampler starts as 0xFF bytes (NaN doubles,
// 4 294 967 295 counters), so that anything the code forgot to initialise shows up in the tests instead of depending on what the
// allocator happened to return.
// ... and is followed by a 4 KiB guard of 0xA5 that kmc_sampler_destroy checks: a kernel that writes past the end of one of
// its buffers aborts the process there, with the size of the allocation (the tests then fail loudly).
template <class T>
hipError_t dev_alloc(kmc_sampler* s, T** p, size_t bytes)
{
    static const bool poison = debug_opt("poison");
    if (!poison || bytes == 0) {
        // (IPC-exported buffers stay plain allocations: a peer maps them by their base address)
        const hipError_t e0 = (s->cfg.flags & KMC_P2P) ? hipMalloc(reinterpret_cast<void**>(p), bytes) : cache_alloc(reinterpret_cast<void**>(p), bytes);
        return e0;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes + kGuardBytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(*p, 0xFF, bytes, s->stream);
    if (e == hipSuccess) e = hipMemsetAsync(reinterpret_cast<char*>(*p) + bytes, 0xA5, kGuardBytes, s->stream);
    s->guards.emplace_back(reinterpret_cast<char*>(*p) + bytes, bytes);
    return e;
}

namespace {
// A body that kmc_user_density_create_body recognised as a sum over elements (kmc_rtc.hip: recognise_separable -- a TEXT matcher) is
// run in its generated per-element form only after that form has been evaluated next to the body itself: 256 test rows of the
// sampler's ndim (normal / half-normal / uniform entries on several scales, so that densities on the line, the half-line and the
// unit box all see admissible points), both through the sequential log-pdf kernels of the module just loaded.  They must agree on
// every row (both non-finite alike, or within 1e-9 relative) and at least 8 rows must carry a finite value; otherwise the density's
// density loses its routing.  Once per (ndim, parameter values) of a density -- what agrees at one row length and one parameter set says nothing
// about another: a case where the forms DISAGREE refutes the recogniser (`sep` cleared for every later sampler, describe() says why);
// a case where too few rows are finite shows nothing: THIS sampler runs the body as written (kmc_sampler::sep_off), the density keeps its routing.
// KMC_DEBUG=sum-form-check=fail / =blind force the two negative outcomes (tests).
uint64_t sum_form_case_digest(const kmc_sampler* s)
{
    uint64_t h = 0xcbf29ce484222325ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(s->dp.p);
    for (size_t i = 0; i < sizeof(s->dp.p); ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
// 0: not checked yet for this sampler's (ndim, parameters); 1: agreed; 3: blind
int sum_form_case(kmc_sampler* s)
{
    kmc_user_density* ud = s->user;
    std::lock_guard<std::mutex> lock(ud->mu);
    auto it = ud->sep_case.find({s->cfg.ndim, sum_form_case_digest(s)});
    return it == ud->sep_case.end() ? 0 : it->second;
}
kmc_status check_sum_form(kmc_sampler* s)
{
    kmc_user_density* ud = s->user;
    constexpr int kRows = 256, kFamilies = 8;
    const int64_t nd = s->cfg.ndim, ld = s->ld;
    std::vector<double> rows((size_t)kRows * (size_t)ld, 0.0);
    std::mt19937_64 gen(0x6b6d632d73756d21ull);
    std::normal_distribution<double> normal(0.0, 1.0);
    std::uniform_real_distribution<double> unit(0.0, 1.0);
    for (int r = 0; r < kRows; ++r) {
        const int fam = r % kFamilies;
        for (int64_t d = 0; d < nd; ++d) {
            const double z = normal(gen), u = unit(gen);
            double v = z;
            switch (fam) {
                case 1: v = 0.01 * z; break;
                case 2: v = 10.0 * z; break;
                case 3: v = std::fabs(z); break;
                case 4: v = 0.01 * std::fabs(z); break;
                case 5: v = 10.0 * std::fabs(z); break;
                case 6: v = u; break;
                case 7: v = -std::fabs(z); break;
                default: break;
            }
            rows[(size_t)r * (size_t)ld + (size_t)d] = v;
        }
    }
    // (a stream of its own and copy_sync: nothing here may touch the legacy stream -- another thread may be capturing a graph, kmc_host.hpp)
    double *d_rows = nullptr, *d_out = nullptr;
    std::vector<double> out(2 * kRows);
    ScopedStream ss;
    hipError_t e = ss.create();
    if (e == hipSuccess) e = hipMalloc((void**)&d_rows, rows.size() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, out.size() * sizeof(double));
    if (e == hipSuccess) e = copy_sync(d_rows, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, ss.st);
    if (e == hipSuccess) {
        const LogpdfArgs la{d_rows, d_out, (int64_t)kRows, (int32_t)nd, (int32_t)ld, s->dp, nullptr};
        e = launch_module(s->uk.logpdf, 1u, 256u, ss.st, la);
    }
    if (e == hipSuccess) {
        const LogpdfArgs lb{d_rows, d_out + kRows, (int64_t)kRows, (int32_t)nd, (int32_t)ld, s->dp, nullptr};
        e = launch_module(s->uk.logpdf_sep, 1u, 256u, ss.st, lb);
    }
    if (e == hipSuccess) e = copy_sync(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost, ss.st);
    if (ss.st) (void)hipStreamSynchronize(ss.st);
    if (d_rows) (void)hipFree(d_rows);
    if (d_out) (void)hipFree(d_out);
    HIP_TRY(e);
    int informative = 0, differ = 0, first = -1;
    for (int r = 0; r < kRows; ++r) {
        const double a = out[r], b = out[kRows + r];
        const bool fa = std::isfinite(a), fb = std::isfinite(b);
        bool same;
        if (!fa || !fb) same = (fa == fb) && (std::isnan(a) == std::isnan(b)) && (std::isnan(a) || (a < 0) == (b < 0));
        else { same = std::fabs(a - b) <= 1e-9 * std::fmax(1.0, std::fmax(std::fabs(a), std::fabs(b))); ++informative; }
        if (!same) { if (first < 0) first = r; ++differ; }
    }
    std::string forced;
    if (debug_opt("sum-form-check", &forced)) {
        if (forced == "fail") { differ = 1; first = 0; }
        if (forced == "blind") { differ = 0; informative = 0; }
    }
    std::lock_guard<std::mutex> lock(ud->mu);
    const std::pair<int64_t, uint64_t> key{s->cfg.ndim, sum_form_case_digest(s)};
    if (differ == 0 && informative >= 8) { ud->sep_case[key] = 1; return KMC_OK; }
    char why[256];
    if (differ) {
        std::snprintf(why, sizeof(why), "its per-element form disagrees with the body on %d of %d test rows (row %d: body %.17g, per-element form %.17g)",
                      differ, kRows, first, out[first], out[kRows + first]);
        ud->sep_note = why;
        ud->sep = false;                         // refuted: no later sampler routes this body
    } else {
        std::snprintf(why, sizeof(why), "only %d of %d test rows had a finite log-density at ndim %lld with these parameters: its per-element form could not be checked",
                      informative, kRows, (long long)s->cfg.ndim);
        ud->sep_case[key] = 3;                   // nothing shown HERE: this sampler runs the body as written; other cases are checked on their own
        s->sep_off = true;
        s->sep_off_note = why;
    }
    if (differ)
        std::fprintf(stderr, "kissmcmc_hip: a function body was recognised as a sum over elements, but %s -- it is evaluated per walker, as written "
                             "(please report the body; KMC_DEBUG=no-body-routing skips the attempt)\n", why);
    return KMC_OK;
}
}  // namespace

namespace {
// One launch per generation (kmc_generation.hpp) -- the dependent-launch boundary, which is most of a half-step for small states, is paid
// once per generation instead of twice.  0: no; 1: one walker per lane (ndim <= 8); 2: rows lane-striped like the vector kernels (longer
// rows; lane-striped densities).  Measured against the two-launch kernels (profiles/r04_generation_map.txt): short rows 1.3-1.9 x ahead up
// to 32 768 walkers, 1.25-1.4 x at 65 536 walkers of one or two doubles, behind beyond that; longer rows 1.15-1.55 x ahead while the state
// stays within ~2.3 MiB, 1.25-1.4 x at 4 MiB (8 192 x 64, 32 768 x 16, 16 384 x 32), 1.06-1.3 x between 4 and 8 MiB (C3 1.17 x, 16 384 x 48 1.33, 24 576 x 32 1.18,
// 32 768 x 32 1.06, 49 152 x 16 1.10) except with 65 536 walkers (x 16: 0.94), mixed at 10 MiB (0.91-1.07) and behind from 12 MiB on -- the kernel moves 1.5 x the
// walkers and reads 2.5 x the rows (profiles/r05_generation_mid.txt, profiles/r05_generation_limit.txt: measured again after the row masks went and the stores became
// write-through; C3 was 0.95 x before both).
// KMC_DEBUG=fused=0 / =1: never / wherever a kernel exists.  (Resident and island mode are decided by the caller.)
int generation_wanted(const kmc_sampler* s)
{
    const kmc_config& c = s->cfg;
    if (c.density == KMC_HOST_DENSITY || s->f32 || s->nblob != 0 || c.shard_count != 1 || c.deal_count != 0 ||
        (c.flags & (KMC_P2P | KMC_NO_GRAPH | KMC_ISLANDS)) || std::getenv("KMC_PLAN") != nullptr)      // (KMC_PLAN: a geometry of the two-launch kernels was asked for)
        return 0;
    // lane-striped forms need a lane-striped density (menu, term / pair, a body recognised as a sum) and the vector kernels' plan
    const bool striped = s->plan.vec && !(s->user && s->user->is_body && !(s->user->sep && !s->sep_off));
    // ndim <= 8: one walker per lane -- or, from 5 dimensions on, the row over a quad (generation_group<4, 1>: the quad shares the draws' four
    // logarithms; measured 1.65 against 1.95 us per half-step at 4 096 x 8, 1.65 against 1.76 at 4 096 x 6, behind at 16 384 x 6)
    int kind = c.ndim <= 8 ? 1 : 2;
    if (kind == 2 && !striped) return 0;
    std::string forced;
    const bool have = debug_opt("fused", &forced);
    if (have && forced == "0") return 0;
    if (kind == 1 && c.ndim >= 5 && striped && forced != "lane" && (have || c.nwalkers <= (s->ld == 8 ? 49152 : 8192))) kind = 3;     // (49 152 x 8: 1.12 x the two launches; 65 536 x 8: 1.01)
    if (have) return kind;                                   // (=1 / =lane: wherever a kernel exists; =lane keeps short rows one walker per lane)
    if (kind == 3) return 3;
    if (kind == 1) return ((c.nwalkers <= 49152 && c.nwalkers * s->ld <= 196608) || (c.nwalkers <= 65536 && s->ld <= 2)) ? 1 : 0;       // (49 152 x 4: 1.09 x; 65 536 x 3 / x 4: 1.03 / 0.99)
    return c.nwalkers * s->ld <= (c.nwalkers <= 49152 ? 1048576 : 524288) ? 2 : 0;     // (8 MiB of state up to 49 152 walkers, 4 MiB beyond)
}
}  // namespace

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
        s->nblob = s->user->nblob;
        // small ensembles: resident mode too (one workgroup, LDS within the default 64 KiB limit)
        int rK = 0, rK0 = 1;
        while (2 * rK0 < s->ld / 2) rK0 *= 2;
        size_t rlds = ((size_t)cfg->nwalkers * (size_t)(4 * (rK0 + 1)) + (size_t)cfg->nwalkers) * sizeof(double);
        // a body density: one walker per thread (kmc_islands.hpp: resident_lane_body), rows of ndim | 1 doubles, up to 1024 walkers
        if (s->user->is_body) rlds = ((size_t)cfg->nwalkers * (size_t)((cfg->ndim | 1) + 1)) * sizeof(double);
        const bool expr_lane = !s->user->is_body && resident_lane_wanted(cfg->ndim);     // term / pair density, short rows: the lane kernel too
        if (expr_lane) rlds = ((size_t)cfg->nwalkers * (size_t)((lane_nd(cfg->ndim) | 1) + 1)) * sizeof(double);
        // (float rows: the one-walker-per-thread kernels only -- their LDS rows are double either way)
        // 1026 .. 2048 walkers: two walkers per thread (kmc_islands.hpp: resident_lane2_body; double rows, ndim <= 8, no blobs)
        const bool lane2 = cfg->nwalkers > 1024 && cfg->nwalkers <= 2048 && !s->f32 && cfg->ndim <= 8 && s->user->nblob == 0 && resident_lane_wanted(cfg->ndim);
        if (lane2) rlds = ((size_t)cfg->nwalkers * (size_t)((cfg->ndim | 1) + 1)) * sizeof(double);
        if ((!s->f32 || s->user->is_body || expr_lane) && (cfg->nwalkers <= ((s->user->is_body || expr_lane) ? 1024 : 256) || lane2) && cfg->ndim <= 32 && s->cfg.shard_count == 1 && !(s->user->is_body && cfg->deal_count > 0) &&
            !(cfg->flags & (KMC_P2P | KMC_NO_GRAPH | KMC_ISLANDS)) &&
            rlds <= 156 * 1024 && !debug_opt("no-resident"))    // (hipModuleLaunchKernel takes dynamic LDS beyond 64 KiB as it is)
            rK = rK0;
        int iS = 0;
        if (cfg->flags & KMC_ISLANDS) {
            iS = cfg->island_size > 0 ? cfg->island_size : kIslandSizeDefault;
            rK = 1;
            while (2 * rK < s->ld / 2) rK *= 2;
        }
        if (s->user->is_body && iS > 0) { kmc_sampler_destroy(s); return fail(KMC_ERR_UNSUPPORTED, "KMC_ISLANDS needs a menu or term / pair density (a body density runs one walker per lane)"); }
        if (s->user->is_body && cfg->ndim > 1024) { kmc_sampler_destroy(s); return fail(KMC_ERR_UNSUPPORTED, "a body density holds the proposal per lane: ndim <= 1024"); }
        // (what the resident kernel is compiled as: term / pair density on short rows -> -ndim = one walker per thread; body density ->
        //  the workgroup size bound, 256 or 1024: a body of 32 dimensions needs its registers; else the two-lane kernel's K)
        //  two walkers per thread: 2048 for a body, -(100 + ndim) for a term / pair density)
        const int rcode = (rK > 0 && iS == 0) ? (lane2 ? (s->user->is_body ? 2048 : -(100 + (int)cfg->ndim))
                                                       : s->user->is_body ? (cfg->nwalkers <= 256 ? 256 : cfg->nwalkers <= 512 ? 512 : 1024) : (expr_lane ? -lane_nd(cfg->ndim) : rK)) : rK;
        // a big ensemble's kernels are worth the better compiler (hipcc as a child process: ~1.2 s once per density and geometry, then
        // cached on disk): inside a PyTorch process hiprtc means the older comgr the wheel bundles (kmc_rtc.hip: offline_compiler_wanted)
        // the first sampler over a body taken for a sum over elements finds out whether it is one (check_sum_form); samplers created
        // meanwhile on other threads wait for the answer instead of planning on a guess
        std::unique_lock<std::mutex> first_use;
        if (s->user->is_body && s->user->sep && sum_form_case(s) == 0) {
            first_use = std::unique_lock<std::mutex>(s->user->check_mu);
            s->plan = make_plan(s->cfg, s->h_loc);
        }
        if (s->user->is_body && s->user->sep && sum_form_case(s) == 3) {       // (an earlier sampler of this very case found nothing to check the form against)
            s->sep_off = true;
            s->sep_off_note = "no test row had a finite log-density at this ndim with these parameters: its per-element form could not be checked";
        }
        SepOff sep_off_scope(s->sep_off);        // (for make_plan / compile_user / load_user below, which see the density, not the sampler; the rest of the set-up reads s->sep_off)
        if (s->sep_off) s->plan = make_plan(s->cfg, s->h_loc);
        auto load = [&]() {
            set_offline_compiler_hint(s->h_loc >= 8192 && rK == 0 && iS == 0);
            const kmc_status lst = load_user(s->user, s->plan.vec, s->plan.L, s->plan.K, s->plan.ITER, s->plan.ragged, &s->uk, rcode, 4 * rK != cfg->ndim, iS, s->f32,
                                             cfg->ndim, (cfg->flags & KMC_P2P) != 0, (rK != 0 || iS != 0) ? 0 : generation_wanted(s) == 1 ? (int)cfg->ndim : generation_wanted(s) == 2 ? -(100 * s->plan.L + s->plan.K) : generation_wanted(s) == 3 ? -401 : 0);
            set_offline_compiler_hint(false);
            return lst;
        };
        st = load();
        // a body taken for a sum over elements: its generated form against the body itself, once, before anything runs it
        if (st == KMC_OK && s->uk.logpdf_sep && sum_form_case(s) != 1) {
            st = check_sum_form(s);
            if (st == KMC_OK && (!s->user->sep || s->sep_off)) {          // not shown equal: plan and kernels for the body as written
                g_sep_off = g_sep_off || s->sep_off;                      // (restored by sep_off_scope at the end of this block)
                s->uk = UserKernels{};
                s->plan = make_plan(s->cfg, s->h_loc);
                st = load();
            }
        }
        if (st != KMC_OK) { kmc_sampler_destroy(s); return st; }
        if (iS > 0) rK = 0;     // island mode is set up below, not resident mode
        if (rK > 0) {
            s->resident = true;
            s->island_K = rK;
            s->nislands = 1;
            s->island_lds = rlds < 4096 ? 4096 : rlds;
            s->resident_lane = s->user->is_body || expr_lane || lane2;
            s->resident_lane2 = lane2;
            s->resident_tpb = lane2 ? (int)((cfg->nwalkers / 2 + 63) / 64 * 64) : s->resident_lane ? (int)((cfg->nwalkers + 63) / 64 * 64) : 256;
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
    if (!s->islands && cfg->density != KMC_USER_DENSITY && !s->host_eval && cfg->nwalkers <= 2048 && cfg->ndim <= 32 &&
        s->cfg.shard_count == 1 && !(cfg->flags & (KMC_P2P | KMC_NO_GRAPH)) && !debug_opt("no-resident")) {
        const int64_t chunks = s->ld / 2;
        int K = 1;
        while (2 * K < chunks) K *= 2;
        int rtpb = cfg->nwalkers <= 256 ? 256 : (cfg->nwalkers <= 512 ? 512 : 1024);
        size_t need = ((size_t)cfg->nwalkers * (size_t)(4 * (K + 1)) + (size_t)cfg->nwalkers) * sizeof(double);
        ResidentFn rf = (!s->f32 && need <= 156 * 1024 && cfg->nwalkers <= 1024) ? resident_fn(cfg->density, rtpb, K, 4 * K != cfg->ndim) : nullptr;   // (float rows: the lane kernels only)
        if (cfg->nwalkers > 1024) {                       // 1026 .. 2048 walkers: two walkers per thread, short double rows, as LDS allows
            const size_t need2 = ((size_t)cfg->nwalkers * (size_t)((lane_nd(cfg->ndim) | 1) + 1)) * sizeof(double);
            ResidentFn l2 = (!s->f32 && resident_lane_wanted(cfg->ndim) && need2 <= 156 * 1024) ? resident_lane2_fn(cfg->density, (int)cfg->ndim) : nullptr;
            if (l2) {
                rf = l2;
                rtpb = (int)((cfg->nwalkers / 2 + 63) / 64 * 64);
                need = need2;
                s->resident_lane = true;
                s->resident_lane2 = true;
            }
        } else
        if (resident_lane_wanted(cfg->ndim)) {            // short rows: one walker per thread (measured faster up to ndim 8)
            ResidentFn lf = resident_lane_fn(cfg->density, (int)cfg->ndim, s->f32);
            if (lf) {
                rf = lf;
                rtpb = (int)((cfg->nwalkers + 63) / 64 * 64);
                need = ((size_t)cfg->nwalkers * (size_t)((lane_nd(cfg->ndim) | 1) + 1)) * sizeof(double);
                s->resident_lane = true;
            }
        }
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
    if (const int kind = (!s->islands && !s->resident) ? generation_wanted(s) : 0) {          // one launch per generation (see generation_wanted)
        if (s->user) s->fused = s->uk.generation != nullptr;
        else {
            s->generation_kernel = kind == 1 ? generation_fn(cfg->density, (int)cfg->ndim) : kind == 3 ? generation_group_fn(cfg->density, 4, 1)
                                                                                             : generation_group_fn(cfg->density, s->plan.L, s->plan.K);
            s->fused = s->generation_kernel != nullptr;
        }
        if (s->fused) {
            s->fused_L = kind == 1 ? 0 : kind == 3 ? 4 : s->plan.L;
            // one wave per workgroup (measured best at every size for one walker per lane, `profiles/r04_generation_variants_ab.txt`); lane-striped:
            // two once the half's waves exceed the chip's SIMDs about once
            s->fused_tpb = (kind == 2 && s->h * s->plan.L > 64 * 512) ? 128 : 64;
            s->nislands = cfg->nwalkers;                  // per-walker moment sums [nwalkers][ld], within [nislands][4 island_K] (kmc_sampler_get_moments)
            s->island_K = (int)((s->ld + 3) / 4);
            s->fused_fold = s->fused_L > 0 && s->plan.K == 2 && (s->fused_L == 8 || s->fused_L == 16 || s->fused_L == 32) && kind == 2;
            if (s->fused_fold) {                          // ... or per-wave accumulators [waves][NVL][64] (generation_group, FoldT): the same sizing fields
                const int64_t per_wg = s->fused_tpb / s->fused_L;
                s->nislands = 2 * ((s->h + per_wg - 1) / per_wg) * (s->fused_tpb / 64);      // waves of a launch
                s->island_K = (8 * s->fused_L / 64) * 16;                                     // 4 island_K = NVL * 64 doubles per wave
            }
        }
    }
    // vec kernels: vec_tpb(L) threads per workgroup; the generic kernel keeps 256
    const bool staged = s->user && s->uk.staged != nullptr;         // a body density's staged kernel: two waves per workgroup
    s->vec_lds = (s->user && s->user->is_body && !(s->user->sep && !s->sep_off) && s->plan.vec) ? (unsigned)body_vec_lds_bytes(s->plan.L, s->plan.K, s->plan.ITER) : 0u;
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
        // KMC_P2P_PUSH: local copies of the other shards, which their owners write; the menu densities' vector kernels carry it (anything
        // else keeps the pull -- kmc_sampler_describe says which runs)
        s->push = (cfg->flags & KMC_P2P_PUSH) != 0 && s->plan.vec && s->user == nullptr && !(cfg->flags & KMC_P2P_FINEGRAINED) &&
                  s->cfg.shard_count > 1;
    }
    const size_t ldz = (size_t)s->ld;
    const size_t esz = s->f32 ? sizeof(float) : sizeof(double);      // element size of rows and chain
    if (s->p2p && (cfg->flags & KMC_P2P_FINEGRAINED))   // peers map the rows uncached: nothing of them can go stale in a reader's L2
        CREATE_TRY(hipExtMallocWithFlags((void**)&s->d_pos, nw * ldz * sizeof(double), hipDeviceMallocFinegrained));
    else
        CREATE_TRY(dev_alloc(s, &s->d_pos, (s->push ? 1 + (size_t)s->cfg.shard_count : 1) * nw * ldz * esz ));
    CREATE_TRY(hipMemsetAsync(s->d_pos, 0, (s->push ? 1 + (size_t)s->cfg.shard_count : 1) * nw * ldz * esz, s->stream));   // the pad column of odd ndim stays 0
    // per-walker block {logp[nrows], naccept[nrows], klast[nrows]}: one allocation, so the half-step kernels reach all
    // three from one preloaded pointer (HalfStepFront::logp)
    CREATE_TRY(dev_alloc(s, &s->d_logp, nw * (sizeof(double) + 2 * sizeof(uint32_t))));
    s->d_naccept = reinterpret_cast<uint32_t*>(s->d_logp + nw);
    s->d_klast = s->d_naccept + nw;
    if (s->ragged_vec() && (reinterpret_cast<uint64_t>(s->d_logp) >> 48) != 0) {   // (kmc_launch.hip: front_of -- ndim rides in the 16 bits above this address)
        kmc_sampler_destroy(s);
        return fail(KMC_ERR_UNSUPPORTED, "device addresses beyond 2^48: the ragged vector kernels carry ndim in the upper 16 bits of a pointer");
    }
    CREATE_TRY(hipMemsetAsync(s->d_klast, 0, nw * sizeof(uint32_t), s->stream));
    static_assert(kGraphChunk <= 64, "advance_schedule runs one 64-thread block");
    CREATE_TRY(dev_alloc(s, &s->d_gen, 64));
    CREATE_TRY(hipMemsetAsync(s->d_gen, 0, 64, s->stream));
    CREATE_TRY(dev_alloc(s, &s->d_sched, (size_t)kGraphChunk * sizeof(SchedEntry)));
    CREATE_TRY(hipMemsetAsync(s->d_naccept, 0, nw * sizeof(uint32_t), s->stream));
    if (cfg->deal_count > 0) CREATE_TRY(dev_alloc(s, (void**)&s->d_ids, nw * sizeof(uint32_t)));
    // Draw ring (kmc_kernels.hpp: kRing; possible for L = 16 / 32 with L / ITER >= 2): a wave computes its walkers' next steps' draws with its idle lanes and parks
    // them, 4 slots x rows x 32 B; tags start at 0xffffffff (no step carries it), so nothing is "parked" yet.  Where it pays was re-measured in round 5
    // (scripts/probes/ring_ab.py, profiles/r05_ring_ab.txt) -- since the launches carry their step among the preloaded parameters Philox starts at wave entry and the
    // ring's entry is one more dependent load in front of the partner row: rows LOSE 2-7 % with it at ITER = 2 and at L = 32, exact-size and ragged alike (the ragged
    // kernels gained 6-10 % from it only while their loads sat behind exec-mask regions and scalar waits; without those they behave like the exact-size ones), and
    // 6-14 % in the large-ensemble geometries (ITER >= 4: bandwidth-bound, the ring is bytes); L = 16, ITER = 1 (C3's geometry) is +-1 % exact-size, -1..2 % ragged, and keeps it.
    // KMC_DEBUG=ring=0|1 forces it off / on wherever the kernel has one.
    bool want_ring = s->plan.vec && !s->islands && !s->resident && s->plan.L >= 16 && s->plan.L <= 32 && s->plan.L / s->plan.ITER >= 2;
    {
        std::string forced;
        if (want_ring && debug_opt("ring", &forced)) want_ring = forced != "0";
        else if (want_ring) want_ring = s->plan.L == 16 && s->plan.ITER == 1;
    }
    if (want_ring) {
        const size_t nb = 4 * (size_t)s->nrows * 2 * sizeof(double2);
        CREATE_TRY(dev_alloc(s, (void**)&s->d_ring, nb));
        CREATE_TRY(hipMemsetAsync(s->d_ring, 0xff, nb, s->stream));
    }
    if (s->resident)                                 // the launch's draws come from a wide kernel (draw_table_fill, kmc_islands.hpp)
        CREATE_TRY(dev_alloc(s, (void**)&s->d_draws, (size_t)kDrawTableGens * nw * 2 * sizeof(double2)));
    if (cfg->flags & KMC_MOMENTS) {
        CREATE_TRY(dev_alloc(s, &s->d_msum, (size_t)s->macc_elems * sizeof(double)));
        CREATE_TRY(dev_alloc(s, &s->d_msumsq, (size_t)s->macc_elems * sizeof(double)));
        CREATE_TRY(hipMemsetAsync(s->d_msum, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        CREATE_TRY(hipMemsetAsync(s->d_msumsq, 0, (size_t)s->macc_elems * sizeof(double), s->stream));
        if (s->plan.vec && s->plan.L == 64 && !s->islands && !s->resident && !debug_opt("no-moment-ring")) {
            // moment ring for long rows (kmc_kernels.hpp, HalfStepArgs::mring): up to 128 posted rows per wave,
            // within 512 MiB in all; swept every kSweepEvery generations
            const int64_t nwaves = s->macc_stride / 64;
            const size_t slot = (size_t)s->plan.K * 64 * sizeof(double2);       // one row
            int64_t depth = (int64_t)(((size_t)1 << 29) / ((size_t)nwaves * slot));
            if (depth > 128) depth = 128;
            const long forced_depth = debug_opt_long("moment-ring-depth", 0);                         // tests: force overflows
            if (forced_depth >= 1 && forced_depth < depth) depth = forced_depth;
            if (depth >= 4 || (depth >= 2 && forced_depth > 0)) {
                CREATE_TRY(dev_alloc(s, (void**)&s->d_mring, (size_t)nwaves * (size_t)depth * slot));
                CREATE_TRY(dev_alloc(s, (void**)&s->d_mring_w, (size_t)nwaves * (size_t)depth * sizeof(double)));
                CREATE_TRY(dev_alloc(s, (void**)&s->d_mcnt, 2 * (size_t)nwaves * sizeof(uint32_t)));
                CREATE_TRY(hipMemsetAsync(s->d_mcnt, 0, 2 * (size_t)nwaves * sizeof(uint32_t), s->stream));
                s->mring_depth = (int)depth;
                s->mring_waves = nwaves;
            }
        }
        if (s->islands || s->resident || s->fused) {
            const size_t ne = (size_t)s->nislands * 4 * (size_t)s->island_K;
            CREATE_TRY(dev_alloc(s, &s->d_isum, ne * sizeof(double)));
            CREATE_TRY(dev_alloc(s, &s->d_isumsq, ne * sizeof(double)));
            CREATE_TRY(hipMemsetAsync(s->d_isum, 0, ne * sizeof(double), s->stream));
            CREATE_TRY(hipMemsetAsync(s->d_isumsq, 0, ne * sizeof(double), s->stream));
        }
    }
    if (s->fused) {                                  // the second copy of the state (pad column of an odd ndim: 0 in both)
        CREATE_TRY(dev_alloc(s, &s->d_pos2, nw * ldz * sizeof(double)));
        CREATE_TRY(hipMemsetAsync(s->d_pos2, 0, nw * ldz * sizeof(double), s->stream));
        CREATE_TRY(dev_alloc(s, &s->d_logp2, nw * sizeof(double)));
        if (s->fused_L > 0) {
            CREATE_TRY(dev_alloc(s, (void**)&s->d_glast, nw * sizeof(uint32_t)));
            CREATE_TRY(hipMemsetAsync(s->d_glast, 0, nw * sizeof(uint32_t), s->stream));
        }
    }
    if (s->host_eval) {
        CREATE_TRY(dev_alloc(s, &s->d_prop, (size_t)s->h * ldz * sizeof(double)));
        CREATE_TRY(dev_alloc(s, &s->d_p1, (size_t)s->h * sizeof(double)));
        CREATE_TRY(hipHostMalloc((void**)&s->h_prop, (size_t)s->h * (size_t)cfg->ndim * sizeof(double), hipHostMallocDefault));
        CREATE_TRY(hipHostMalloc((void**)&s->h_p1, (size_t)s->h * sizeof(double), hipHostMallocDefault));
        if (hipHostGetDevicePointer((void**)&s->h_prop_dev, s->h_prop, 0) != hipSuccess ||
            hipHostGetDevicePointer((void**)&s->h_p1_dev, s->h_p1, 0) != hipSuccess) {
            (void)hipGetLastError();
            s->h_prop_dev = s->h_p1_dev = nullptr;          // (then every batch goes through the copies)
        }
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
        // returns -- nthin = 100: +19 % on the loop with 64 MiB blocks, +6 % with 512 MiB; KMC_DEBUG=chain-block=n: samples per block, for tests)
        const int64_t unit = s->resident ? 1 : std::max<int64_t>(kGraphChunk, s->uchunk);     // (a resident launch is cut to the block: kmc_sampler_run)
        const int64_t per_unit = (unit + cfg->nthin - 1) / cfg->nthin + 1;
        const size_t sample_bytes = (size_t)s->nlocal * ldz * sizeof(double);
        int64_t blk = (int64_t)(((size_t)512 << 20) / sample_bytes);
        if (blk < 1) blk = 1;
        if (blk > 4096) blk = 4096;
        { const long v = debug_opt_long("chain-block", 0); if (v >= 1) blk = v; }
        if (blk < per_unit) blk = per_unit;
        s->stream_chain = true;
        s->stream_by_walker = (cfg->flags & KMC_CHAIN_BY_WALKER) != 0;
        if (s->stream_by_walker) {
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
                                              (s->nblob > 0 ? " MiB are free: thin it (nthin)" : " MiB are free: thin it (nthin), or stream it to host memory (KMC_STREAM_CHAIN)"));
            kmc_sampler_destroy(s);
            return r_;
        }
    }
    if ((cfg->flags & KMC_STORE_CHAIN) && s->nsamples > 0)
        CREATE_TRY(dev_alloc(s, &s->d_chain, (size_t)chain_slots * (size_t)s->nlocal * ldz * esz));
    if ((cfg->flags & KMC_STORE_LOGP) && s->nsamples > 0)
        CREATE_TRY(dev_alloc(s, &s->d_chain_logp, (size_t)chain_slots * (size_t)s->nlocal * sizeof(double)));
    if (s->nblob > 0) {
        CREATE_TRY(dev_alloc(s, &s->d_blob, nw * (size_t)s->nblob * sizeof(double)));
        CREATE_TRY(hipMemsetAsync(s->d_blob, 0, nw * (size_t)s->nblob * sizeof(double), s->stream));
        if ((cfg->flags & KMC_STORE_BLOBS) && s->nsamples > 0) {
            const size_t need_b = (size_t)s->nsamples * (size_t)s->nlocal * (size_t)s->nblob * sizeof(double);
            size_t free_b = 0, total_b = 0;
            CREATE_TRY(hipMemGetInfo(&free_b, &total_b));
            if (need_b > free_b) {
                kmc_status r_ = fail(KMC_ERR_OOM, "the stored blobs need " + std::to_string(need_b >> 20) + " MiB of device memory, " + std::to_string(free_b >> 20) + " MiB are free: thin the chain (nthin)");
                kmc_sampler_destroy(s);
                return r_;
            }
            CREATE_TRY(dev_alloc(s, &s->d_chain_blob, need_b));
        }
    }
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
    // The buffers go back to the allocation cache below, without the device-wide wait hipFree used to imply.  Work that OTHER
    // streams may still have in flight on them -- a caller's stream bound earlier (kmc_sampler_set_stream), a framework's
    // collectives on a replica shard's rows -- is waited for here; a plain single-GPU sampler (its own stream only) pays nothing.
    if (s->foreign_stream_seen || s->cfg.shard_count > 1 || s->comm) (void)hipDeviceSynchronize();
    if (!s->guards.empty() && s->stream) check_guards(s);
    drop_updated_graph(s);
    if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
    s->uk.keep.reset();                 // (the module goes when its last holder does: the density's cache, other samplers)
    if (s->graph) (void)hipGraphDestroy(s->graph);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->p2p) {
        for (int r = 0; r < 8; ++r) {
            if (r == s->cfg.shard_rank) continue;
            if (s->peer_pos[r]) (void)hipIpcCloseMemHandle(s->peer_pos[r]);
            if (s->peer_flags[r]) (void)hipIpcCloseMemHandle(s->peer_flags[r]);
        }
        cache_free(s->d_flags);
        cache_free(s->d_err);
    }
    if (s->comm) { rccl_comm_destroy(s->comm); s->comm = nullptr; }
    if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
    for (int i = 0; i < 3; ++i) {
        if (s->ev_filled[i]) (void)hipEventDestroy(s->ev_filled[i]);
        if (s->ev_copied[i]) (void)hipEventDestroy(s->ev_copied[i]);
    }
    chain_unregister(s);
    if (s->own_pos) cache_free(s->d_pos);
    cache_free(s->d_logp);          // the {logp, naccept, klast} block
    cache_free(s->d_mring);
    cache_free(s->d_ids);
    cache_free(s->d_mring_w);
    cache_free(s->d_mcnt);
    cache_free(s->d_gen);
    cache_free(s->d_sched);
    cache_free(s->d_chain);
    cache_free(s->bw_scratch);
    cache_free(s->bw_scratch_logp);
    cache_free(s->d_chain_logp);
    cache_free(s->d_blob);
    cache_free(s->d_chain_blob);
    cache_free(s->d_msum);
    cache_free(s->d_msumsq);
    cache_free(s->d_ring);
    cache_free(s->d_draws);
    cache_free(s->d_isum);
    cache_free(s->d_isumsq);
    cache_free(s->d_pos2);
    cache_free(s->d_logp2);
    cache_free(s->d_glast);
    cache_free(s->d_prop);
    cache_free(s->d_p1);
    if (s->h_prop) (void)hipHostFree(s->h_prop);
    if (s->h_p1) (void)hipHostFree(s->h_p1);
    cache_free(s->d_acc);
    if (s->h_acc) (void)hipHostFree(s->h_acc);
    for (int i = 0; i < kHostPieces; ++i) if (s->host_ev[i]) (void)hipEventDestroy(s->host_ev[i]);
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
    s->foreign_stream_seen = true;
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
    drop_updated_graph(s);                                              // (ensure_updated_graph builds it anew)
    if (s->own_pos) {
        for (size_t i = 0; i < s->guards.size(); ++i)                   // (KMC_DEBUG=poison: this allocation's guard goes with it)
            if (s->guards[i].first - s->guards[i].second == reinterpret_cast<char*>(s->d_pos)) { s->guards.erase(s->guards.begin() + (long)i); break; }
        cache_free(s->d_pos);
    }
    s->d_pos = static_cast<double*>(pos_dev);
    s->own_pos = false;
    s->pos2_current = false;
    s->positions_set = false;
    return KMC_OK;
}

KMC_EXPORT int64_t kmc_sampler_generation(const kmc_sampler* s) { return s ? s->generation : -1; }
KMC_EXPORT int64_t kmc_sampler_nsamples(const kmc_sampler* s) { return s ? s->nsamples : -1; }
KMC_EXPORT int64_t kmc_sampler_launch_count(const kmc_sampler* s) { return s ? s->launches : -1; }

namespace {
// how kmc_sampler_run issues the launches of the multi-launch kernels (two per generation, or one): kmc_launch.hip
std::string launch_mode_text(const kmc_sampler* s, const char* what)
{
    std::string t = ((s->cfg.flags & KMC_NO_GRAPH) || s->launch_mode == 2) ? std::string(", eager launches (") + what + " among the preloaded kernel parameters)"
                    : s->launch_mode == 3 ? ", hipGraph replay of " + std::to_string(s->uchunk) + " generations with per-replay parameter updates (" + what + " preloaded)"
                                          : std::string(", hipGraph replay of 64 generations");
    if (s->calib_graph_ms > 0.f) {
        char b[128];
        std::snprintf(b, sizeof(b), " (measured per 64 generations: table graph %.3f ms, %s %.3f ms)", s->calib_graph_ms, s->launch_mode == 2 ? "eager launches" : "updated graph", s->calib_eager_ms);
        t += b;
    }
    return t;
}
}  // namespace

// Human-readable description of how this sampler executes (kernel family, geometry, exchange).
KMC_EXPORT kmc_status kmc_sampler_describe(const kmc_sampler* s, char* buf, int64_t buflen)
{
    if (!s || !buf || buflen <= 0) return fail(KMC_ERR_BAD_ARG, "bad argument");
    std::ostringstream o;
    if (s->islands)
        o << "island mode: " << s->nislands << " islands of " << s->island_size << " walkers in LDS, " << s->island_gens
          << " generations per launch, rows 2 lanes x " << s->island_K << " chunks";
    else if (s->resident)
        o << "resident mode (exact): whole ensemble in one workgroup's LDS (" << s->resident_tpb
          << " threads), up to " << kDrawTableGens << " generations per launch, "
          << (s->resident_lane2 ? std::string("two walkers per thread") : s->resident_lane ? std::string("one walker per thread") : "rows 2 lanes x " + std::to_string(s->island_K) + " chunks")
          << ", the launch's draws from a wide kernel before it";
    else if (s->fused) {
        if (s->fused_L == 0)
            o << "one launch per generation (exact): generation_lane ND=" << s->cfg.ndim << ", one walker per lane, second-half walkers recompute their partner's first-half move, grid "
              << 2 * ((s->h + s->fused_tpb - 1) / s->fused_tpb) << " x " << s->fused_tpb;
        else
            o << "one launch per generation (exact): generation_group L=" << s->fused_L << " K=" << s->plan.K << ", rows lane-striped, second-half walkers recompute their partner's first-half move, grid "
              << 2 * ((s->h + s->fused_tpb / s->fused_L - 1) / (s->fused_tpb / s->fused_L)) << " x " << s->fused_tpb;
        o << launch_mode_text(s, "generation");
    } else if (s->host_eval)
        o << "host-evaluated density (exact): per half-step propose kernel -> D2H -> callback -> H2D -> accept kernel, grid "
          << s->grid << " x 256";
    else if (s->plan.vec) {
        o << "multi-launch (exact): half_step_vec L=" << s->plan.L << " K=" << s->plan.K << " ITER=" << s->plan.ITER
          << (s->plan.ragged ? " ragged" : " exact-size") << ", grid " << s->grid << " x " << s->tpb
          << launch_mode_text(s, "step");
    } else
        o << "multi-launch (exact): " << (s->user && s->uk.staged ? "half_step_staged (one walker per lane, rows staged through LDS)" : "half_step_generic (one walker per lane)")
          << ", grid " << s->grid << " x " << s->tpb;
    if (s->feed_replays > 0 && debug_opt("feed-stats")) {
        char b[200];
        std::snprintf(b, sizeof(b), "; feeding thread per replay of %lld generations: wait for a free executable %.1f us, parameter updates %.1f us, hipGraphLaunch %.1f us (%lld replays)",
                      (long long)s->uchunk, s->feed_wait_ns / 1e3 / s->feed_replays, s->feed_update_ns / 1e3 / s->feed_replays, s->feed_launch_ns / 1e3 / s->feed_replays, (long long)s->feed_replays);
        o << b;
    }
    if (s->budget_fallback) o << "; updated-graph budget of the process spent (kmc_set_updated_budget_mb): fell back to " << (s->launch_mode == 2 ? "eager launches" : "the table graph");
    if (s->push) o << "; accepted rows pushed into the peers' local copies (KMC_P2P_PUSH)";
    if (s->f32) o << "; rows kept in float (KMC_F32), arithmetic in double";
    if (s->stream_chain) o << "; chain streamed to host memory in blocks of " << s->ring_blk << " samples (device ring of 3 blocks" << (s->dst_chain_reg || s->dst_logp_reg ? ", destination page-locked" : "") << ")";
    if (s->d_ids) o << "; dealt sub-ensemble " << s->cfg.deal_rank << "/" << s->cfg.deal_count << " (walkers re-dealt between epochs)";
    if (s->d_mring) o << "; moments through a ring of " << s->mring_depth << " posted rows per wave";
    if (s->user) o << "; runtime-compiled density";
    if (s->user && s->user->is_body && s->fused)
        o << (s->fused_L == 0 ? " (function body, evaluated as written: one walker per lane)"
                              : " (function body recognised as a sum over elements and checked against the body on test rows: lane-striped)");
    else if (s->user && s->user->is_body && s->plan.vec && !s->resident && !s->islands)
    {
        const std::string& why_not = s->sep_off ? s->sep_off_note : s->user->sep_note;
        o << ((s->user->sep && !s->sep_off) ? " (function body recognised as a sum over elements and checked against the body on test rows: lane-striped)"
                                            : " (function body: rows lane-striped, the body evaluated per walker on the whole proposal" + (why_not.empty() ? std::string() : "; taken for a sum over elements, but " + why_not) + ")");
    }
    if (s->nblob > 0) o << " with a blob of " << s->nblob << " doubles per walker" << (s->d_chain_blob ? " (stored with every sample)" : "");
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
    if (which == 0 || which == 1) s->pos_exposed = true;         // (the caller may write there: a one-launch-per-generation sampler then re-synchronises its second copy before every run)
    switch (which) {
    case 0: return s->d_pos;
    case 1: return s->d_logp;
    case 2: return s->d_naccept;
    default: return nullptr;
    }
}


KMC_EXPORT int kmc_sizeof_config(void) { return (int)sizeof(kmc_config); }
KMC_EXPORT int kmc_sizeof_metropolis_config(void) { return (int)sizeof(kmc_metropolis_config); }
KMC_EXPORT int kmc_sizeof_outputs(void) { return (int)sizeof(kmc_outputs); }
KMC_EXPORT int kmc_sizeof_metropolis_outputs(void) { return (int)sizeof(kmc_metropolis_outputs); }

