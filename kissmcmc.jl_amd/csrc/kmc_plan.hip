// kmc_plan.hip -- which kernel, in which geometry, for a configuration: the per-density kernel tables (instantiated in
// kmc_inst_<density>*.hip), the launch plan per ndim (lanes per row, chunks per lane, walkers per wave), the parameter digest the
// kernels read, and the island / dealt permutations.  The sampler's lifecycle that uses them is kmc_sampler.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "kmc_sampler.hpp"

using namespace kmc;
using namespace kmc_host;

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

namespace kmc_host {
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

ResidentFn resident_lane_fn(int density, int ndim, bool f32)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return resident_lane_gaussian_iso(ndim, f32);
    case KMC_EXPONENTIAL: return resident_lane_exponential(ndim, f32);
    case KMC_ROSENBROCK: return resident_lane_rosenbrock(ndim, f32);
    case KMC_LOGNORMAL: return resident_lane_lognormal(ndim, f32);
    case KMC_MVNORMAL2: return resident_lane_mvnormal2(ndim, f32);
    default: return nullptr;
    }
}

ResidentFn resident_lane2_fn(int density, int ndim)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return resident_lane2_gaussian_iso(ndim);
    case KMC_EXPONENTIAL: return resident_lane2_exponential(ndim);
    case KMC_ROSENBROCK: return resident_lane2_rosenbrock(ndim);
    case KMC_LOGNORMAL: return resident_lane2_lognormal(ndim);
    case KMC_MVNORMAL2: return resident_lane2_mvnormal2(ndim);
    default: return nullptr;
    }
}

GenerationFn generation_fn(int density, int ndim)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return generation_lane_gaussian_iso(ndim);
    case KMC_EXPONENTIAL: return generation_lane_exponential(ndim);
    case KMC_ROSENBROCK: return generation_lane_rosenbrock(ndim);
    case KMC_LOGNORMAL: return generation_lane_lognormal(ndim);
    case KMC_MVNORMAL2: return generation_lane_mvnormal2(ndim);
    default: return nullptr;
    }
}

GenerationFn generation_group_fn(int density, int L, int K)
{
    switch (density) {
    case KMC_GAUSSIAN_ISO: return generation_group_gaussian_iso(L, K);
    case KMC_EXPONENTIAL: return generation_group_exponential(L, K);
    case KMC_ROSENBROCK: return generation_group_rosenbrock(L, K);
    case KMC_LOGNORMAL: return generation_group_lognormal(L, K);
    case KMC_MVNORMAL2: return generation_group_mvnormal2(L, K);
    default: return nullptr;
    }
}

// resident mode with one walker per thread (short rows) or two lanes per walker: KMC_DEBUG=resident=pair decides for tests
bool resident_lane_wanted(int64_t ndim)
{
    std::string e;
    if (debug_opt("resident", &e) && e == "pair") return false;
    return ndim <= 8;
}
int lane_nd(int64_t ndim) { return (int)ndim; }          // (the lane kernels are instantiated for the exact row length)

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
        if (c.density == KMC_USER_DENSITY && chunks > 64 && chunks <= 128) {
            // a function body evaluated once per walker (RowEvalTrait) on rows of 129 ... 256 doubles: one lane walks the whole row, so the walkers per wave
            // count -- 16 lanes x 8 chunks (4 walkers per wave) instead of 64 x 2 (one): 16 384 x 256 30.1 -> 22.7 us per half-step, 4 096 x 256 22.1 -> 16.0, 2 048 x 256
            // 16.9 -> 13.7, 8 192 x 200 26.2 -> 24.7 (profiles/NOTES.md round 5; shorter and longer rows measured no better that way)
            const kmc_user_density* ud = static_cast<const kmc_user_density*>(c.user_density);
            if (ud && ud->is_body && !sep_routed(ud) && body_vec_possible(ud, c.ndim)) { L = 16; K = 8; }
        }
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
        const kmc_user_density* ud = static_cast<const kmc_user_density*>(c.user_density);
        // a function body in the vector kernel: lane-striped like term / pair when it was recognised as a sum over elements, else
        // with its rows lane-striped and only the evaluation per walker (RowEvalTrait); KMC_DEBUG=no-body-routing / no-body-vec
        // keep it one walker per lane (ensembles small enough for the resident kernels: decided by the caller)
        const bool body = ud && ud->is_body && !(sep_routed(ud) || body_vec_possible(ud, c.ndim));
        if (!body && !force_generic && L > 0 && 2 * L * K >= c.ndim && iter <= L && iter * K <= 16) {
            // (a body evaluated per walker keeps a tile of the wave's proposals in LDS: at most 64 KiB per workgroup)
            if (ud && ud->is_body && !sep_routed(ud)) while (iter > 1 && body_vec_lds_bytes(L, K, iter) > 65536) iter /= 2;
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

}  // namespace kmc_host
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
