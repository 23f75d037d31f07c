// kmc_tables.hpp -- kernel instantiation tables.  Each density's table is compiled in its own
// translation unit (kmc_inst_<density>.hip) so the build can run them in parallel; the host driver only
// sees the per-density entry points declared at the bottom.
#pragma once
#include <type_traits>
#include "kmc_islands.hpp"
#include "kmc_generation.hpp"
#include "kmc_metropolis.hpp"

namespace kmc {

using HalfStepFn = void (*)(KMC_FRONT_TYPES, const HalfStepArgs);
using LogpdfFn = void (*)(const LogpdfArgs);
using FlushFn = void (*)(const FlushArgs);
using IslandFn = void (*)(const IslandArgs);
using ResidentFn = void (*)(const ResidentArgs);
using GenerationFn = void (*)(KMC_GEN_FRONT_TYPES, const GenerationArgs);
using InitBallFn = void (*)(const InitBallArgs);
using MetropolisFn = void (*)(const MetropolisArgs);
using MetropolisTabledFn = void (*)(const MetropolisArgs, const double*, int);

#ifdef KMC_TABLES_IMPL
template <class D, int L, int K, int ITER, bool P2P, bool RAGGED, class T>
HalfStepFn vec_one()
{
    // a group's ITER scalar lanes must fit in its L lanes; keep the register tile (ITER*K chunks) bounded
    if constexpr (ITER <= L && ITER * K <= 16) return half_step_vec<D, L, K, ITER, P2P, RAGGED, T>;
    else return nullptr;
}

template <class D, int L, int K, bool P2P, bool RAGGED, class T>
HalfStepFn vec_iter(int iter)
{
    // full ITER range only for the single-GPU exact double kernels (tuning); the others: what make_plan picks
    constexpr bool kWide = !P2P && !RAGGED && sizeof(T) == 8;
    switch (iter) {
    case 1: return vec_one<D, L, K, 1, P2P, RAGGED, T>();
    case 2: return vec_one<D, L, K, 2, P2P, RAGGED, T>();
    case 4: return vec_one<D, L, K, 4, P2P, RAGGED, T>();
    case 8: if constexpr (kWide || (P2P && !RAGGED)) return vec_one<D, L, K, 8, P2P, RAGGED, T>(); else return nullptr;
    case 16: if constexpr (kWide) return vec_one<D, L, K, 16, P2P, RAGGED, T>(); else return nullptr;
    default: return nullptr;
    }
}

// A density's kernels are instantiated in FOUR translation units, so that the build's longest job is a quarter of a density
// (kmc_inst_<density>{,_var,_p2p,_lds}.hip): PART 0 = double rows, exact size, one GPU (incl. the tuning geometries);
// PART 1 = ragged sizes and KMC_F32 rows, one GPU; PART 2 = the peer-to-peer kernels; (the LDS-resident and Metropolis
// kernels are the fourth).  Each part only names -- and therefore only compiles -- its own instantiations.
template <class D, int L, int K, int PART>
HalfStepFn vec_pick(int iter, bool ragged, bool f32)
{
    if constexpr (PART == 0) return vec_iter<D, L, K, false, false, double>(iter);
    else if constexpr (PART == 1) {
        if (f32) return ragged ? vec_iter<D, L, K, false, true, float>(iter) : vec_iter<D, L, K, false, false, float>(iter);   // KMC_F32: single rows, one GPU
        return vec_iter<D, L, K, false, true, double>(iter);
    } else return ragged ? vec_iter<D, L, K, true, true, double>(iter) : vec_iter<D, L, K, true, false, double>(iter);
}

// geometries make_plan can pick: exact + ragged, single-GPU + P2P; the extra exact single-GPU ones
// exist for tuning (KMC_PLAN)
template <class D, int PART>
HalfStepFn vec_lookup(int L, int K, int iter, bool ragged, bool f32)
{
    if constexpr (!D::kHasFrag) {
        return nullptr;
    } else {
#define KMC_LK(l, k) if (L == l && K == k) return vec_pick<D, l, k, PART>(iter, ragged, f32);
        KMC_LK(1, 1) KMC_LK(2, 1) KMC_LK(4, 1) KMC_LK(4, 2) KMC_LK(8, 2) KMC_LK(16, 2) KMC_LK(32, 2) KMC_LK(64, 2)
        KMC_LK(64, 4) KMC_LK(64, 8)
#undef KMC_LK
        if constexpr (PART == 0 && std::is_same<D, GaussianIso>::value) {       // (tuning geometries, KMC_PLAN: the bench density only -- build time)
#define KMC_LK(l, k) if (L == l && K == k) return vec_iter<D, l, k, false, false, double>(iter);
            KMC_LK(8, 1) KMC_LK(16, 1) KMC_LK(32, 1) KMC_LK(64, 1) KMC_LK(4, 4) KMC_LK(8, 4)
#undef KMC_LK
        }
        return nullptr;
    }
}

// (the part's vector kernel for this geometry or nullptr, its generic kernel)
template <class D, int PART>
void density_part(int L, int K, int iter, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen)
{
    *vec = vec_lookup<D, PART>(L, K, iter, ragged, f32);
    if constexpr (PART == 0) *gen = half_step_generic<D, false, double>;
    else if constexpr (PART == 1) *gen = f32 ? half_step_generic<D, false, float> : half_step_generic<D, false, double>;
    else *gen = half_step_generic<D, true, double>;
}

// island mode: one workgroup per S-walker island, rows of up to 4*K doubles
template <class D, int S>
IslandFn island_lookup_s(int K, bool ragged)
{
    switch (K) {
    case 1: return ragged ? island_epoch<D, S, 1, true> : island_epoch<D, S, 1, false>;
    case 2: return ragged ? island_epoch<D, S, 2, true> : island_epoch<D, S, 2, false>;
    case 4: return ragged ? island_epoch<D, S, 4, true> : island_epoch<D, S, 4, false>;
    case 8: return ragged ? island_epoch<D, S, 8, true> : island_epoch<D, S, 8, false>;
    default: return nullptr;
    }
}

template <class D>
IslandFn island_lookup(int S, int K, bool ragged)
{
    if constexpr (!D::kHasFrag) {
        return nullptr;
    } else {
        switch (S) {
        case 64: return island_lookup_s<D, 64>(K, ragged);
        case 128: return island_lookup_s<D, 128>(K, ragged);
        case 256: return island_lookup_s<D, 256>(K, ragged);
        default: return nullptr;
        }
    }
}
// resident mode: the exact sampler for ensembles that fit one workgroup's LDS
template <class D, int TPB>
ResidentFn resident_lookup_t(int K, bool ragged)
{
    switch (K) {
    case 1: return ragged ? resident_epoch<D, 1, true, TPB> : resident_epoch<D, 1, false, TPB>;
    case 2: return ragged ? resident_epoch<D, 2, true, TPB> : resident_epoch<D, 2, false, TPB>;
    case 4: return ragged ? resident_epoch<D, 4, true, TPB> : resident_epoch<D, 4, false, TPB>;
    case 8: return ragged ? resident_epoch<D, 8, true, TPB> : resident_epoch<D, 8, false, TPB>;
    default: return nullptr;
    }
}

template <class D>
ResidentFn resident_lookup(int tpb, int K, bool ragged)
{
    if constexpr (!D::kHasFrag) {
        return nullptr;
    } else {
        switch (tpb) {
        case 256: return resident_lookup_t<D, 256>(K, ragged);
        case 512: return resident_lookup_t<D, 512>(K, ragged);
        case 1024: return resident_lookup_t<D, 1024>(K, ragged);
        default: return nullptr;
        }
    }
}
// resident mode, one walker per thread (short rows: ndim <= ND <= 8); double or float rows
template <class D, class T>
ResidentFn resident_lane_lookup_t(int ndim)
{
    switch (ndim) {           // exact row lengths: no per-element guards in the kernel
    case 1: return resident_lane<D, 1, T>;
    case 2: return resident_lane<D, 2, T>;
    case 3: return resident_lane<D, 3, T>;
    case 4: return resident_lane<D, 4, T>;
    case 5: return resident_lane<D, 5, T>;
    case 6: return resident_lane<D, 6, T>;
    case 7: return resident_lane<D, 7, T>;
    case 8: return resident_lane<D, 8, T>;
    default: return nullptr;
    }
}
template <class D>
ResidentFn resident_lane_lookup(int ndim, bool f32)
{
    return f32 ? resident_lane_lookup_t<D, float>(ndim) : resident_lane_lookup_t<D, double>(ndim);
}
// ... two walkers per thread (1026 .. 2048 walkers, double rows)
template <class D>
ResidentFn resident_lane2_lookup(int ndim)
{
    switch (ndim) {
    case 1: return resident_lane2<D, 1>;
    case 2: return resident_lane2<D, 2>;
    case 3: return resident_lane2<D, 3>;
    case 4: return resident_lane2<D, 4>;
    case 5: return resident_lane2<D, 5>;
    case 6: return resident_lane2<D, 6>;
    case 7: return resident_lane2<D, 7>;
    case 8: return resident_lane2<D, 8>;
    default: return nullptr;
    }
}
// one launch per generation, one walker per lane (mid-size ensembles, short double rows: kmc_generation.hpp)
template <class D>
GenerationFn generation_lane_lookup(int ndim)
{
    switch (ndim) {
    case 1: return generation_lane<D, 1>;
    case 2: return generation_lane<D, 2>;
    case 3: return generation_lane<D, 3>;
    case 4: return generation_lane<D, 4>;
    case 5: return generation_lane<D, 5>;
    case 6: return generation_lane<D, 6>;
    case 7: return generation_lane<D, 7>;
    case 8: return generation_lane<D, 8>;
    default: return nullptr;
    }
}
// ... and lane-striped (longer rows of small ensembles): the geometries make_plan picks for the vector kernels
template <class D>
GenerationFn generation_group_lookup(int L, int K)
{
    if constexpr (!D::kHasFrag) {
        return nullptr;
    } else {
#define KMC_LK(l, k) if (L == l && K == k) return generation_group<D, l, k>;
        KMC_LK(1, 1) KMC_LK(2, 1) KMC_LK(4, 1) KMC_LK(4, 2) KMC_LK(8, 2) KMC_LK(16, 2) KMC_LK(32, 2) KMC_LK(64, 2)
        KMC_LK(64, 4) KMC_LK(64, 8)
#undef KMC_LK
        return nullptr;
    }
}
// many-chain Metropolis: the chain in registers up to 32 dimensions, in memory beyond
template <class D>
MetropolisFn metropolis_lookup(int ndim)
{
    if (ndim <= 1) return metropolis_chains<D, 1>;
    if (ndim <= 2) return metropolis_chains<D, 2>;
    if (ndim <= 4) return metropolis_chains<D, 4>;
    if (ndim <= 8) return metropolis_chains<D, 8>;
    if (ndim <= 16) return metropolis_chains<D, 16>;
    if (ndim <= 32) return metropolis_chains<D, 32>;
    return metropolis_chains<D, 0>;
}
// few chains: the draws from a table (kmc_metropolis.hpp: metropolis_chains_tabled), chains in registers up to 8 dimensions (menu densities -- build time; runtime-compiled ones up to 32)
template <class D>
MetropolisTabledFn metropolis_tabled_lookup(int ndim)
{
    if (ndim <= 1) return metropolis_chains_tabled<D, 1>;
    if (ndim <= 2) return metropolis_chains_tabled<D, 2>;
    if (ndim <= 4) return metropolis_chains_tabled<D, 4>;
    if (ndim <= 8) return metropolis_chains_tabled<D, 8>;
    return nullptr;
}
#endif  // KMC_TABLES_IMPL

// entry points per density: table_<density> (kmc_inst_<density>.hip: PART 0, the log-pdf and initial-ball kernels, and the
// dispatch to the other parts), part_var_ / part_p2p_<density> (kmc_inst_<density>_var.hip / _p2p.hip), and the LDS-resident +
// Metropolis tables below (kmc_inst_<density>_lds.hip)
#define KMC_DECLARE_DENSITY_TABLE(name) \
    void table_##name(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp); \
    void part_var_##name(int L, int K, int iter, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen);                      \
    void part_p2p_##name(int L, int K, int iter, bool ragged, HalfStepFn* vec, HalfStepFn* gen)
KMC_DECLARE_DENSITY_TABLE(gaussian_iso);
KMC_DECLARE_DENSITY_TABLE(exponential);
KMC_DECLARE_DENSITY_TABLE(rosenbrock);
KMC_DECLARE_DENSITY_TABLE(lognormal);
KMC_DECLARE_DENSITY_TABLE(mvnormal2);
HalfStepFn half_step_host();
IslandFn island_gaussian_iso(int S, int K, bool ragged);
ResidentFn resident_gaussian_iso(int tpb, int K, bool ragged);
ResidentFn resident_lane_gaussian_iso(int ndim, bool f32);
ResidentFn resident_lane2_gaussian_iso(int ndim);
GenerationFn generation_lane_gaussian_iso(int ndim);
GenerationFn generation_group_gaussian_iso(int L, int K);
InitBallFn init_ball_gaussian_iso();
MetropolisFn metropolis_gaussian_iso(int ndim);
MetropolisTabledFn metropolis_tabled_gaussian_iso(int ndim);
IslandFn island_exponential(int S, int K, bool ragged);
ResidentFn resident_exponential(int tpb, int K, bool ragged);
ResidentFn resident_lane_exponential(int ndim, bool f32);
ResidentFn resident_lane2_exponential(int ndim);
GenerationFn generation_lane_exponential(int ndim);
GenerationFn generation_group_exponential(int L, int K);
InitBallFn init_ball_exponential();
MetropolisFn metropolis_exponential(int ndim);
MetropolisTabledFn metropolis_tabled_exponential(int ndim);
IslandFn island_rosenbrock(int S, int K, bool ragged);
ResidentFn resident_rosenbrock(int tpb, int K, bool ragged);
ResidentFn resident_lane_rosenbrock(int ndim, bool f32);
ResidentFn resident_lane2_rosenbrock(int ndim);
GenerationFn generation_lane_rosenbrock(int ndim);
GenerationFn generation_group_rosenbrock(int L, int K);
InitBallFn init_ball_rosenbrock();
MetropolisFn metropolis_rosenbrock(int ndim);
MetropolisTabledFn metropolis_tabled_rosenbrock(int ndim);
IslandFn island_lognormal(int S, int K, bool ragged);
ResidentFn resident_lognormal(int tpb, int K, bool ragged);
ResidentFn resident_lane_lognormal(int ndim, bool f32);
ResidentFn resident_lane2_lognormal(int ndim);
GenerationFn generation_lane_lognormal(int ndim);
GenerationFn generation_group_lognormal(int L, int K);
InitBallFn init_ball_lognormal();
MetropolisFn metropolis_lognormal(int ndim);
MetropolisTabledFn metropolis_tabled_lognormal(int ndim);
IslandFn island_mvnormal2(int S, int K, bool ragged);
ResidentFn resident_mvnormal2(int tpb, int K, bool ragged);
ResidentFn resident_lane_mvnormal2(int ndim, bool f32);
ResidentFn resident_lane2_mvnormal2(int ndim);
GenerationFn generation_lane_mvnormal2(int ndim);
GenerationFn generation_group_mvnormal2(int L, int K);
InitBallFn init_ball_mvnormal2();
MetropolisFn metropolis_mvnormal2(int ndim);
MetropolisTabledFn metropolis_tabled_mvnormal2(int ndim);

}  // namespace kmc
