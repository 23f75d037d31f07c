// Kernel instantiations for the Exponential log-density (one translation unit per density).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_exponential(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    density_fns<Exponential>(L, K, iter, p2p, ragged, f32, vec, gen, lp);
}
IslandFn island_exponential(int S, int K, bool ragged) { return island_lookup<Exponential>(S, K, ragged); }
ResidentFn resident_exponential(int tpb, int K, bool ragged) { return resident_lookup<Exponential>(tpb, K, ragged); }
InitBallFn init_ball_exponential() { return init_ball<Exponential>; }
MetropolisFn metropolis_exponential(int ndim) { return metropolis_lookup<Exponential>(ndim); }
}  // namespace kmc
