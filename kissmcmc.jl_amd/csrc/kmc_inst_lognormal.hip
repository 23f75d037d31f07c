// Kernel instantiations for the LogNormal log-density (one translation unit per density).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_lognormal(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    density_fns<LogNormal>(L, K, iter, p2p, ragged, f32, vec, gen, lp);
}
IslandFn island_lognormal(int S, int K, bool ragged) { return island_lookup<LogNormal>(S, K, ragged); }
ResidentFn resident_lognormal(int tpb, int K, bool ragged) { return resident_lookup<LogNormal>(tpb, K, ragged); }
InitBallFn init_ball_lognormal() { return init_ball<LogNormal>; }
MetropolisFn metropolis_lognormal(int ndim) { return metropolis_lookup<LogNormal>(ndim); }
}  // namespace kmc
