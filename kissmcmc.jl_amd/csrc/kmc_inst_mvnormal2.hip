// Kernel instantiations for the MvNormal2 log-density (one translation unit per density).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_mvnormal2(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    density_fns<MvNormal2>(L, K, iter, p2p, ragged, f32, vec, gen, lp);
}
IslandFn island_mvnormal2(int S, int K, bool ragged) { return island_lookup<MvNormal2>(S, K, ragged); }
ResidentFn resident_mvnormal2(int tpb, int K, bool ragged) { return resident_lookup<MvNormal2>(tpb, K, ragged); }
InitBallFn init_ball_mvnormal2() { return init_ball<MvNormal2>; }
MetropolisFn metropolis_mvnormal2(int ndim) { return metropolis_lookup<MvNormal2>(ndim); }
}  // namespace kmc
