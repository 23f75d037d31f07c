// Kernel instantiations for the 2-D correlated normal (test/runtests.jl:60) log-density, part 3 of 4: the LDS-resident kernels (islands, resident mode) and
// the many-chain Metropolis kernels.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
IslandFn island_mvnormal2(int S, int K, bool ragged) { return island_lookup<MvNormal2>(S, K, ragged); }
ResidentFn resident_mvnormal2(int tpb, int K, bool ragged) { return resident_lookup<MvNormal2>(tpb, K, ragged); }
ResidentFn resident_lane_mvnormal2(int ndim, bool f32) { return resident_lane_lookup<MvNormal2>(ndim, f32); }
ResidentFn resident_lane2_mvnormal2(int ndim) { return resident_lane2_lookup<MvNormal2>(ndim); }
GenerationFn generation_lane_mvnormal2(int ndim) { return generation_lane_lookup<MvNormal2>(ndim); }
GenerationFn generation_group_mvnormal2(int L, int K) { return generation_group_lookup<MvNormal2>(L, K); }
MetropolisFn metropolis_mvnormal2(int ndim) { return metropolis_lookup<MvNormal2>(ndim); }
MetropolisTabledFn metropolis_tabled_mvnormal2(int ndim) { return metropolis_tabled_lookup<MvNormal2>(ndim); }
}  // namespace kmc
