// Kernel instantiations for the exponential (README.md:15) log-density, part 3 of 4: the LDS-resident kernels (islands, resident mode) and
// the many-chain Metropolis kernels.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
IslandFn island_exponential(int S, int K, bool ragged) { return island_lookup<Exponential>(S, K, ragged); }
ResidentFn resident_exponential(int tpb, int K, bool ragged) { return resident_lookup<Exponential>(tpb, K, ragged); }
ResidentFn resident_lane_exponential(int ndim, bool f32) { return resident_lane_lookup<Exponential>(ndim, f32); }
ResidentFn resident_lane2_exponential(int ndim) { return resident_lane2_lookup<Exponential>(ndim); }
GenerationFn generation_lane_exponential(int ndim) { return generation_lane_lookup<Exponential>(ndim); }
GenerationFn generation_group_exponential(int L, int K) { return generation_group_lookup<Exponential>(L, K); }
MetropolisFn metropolis_exponential(int ndim) { return metropolis_lookup<Exponential>(ndim); }
MetropolisTabledFn metropolis_tabled_exponential(int ndim) { return metropolis_tabled_lookup<Exponential>(ndim); }
}  // namespace kmc
