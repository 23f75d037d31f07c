// Kernel instantiations for the LogNormal log-density, part 3 of 4: the LDS-resident kernels (islands, resident mode) and
// the many-chain Metropolis kernels.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
IslandFn island_lognormal(int S, int K, bool ragged) { return island_lookup<LogNormal>(S, K, ragged); }
ResidentFn resident_lognormal(int tpb, int K, bool ragged) { return resident_lookup<LogNormal>(tpb, K, ragged); }
ResidentFn resident_lane_lognormal(int ndim, bool f32) { return resident_lane_lookup<LogNormal>(ndim, f32); }
ResidentFn resident_lane2_lognormal(int ndim) { return resident_lane2_lookup<LogNormal>(ndim); }
GenerationFn generation_lane_lognormal(int ndim) { return generation_lane_lookup<LogNormal>(ndim); }
GenerationFn generation_group_lognormal(int L, int K) { return generation_group_lookup<LogNormal>(L, K); }
MetropolisFn metropolis_lognormal(int ndim) { return metropolis_lookup<LogNormal>(ndim); }
MetropolisTabledFn metropolis_tabled_lognormal(int ndim) { return metropolis_tabled_lookup<LogNormal>(ndim); }
}  // namespace kmc
