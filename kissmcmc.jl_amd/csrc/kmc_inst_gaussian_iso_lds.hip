// Kernel instantiations for the isotropic Gaussian log-density, part 3 of 4: the LDS-resident kernels (islands, resident mode) and
// the many-chain Metropolis kernels.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
IslandFn island_gaussian_iso(int S, int K, bool ragged) { return island_lookup<GaussianIso>(S, K, ragged); }
ResidentFn resident_gaussian_iso(int tpb, int K, bool ragged) { return resident_lookup<GaussianIso>(tpb, K, ragged); }
ResidentFn resident_lane_gaussian_iso(int ndim, bool f32) { return resident_lane_lookup<GaussianIso>(ndim, f32); }
ResidentFn resident_lane2_gaussian_iso(int ndim) { return resident_lane2_lookup<GaussianIso>(ndim); }
GenerationFn generation_lane_gaussian_iso(int ndim) { return generation_lane_lookup<GaussianIso>(ndim); }
GenerationFn generation_group_gaussian_iso(int L, int K) { return generation_group_lookup<GaussianIso>(L, K); }
MetropolisFn metropolis_gaussian_iso(int ndim) { return metropolis_lookup<GaussianIso>(ndim); }
MetropolisTabledFn metropolis_tabled_gaussian_iso(int ndim) { return metropolis_tabled_lookup<GaussianIso>(ndim); }
}  // namespace kmc

#ifdef KMC_PROBE   // diagnostic build only (scripts/probe_generation.py): the stamps of THIS translation unit's kernels (generation_lane)
extern "C" __attribute__((visibility("default"))) int kmc_probe_read_generation(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc::g_probe), sizeof(kmc::g_probe));
}
#endif
