// Kernel instantiations for the chained Rosenbrock (test/runtests.jl:68 at N = 2) log-density, part 3 of 4: the LDS-resident kernels (islands, resident mode) and
// the many-chain Metropolis kernels.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
IslandFn island_rosenbrock(int S, int K, bool ragged) { return island_lookup<Rosenbrock>(S, K, ragged); }
ResidentFn resident_rosenbrock(int tpb, int K, bool ragged) { return resident_lookup<Rosenbrock>(tpb, K, ragged); }
ResidentFn resident_lane_rosenbrock(int ndim, bool f32) { return resident_lane_lookup<Rosenbrock>(ndim, f32); }
ResidentFn resident_lane2_rosenbrock(int ndim) { return resident_lane2_lookup<Rosenbrock>(ndim); }
GenerationFn generation_lane_rosenbrock(int ndim) { return generation_lane_lookup<Rosenbrock>(ndim); }
GenerationFn generation_group_rosenbrock(int L, int K) { return generation_group_lookup<Rosenbrock>(L, K); }
MetropolisFn metropolis_rosenbrock(int ndim) { return metropolis_lookup<Rosenbrock>(ndim); }
MetropolisTabledFn metropolis_tabled_rosenbrock(int ndim) { return metropolis_tabled_lookup<Rosenbrock>(ndim); }
}  // namespace kmc

#ifdef KMC_PROBE   // diagnostic build only (scripts/probe_timeline.py C3): the stamps of THIS translation unit's kernels (generation_group)
extern "C" __attribute__((visibility("default"))) int kmc_probe_read_generation_rosenbrock(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc::g_probe), sizeof(kmc::g_probe));
}
#endif
