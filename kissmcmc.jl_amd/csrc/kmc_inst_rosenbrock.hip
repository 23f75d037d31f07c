// Kernel instantiations for the Rosenbrock log-density (one translation unit per density).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_rosenbrock(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    density_fns<Rosenbrock>(L, K, iter, p2p, ragged, f32, vec, gen, lp);
}
IslandFn island_rosenbrock(int S, int K, bool ragged) { return island_lookup<Rosenbrock>(S, K, ragged); }
ResidentFn resident_rosenbrock(int tpb, int K, bool ragged) { return resident_lookup<Rosenbrock>(tpb, K, ragged); }
InitBallFn init_ball_rosenbrock() { return init_ball<Rosenbrock>; }
MetropolisFn metropolis_rosenbrock(int ndim) { return metropolis_lookup<Rosenbrock>(ndim); }
}  // namespace kmc

#ifdef KMC_PROBE   // diagnostic build only (scripts/probe_timeline.py C3): this translation unit's copy of the stamps
extern "C" __attribute__((visibility("default"))) int kmc_probe_read_rosenbrock(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc::g_probe), sizeof(kmc::g_probe));
}
#endif
