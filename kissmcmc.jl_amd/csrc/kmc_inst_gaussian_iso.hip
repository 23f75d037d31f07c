// Kernel instantiations for the GaussianIso log-density (one translation unit per density).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_gaussian_iso(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    density_fns<GaussianIso>(L, K, iter, p2p, ragged, f32, vec, gen, lp);
}
IslandFn island_gaussian_iso(int S, int K, bool ragged) { return island_lookup<GaussianIso>(S, K, ragged); }
ResidentFn resident_gaussian_iso(int tpb, int K, bool ragged) { return resident_lookup<GaussianIso>(tpb, K, ragged); }
InitBallFn init_ball_gaussian_iso() { return init_ball<GaussianIso>; }
MetropolisFn metropolis_gaussian_iso(int ndim) { return metropolis_lookup<GaussianIso>(ndim); }
}  // namespace kmc

#ifdef KMC_PROBE   // diagnostic build only (scripts/probe_timeline.py)
extern "C" __attribute__((visibility("default"))) int kmc_probe_read(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc::g_probe), sizeof(kmc::g_probe));
}
#endif
