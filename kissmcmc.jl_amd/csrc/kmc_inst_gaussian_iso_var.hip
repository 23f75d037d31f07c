// Kernel instantiations for the isotropic Gaussian log-density, part 1 of 4: ragged row sizes and KMC_F32 rows, one GPU.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void part_var_gaussian_iso(int L, int K, int iter, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen) { density_part<GaussianIso, 1>(L, K, iter, ragged, f32, vec, gen); }
}  // namespace kmc

#ifdef KMC_PROBE   // diagnostic build only (scripts/probe_timeline.py R31 ...): the stamps of THIS translation unit's kernels (ragged rows)
extern "C" __attribute__((visibility("default"))) int kmc_probe_read_var(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc::g_probe), sizeof(kmc::g_probe));
}
#endif
