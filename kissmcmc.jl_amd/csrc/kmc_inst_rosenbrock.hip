// Kernel instantiations for the chained Rosenbrock (test/runtests.jl:68 at N = 2) log-density, part 0 of 4 (kmc_tables.hpp: vec_pick): double rows of exact size on
// one GPU, the generic kernel, the log-pdf and initial-ball kernels -- and the dispatch to the other parts.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_rosenbrock(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    *lp = logpdf_rows<Rosenbrock>;
    if (p2p) { if (f32) { *vec = nullptr; *gen = nullptr; } else part_p2p_rosenbrock(L, K, iter, ragged, vec, gen); }
    else if (ragged || f32) part_var_rosenbrock(L, K, iter, ragged, f32, vec, gen);
    else density_part<Rosenbrock, 0>(L, K, iter, false, false, vec, gen);
}
InitBallFn init_ball_rosenbrock() { return init_ball<Rosenbrock>; }
}  // namespace kmc

#ifdef KMC_PROBE   // diagnostic build only (scripts/probe_timeline.py C3): this translation unit's copy of the stamps
extern "C" __attribute__((visibility("default"))) int kmc_probe_read_rosenbrock(void* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc::g_probe), sizeof(kmc::g_probe));
}
#endif
