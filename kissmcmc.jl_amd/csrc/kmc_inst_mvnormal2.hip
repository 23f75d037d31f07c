// Kernel instantiations for the 2-D correlated normal (test/runtests.jl:60) log-density, part 0 of 4 (kmc_tables.hpp: vec_pick): double rows of exact size on
// one GPU, the generic kernel, the log-pdf and initial-ball kernels -- and the dispatch to the other parts.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void table_mvnormal2(int L, int K, int iter, bool p2p, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen, LogpdfFn* lp)
{
    *lp = logpdf_rows<MvNormal2>;
    if (p2p) { if (f32) { *vec = nullptr; *gen = nullptr; } else part_p2p_mvnormal2(L, K, iter, ragged, vec, gen); }
    else if (ragged || f32) part_var_mvnormal2(L, K, iter, ragged, f32, vec, gen);
    else density_part<MvNormal2, 0>(L, K, iter, false, false, vec, gen);
}
InitBallFn init_ball_mvnormal2() { return init_ball<MvNormal2>; }
}  // namespace kmc
