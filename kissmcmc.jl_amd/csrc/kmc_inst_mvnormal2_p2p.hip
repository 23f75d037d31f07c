// Kernel instantiations for the 2-D correlated normal (test/runtests.jl:60) log-density, part 2 of 4: the peer-to-peer kernels (KMC_P2P).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void part_p2p_mvnormal2(int L, int K, int iter, bool ragged, HalfStepFn* vec, HalfStepFn* gen) { density_part<MvNormal2, 2>(L, K, iter, ragged, false, vec, gen); }
}  // namespace kmc
