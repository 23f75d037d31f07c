// Kernel instantiations for the exponential (README.md:15) log-density, part 2 of 4: the peer-to-peer kernels (KMC_P2P).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void part_p2p_exponential(int L, int K, int iter, bool ragged, HalfStepFn* vec, HalfStepFn* gen) { density_part<Exponential, 2>(L, K, iter, ragged, false, vec, gen); }
}  // namespace kmc
