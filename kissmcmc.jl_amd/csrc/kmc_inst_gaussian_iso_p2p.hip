// Kernel instantiations for the isotropic Gaussian log-density, part 2 of 4: the peer-to-peer kernels (KMC_P2P).
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void part_p2p_gaussian_iso(int L, int K, int iter, bool ragged, HalfStepFn* vec, HalfStepFn* gen) { density_part<GaussianIso, 2>(L, K, iter, ragged, false, vec, gen); }
}  // namespace kmc
