// Kernel instantiations for the exponential (README.md:15) log-density, part 1 of 4: ragged row sizes and KMC_F32 rows, one GPU.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
void part_var_exponential(int L, int K, int iter, bool ragged, bool f32, HalfStepFn* vec, HalfStepFn* gen) { density_part<Exponential, 1>(L, K, iter, ragged, f32, vec, gen); }
}  // namespace kmc
