// Kernel instantiations for host-evaluated log-densities (KMC_HOST_DENSITY): the generic half-step
// kernel only -- the path is bound by the host callback and the PCIe round trip, not by the kernel.
#define KMC_TABLES_IMPL
#include "kmc_tables.hpp"

namespace kmc {
HalfStepFn half_step_host() { return half_step_generic<HostEval, false>; }
}  // namespace kmc
