// Kernel instantiations for state dimension 3, polynomial order 1, flags 3 (bit 0 sine, bit 1 exp): the large
// libraries get a translation unit each so that the build spreads over the host cores.
#include "ops_table.hpp"
SYMODE_DEFINE_OPS_TU_FLAG(3, 1, 3)
