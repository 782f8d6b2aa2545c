// Kernel instantiations for state dimension 2, polynomial order 3 (sine / exp on / off): one translation unit per
// (dimension, order) so that the build spreads over the host cores.
#include "ops_table.hpp"
SYMODE_DEFINE_OPS_TU(2, 3)
