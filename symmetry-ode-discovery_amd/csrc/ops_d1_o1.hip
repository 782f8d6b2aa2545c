// Kernel instantiations for state dimension 1, polynomial order 1 (sine / exp on / off): one translation unit per
// (dimension, order) so that the build spreads over the host cores.
#include "ops_table.hpp"
SYMODE_DEFINE_OPS_TU(1, 1)
