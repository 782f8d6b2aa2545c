// Kernel instantiations for state dimension D = 3 (orders 1-4, sine/exp on/off).
#include "ops_table.hpp"
namespace symode {
static const LibOps kTab[] = {SYMODE_OPS_ALL_FLAGS(3, 1), SYMODE_OPS_ALL_FLAGS(3, 2), SYMODE_OPS_ALL_FLAGS(3, 3),
                              SYMODE_OPS_ALL_FLAGS(3, 4)};
const LibOps* ops_d3(int order, int flags) { return find_in(kTab, sizeof(kTab) / sizeof(kTab[0]), order, flags); }
}  // namespace symode
