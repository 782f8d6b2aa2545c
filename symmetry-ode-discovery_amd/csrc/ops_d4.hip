// Kernel instantiations for state dimension D = 4 (orders 1-3, sine/exp on/off).
#include "ops_table.hpp"
namespace symode {
static const LibOps kTab[] = {SYMODE_OPS_ALL_FLAGS(4, 1), SYMODE_OPS_ALL_FLAGS(4, 2), SYMODE_OPS_ALL_FLAGS(4, 3)};
const LibOps* ops_d4(int order, int flags) { return find_in(kTab, sizeof(kTab) / sizeof(kTab[0]), order, flags); }
}  // namespace symode
