// Kernel instantiations for state dimension D = 2 (orders 1-5, sine/exp on/off).
#include "ops_table.hpp"
namespace symode {
static const LibOps kTab[] = {SYMODE_OPS_ALL_FLAGS(2, 1), SYMODE_OPS_ALL_FLAGS(2, 2), SYMODE_OPS_ALL_FLAGS(2, 3),
                              SYMODE_OPS_ALL_FLAGS(2, 4), SYMODE_OPS_ALL_FLAGS(2, 5)};
const LibOps* ops_d2(int order, int flags) { return find_in(kTab, sizeof(kTab) / sizeof(kTab[0]), order, flags); }
}  // namespace symode
