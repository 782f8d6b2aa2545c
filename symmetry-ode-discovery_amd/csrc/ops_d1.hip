// Kernel instantiations for state dimension D = 1 (orders 1-5, sine/exp on/off).
#include "ops_table.hpp"
namespace symode {
static const LibOps kTab[] = {SYMODE_OPS_ALL_FLAGS(1, 1), SYMODE_OPS_ALL_FLAGS(1, 2), SYMODE_OPS_ALL_FLAGS(1, 3),
                              SYMODE_OPS_ALL_FLAGS(1, 4), SYMODE_OPS_ALL_FLAGS(1, 5)};
const LibOps* ops_d1(int order, int flags) { return find_in(kTab, sizeof(kTab) / sizeof(kTab[0]), order, flags); }
}  // namespace symode
