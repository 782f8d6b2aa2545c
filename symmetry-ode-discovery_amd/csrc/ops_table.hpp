// One translation unit per state dimension D instantiates every kernel for its libraries.
#pragma once
#include "gram.hpp"
#include "gram_valu.hpp"
#include "kernels.hpp"
#include "weak.hpp"

namespace symode {

template <int D, int ORDER, int FLAGS>
constexpr LibOps make_ops() {
    using Lib = Library<D, ORDER, FLAGS>;
    return LibOps{D,
                  ORDER,
                  FLAGS,
                  Lib::P,
                  &launch_theta<Lib>,
                  &launch_forward<Lib>,
                  &launch_odeint<Lib>,
                  &launch_odeint_traj<Lib>,
                  &launch_loss_grad<Lib>,
                  &launch_symreg_linear<Lib>,
                  &launch_symreg_reversed<Lib>,
                  &launch_aug_gram_any<Lib>,
                  &launch_vjp<Lib>,
                  &launch_forward_jvp<Lib>,
                  &launch_jvp_vjp<Lib>,
                  &launch_rk4_traj<Lib>,
                  &launch_euler_jvp<Lib>,
                  &launch_euler_jvp_vjp<Lib>,
                  &launch_weak_gram<Lib>};
}

#define SYMODE_OPS_ALL_FLAGS(D, O) make_ops<D, O, 0>(), make_ops<D, O, 1>(), make_ops<D, O, 2>(), make_ops<D, O, 3>()

// One translation unit per (D, ORDER) defines its four-entry table (index = flags) and the lookup the C ABI calls.
#define SYMODE_DEFINE_OPS_TU(D, O)                                                              \
    namespace symode {                                                                          \
    static const LibOps kTab_##D##_##O[] = {SYMODE_OPS_ALL_FLAGS(D, O)};                        \
    const LibOps* ops_d##D##_o##O(int flags) {                                                  \
        return (flags >= 0 && flags <= 3) ? &kTab_##D##_##O[flags] : nullptr;                   \
    }                                                                                           \
    }

// Large libraries (D >= 3): one translation unit per (D, ORDER, FLAGS); the (D, ORDER) lookup is assembled here.
#define SYMODE_DEFINE_OPS_TU_FLAG(D, O, F)                                                      \
    namespace symode {                                                                          \
    static const LibOps kOne_##D##_##O##_##F = make_ops<D, O, F>();                             \
    const LibOps* ops_d##D##_o##O##_f##F() { return &kOne_##D##_##O##_##F; }                    \
    }
#define SYMODE_SPLIT_OPS(D, O)                                                                  \
    const LibOps* ops_d##D##_o##O##_f0();                                                       \
    const LibOps* ops_d##D##_o##O##_f1();                                                       \
    const LibOps* ops_d##D##_o##O##_f2();                                                       \
    const LibOps* ops_d##D##_o##O##_f3();                                                       \
    inline const LibOps* ops_d##D##_o##O(int flags) {                                           \
        switch (flags) {                                                                        \
            case 0: return ops_d##D##_o##O##_f0();                                              \
            case 1: return ops_d##D##_o##O##_f1();                                              \
            case 2: return ops_d##D##_o##O##_f2();                                              \
            case 3: return ops_d##D##_o##O##_f3();                                              \
            default: return nullptr;                                                            \
        }                                                                                       \
    }

#define SYMODE_DECLARE_OPS(D, O) const LibOps* ops_d##D##_o##O(int flags);
SYMODE_DECLARE_OPS(1, 1) SYMODE_DECLARE_OPS(1, 2) SYMODE_DECLARE_OPS(1, 3) SYMODE_DECLARE_OPS(1, 4) SYMODE_DECLARE_OPS(1, 5)
SYMODE_DECLARE_OPS(2, 1) SYMODE_DECLARE_OPS(2, 2) SYMODE_DECLARE_OPS(2, 3) SYMODE_DECLARE_OPS(2, 4) SYMODE_DECLARE_OPS(2, 5)
SYMODE_SPLIT_OPS(3, 1) SYMODE_SPLIT_OPS(3, 2) SYMODE_SPLIT_OPS(3, 3) SYMODE_SPLIT_OPS(3, 4)
SYMODE_SPLIT_OPS(4, 1) SYMODE_SPLIT_OPS(4, 2) SYMODE_SPLIT_OPS(4, 3)

}  // namespace symode
