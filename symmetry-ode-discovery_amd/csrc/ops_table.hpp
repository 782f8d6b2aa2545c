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

inline const LibOps* find_in(const LibOps* tab, int n, int order, int flags) {
    for (int i = 0; i < n; ++i)
        if (tab[i].order == order && tab[i].flags == flags) return &tab[i];
    return nullptr;
}

const LibOps* ops_d1(int order, int flags);
const LibOps* ops_d2(int order, int flags);
const LibOps* ops_d3(int order, int flags);
const LibOps* ops_d4(int order, int flags);

}  // namespace symode
