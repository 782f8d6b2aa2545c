"""Test double: an engine with HipEngine's interface backed by the CPU oracle.

Lives under tests/ on purpose: it lets the ``-m "not gpu"`` suite drive the product's HOST
logic (STLSQ from the Gram, constraint handling, trainer control flow, sharding) without a
GPU.  The product never imports it; its only engine is the HIP one.
"""
import torch

from oracle import sindy_oracle as O


def _fl(flags):
    return bool(flags & 1), bool(flags & 2)


class OracleEngine:
    def lib_size(self, d, order, flags):
        return O.term_count(d, order, *_fl(flags))

    def theta(self, x, order, flags=0):
        return O.theta(x, order, *_fl(flags))

    def forward(self, x, xi, mask, order, flags=0):
        m = torch.ones_like(xi) if mask is None else mask
        return O.forward(x, xi, m, order, *_fl(flags))

    def forward_jvp(self, x, v, xi, mask, order, flags=0, need_out=True):
        m = torch.ones_like(xi) if mask is None else mask
        out, jv = torch.autograd.functional.jvp(lambda a: O.forward(a, xi, m, order, *_fl(flags)), x, v)
        return out, jv

    def vjp(self, x, g, xi, mask, order, flags=0, need_grad_x=True):
        m = torch.ones_like(xi) if mask is None else mask
        with torch.enable_grad():
            xx = x.detach().clone().requires_grad_(True)
            w = xi.detach().clone().requires_grad_(True)
            out = O.forward(xx, w, m, order, *_fl(flags))
            gx, gw = torch.autograd.grad(out, (xx, w), g)
        return (gx if need_grad_x else None), gw

    def jvp_vjp(self, x, v, g_out, g_jv, xi, mask, order, flags=0):
        """reverse mode of forward_jvp: (grad_x, grad_v, grad_xi)"""
        m = torch.ones_like(xi) if mask is None else mask
        with torch.enable_grad():
            xx = x.detach().clone().requires_grad_(True)
            vv = v.detach().clone().requires_grad_(True)
            w = xi.detach().clone().requires_grad_(True)
            out, jv = torch.autograd.functional.jvp(lambda a: O.forward(a, w, m, order, *_fl(flags)), xx, vv, create_graph=True)
            s = (jv * g_jv).sum() + (0.0 if g_out is None else (out * g_out).sum())
            gx, gv, gw = torch.autograd.grad(s, (xx, vv, w), allow_unused=True)
        z = torch.zeros_like
        return (z(x) if gx is None else gx), (z(v) if gv is None else gv), (z(xi) if gw is None else gw)

    def odeint(self, x, xi, mask, order, flags, n_steps, dt, method="euler"):
        m = torch.ones_like(xi) if mask is None else mask
        f = lambda a: O.forward(a, xi, m, order, *_fl(flags))  # noqa: E731
        return O.odeint(f, x, n_steps * dt + 0.5 * dt, dt, method) if n_steps > 0 else x.clone()

    def loss_grad(self, x, dx, xi, mask, order, flags=0, inv_count=None, out=None):
        batched = x.dim() == 3
        X, DX = (x, dx) if batched else (x[None], dx[None])
        S, n, d = X.shape
        XI = xi.reshape(S, d, -1)
        M = torch.ones_like(XI) if mask is None else mask.reshape(S, d, -1)
        losses, grads = [], []
        for s in range(S):
            with torch.enable_grad():          # callers sit inside autograd.Function.forward (grad mode off)
                l, g = O.mse_loss_and_grad(X[s], DX[s], XI[s], M[s], order, *_fl(flags))
            scale = 1.0 if inv_count is None else inv_count * n * d
            losses.append(l * scale)
            grads.append(g * scale)
        loss, grad = torch.stack(losses), torch.stack(grads)
        return (loss, grad) if batched else (loss[0], grad[0])

    def aug_gram(self, x, dx, order, flags=0):
        batched = x.dim() == 3
        X, DX = (x, dx) if batched else (x[None], dx[None])
        out = []
        for s in range(X.shape[0]):
            A = torch.cat([O.theta(X[s], order, *_fl(flags)), DX[s]], dim=1).double()
            out.append(A.T @ A)
        G = torch.stack(out)
        return G if batched else G[0]

    def aug_gram_gather(self, x, dx, idx, order, flags=0):
        return torch.stack([self.aug_gram(x[i.long()], dx[i.long()], order, flags) for i in idx])

    def symreg_linear(self, z, xi, mask, L, order, flags=0):
        d = z.shape[-1]
        reg = O.OracleRegressor(d, order, *_fl(flags), Xi0=xi)
        reg.mask = torch.ones_like(xi) if mask is None else mask
        with torch.enable_grad():
            loss = O.symreg_linear_latent(z, list(L.reshape(-1, d, d)), reg)
            loss.backward()
        return loss.detach(), reg.Xi.grad

    def symreg_reversed(self, x, gx, jgx, xi, mask, order, flags=0, out=None, ws=None, inv_count=None):
        batched = x.dim() == 3
        X, GX, JGX = (x, gx, jgx) if batched else (x[None], gx[None], jgx[None])
        S, n, d = X.shape
        XI = xi.reshape(S, d, -1)
        M = torch.ones_like(XI) if mask is None else mask.reshape(S, d, -1)
        scale = 1.0 if inv_count is None else inv_count * n * d
        losses, grads = [], []
        for s in range(S):
            reg = O.OracleRegressor(d, order, *_fl(flags), Xi0=XI[s])
            reg.mask = M[s]
            with torch.enable_grad():
                loss = O.symreg_reversed_precomputed(X[s], list(GX[s]), list(JGX[s]), reg)
                loss.backward()
            losses.append(loss.detach() * scale)
            grads.append(reg.Xi.grad * scale)
        loss, grad = torch.stack(losses), torch.stack(grads)
        return (loss, grad) if batched else (loss[0], grad[0])
