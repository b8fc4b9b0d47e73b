"""Minimal stand-ins for the three pytree helpers the reference's optimiser loops use around the
loss function (tsadar/inverse/loops.py:40-41, 54: ``eqx.partition``, ``ravel_pytree``,
``eqx.combine``; tsadar/core/modules/ts_params.py:648-685: ``get_filter_spec``), operating on
:class:`tsadar_amd.params.ThomsonParams`.  The flat ordering is the reference's: parameter-major,
pytree field order (Te, ne, m, Ti_k, Z_k, lam, amp1, amp2, amp3, ne_gradient, Te_gradient, ud, Va).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from .params import ThomsonParams


FVAL_SLOT = -1  # pseudo-slot of the free-form distribution-function leaf (Arbitrary1V.fval, [B, nvx])
GEN2D_SLOT = -2  # pseudo-slot of the trainable scalars of a 2-D generator (SphericalHarmonics), [1, n]
FVAL2D_SLOT = -3  # pseudo-slot of the free-form 2-D distribution-function leaf (Arbitrary2V.fval, [nvx, nvx])


class DiffParams:
    """The trainable leaves of a ThomsonParams in ravel order: arrays [B] per scalar leaf; the free-form
    distribution function is one entry of shape [B, nvx] (B leaves of nvx values each, lineout-major, exactly the
    order ravel_pytree gives the list of Arbitrary1V modules)."""

    def __init__(self, slots, values):
        self.slots = slots  # list of (name, slot)
        self.values = values  # list of arrays [B] (or [B, nvx])

    def ravel(self) -> np.ndarray:
        return np.concatenate([np.ravel(v) for v in self.values]) if self.values else np.zeros(0)

    def like(self, flat: np.ndarray) -> "DiffParams":
        out, o = [], 0
        for v in self.values:
            out.append(np.asarray(flat[o : o + v.size], dtype=np.float64).reshape(v.shape))
            o += v.size
        return DiffParams(self.slots, out)

    def as_dict(self):
        return {name: v for (name, _), v in zip(self.slots, self.values)}


class StaticParams:
    """Everything that is not trained (the full container; trainable leaves are overwritten on combine)."""

    def __init__(self, ts_params: ThomsonParams):
        self.ts_params = ts_params


def get_filter_spec(cfg_params, ts_params: ThomsonParams):
    """Which leaves are trainable: [(name, slot)] in ravel order (electron leaves, then the free-form distribution
    function, then ions and general: the field order of ElectronParams, ts_params.py:46-56)."""
    sm = ts_params.slots
    spec = []
    for k, (name, s) in enumerate(sm.leaves):
        if k == sm.n_electron_leaves and sm.fval_active:
            spec.append((("electron", "fval"), FVAL_SLOT))
        if k == sm.n_electron_leaves and sm.gen2d_active:
            spec.append((("electron", "fe"), GEN2D_SLOT))
        if k == sm.n_electron_leaves and getattr(sm, "fval2d_active", False):
            spec.append((("electron", "fval"), FVAL2D_SLOT))
        if sm.active[s]:
            spec.append((name, s))
    return spec


def partition(ts_params: ThomsonParams, filter_spec=None) -> Tuple[DiffParams, StaticParams]:
    spec = filter_spec if filter_spec is not None else get_filter_spec(None, ts_params)
    vals = [ts_params.fval.copy() if s == FVAL_SLOT else ts_params.sph.get_params()[None, :] if s == GEN2D_SLOT
            else ts_params.fval2d.copy() if s == FVAL2D_SLOT else ts_params.X[:, s].copy() for _, s in spec]
    return DiffParams(list(spec), vals), StaticParams(ts_params)


def combine(a, b) -> ThomsonParams:
    diff, static = (a, b) if isinstance(a, DiffParams) else (b, a)
    out = static.ts_params.copy()
    for (_, s), v in zip(diff.slots, diff.values):
        if s == FVAL_SLOT:
            out.fval = np.array(v, dtype=np.float64)
        elif s == GEN2D_SLOT:
            out.sph.set_params(v)
        elif s == FVAL2D_SLOT:
            out.fval2d = np.array(v, dtype=np.float64).reshape(out.fval2d.shape)
        else:
            out.X[:, s] = v
    return out


def ravel_pytree(diff: DiffParams) -> Tuple[np.ndarray, Callable[[np.ndarray], DiffParams]]:
    return diff.ravel(), diff.like


# ---- what the adam loop of the reference needs around vg_loss (loops.py:74-93) -----------------------------
def tree_map(fn, *diffs: DiffParams) -> DiffParams:
    """Leaf-wise map over DiffParams of identical structure (jax.tree_util.tree_map for this container)."""
    return DiffParams(diffs[0].slots, [fn(*vs) for vs in zip(*(d.values for d in diffs))])


def apply_updates(diff: DiffParams, updates: DiffParams) -> DiffParams:
    """eqx.apply_updates / optax.apply_updates."""
    return tree_map(lambda p, u: p + u, diff, updates)


class Adam:
    """optax.adam(learning_rate) for DiffParams (b1 = 0.9, b2 = 0.999, eps = 1e-8, eps_root = 0, bias-corrected):
    ``state = opt.init(params)``, ``updates, state = opt.update(grads, state)``."""

    def __init__(self, learning_rate: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, b1, b2, eps

    def init(self, params: DiffParams):
        z = tree_map(np.zeros_like, params)
        return (0, z, tree_map(np.zeros_like, params))

    def update(self, grads: DiffParams, state, params=None):
        count, mu, nu = state
        count += 1
        mu = tree_map(lambda m, g: self.b1 * m + (1 - self.b1) * g, mu, grads)
        nu = tree_map(lambda v, g: self.b2 * v + (1 - self.b2) * g * g, nu, grads)
        c1, c2 = 1 - self.b1**count, 1 - self.b2**count
        upd = tree_map(lambda m, v: -self.lr * (m / c1) / (np.sqrt(v / c2) + self.eps), mu, nu)
        return upd, (count, mu, nu)
