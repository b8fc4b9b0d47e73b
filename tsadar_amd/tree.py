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


class DiffParams:
    """The trainable leaves of a ThomsonParams: {slot: array[B]} in ravel order."""

    def __init__(self, slots, values):
        self.slots = slots  # list of (name, slot)
        self.values = values  # list of arrays [B]

    def ravel(self) -> np.ndarray:
        return np.concatenate(self.values) if self.values else np.zeros(0)

    def like(self, flat: np.ndarray) -> "DiffParams":
        B = self.values[0].shape[0] if self.values else 0
        return DiffParams(self.slots, [np.asarray(flat[i * B : (i + 1) * B], dtype=np.float64) for i in range(len(self.slots))])

    def as_dict(self):
        return {name: v for (name, _), v in zip(self.slots, self.values)}


class StaticParams:
    """Everything that is not trained (the full container; trainable leaves are overwritten on combine)."""

    def __init__(self, ts_params: ThomsonParams):
        self.ts_params = ts_params


def get_filter_spec(cfg_params, ts_params: ThomsonParams):
    """Which leaves are trainable: [(name, slot)] in ravel order."""
    return list(ts_params.slots.active_leaves)


def partition(ts_params: ThomsonParams, filter_spec=None) -> Tuple[DiffParams, StaticParams]:
    spec = filter_spec if filter_spec is not None else get_filter_spec(None, ts_params)
    return DiffParams(list(spec), [ts_params.X[:, s].copy() for _, s in spec]), StaticParams(ts_params)


def combine(a, b) -> ThomsonParams:
    diff, static = (a, b) if isinstance(a, DiffParams) else (b, a)
    out = static.ts_params.copy()
    for (_, s), v in zip(diff.slots, diff.values):
        out.X[:, s] = v
    return out


def ravel_pytree(diff: DiffParams) -> Tuple[np.ndarray, Callable[[np.ndarray], DiffParams]]:
    return diff.ravel(), diff.like
