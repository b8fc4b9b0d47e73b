"""Host-side mirror of the reference's parameter container ``ThomsonParams``
(tsadar/core/modules/ts_params.py:498-645) without equinox/JAX.

The reference keeps every fit parameter as a *normalised* leaf of a pytree; the physical value is
``act(x) * (ub - lb) + lb`` with ``act`` = sigmoid for active leaves when ``activate`` is set
(ts_params.py:329-350).  The engine evaluates that transform -- and its chain rule -- on the GPU;
this class only stores the leaves, hands them to the engine as one ``[B, NP]`` matrix in the slot
order of ``include/tsff.h``, and reproduces the reference's flat ordering for scipy
(``ravel_pytree`` of the active leaves, loops.py:40-41: parameter-major, field order of the pytree).
"""
from __future__ import annotations

import copy
from typing import Dict, List

import numpy as np

from . import _lib as L

GENERAL_KEYS = ["lam", "amp1", "amp2", "amp3", "ne_gradient", "Te_gradient", "ud", "Va"]  # ts_params.py:396-403
_GENERAL_SLOT = {
    "lam": L.P_LAM, "amp1": L.P_AMP1, "amp2": L.P_AMP2, "amp3": L.P_AMP3,
    "ne_gradient": L.P_NE_GRADIENT, "Te_gradient": L.P_TE_GRADIENT, "ud": L.P_UD, "Va": L.P_VA,
}


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _inv_act(x):
    # ts_params.py:344 / base.py:259 -- deliberately not the exact inverse of the sigmoid
    return np.log(1e-2 + x / (1 - x + 1e-2))


def ion_species(param_cfg) -> List[str]:
    return [k for k in param_cfg.keys() if "ion" in k]


class SlotMap:
    """Where each leaf of the reference pytree lives in the engine's ``[B, NP]`` matrix, and the
    affine/activation description of every slot."""

    def __init__(self, param_cfg: Dict, activate: bool):
        self.param_cfg = param_cfg
        self.activate = activate
        self.species = ion_species(param_cfg)
        assert len(self.species) > 0, "No ion species found in input deck"  # ts_params.py:537
        self.n_ion = len(self.species)
        if self.n_ion > L.MAX_ION:
            raise NotImplementedError(f"at most {L.MAX_ION} ion species")
        self.NP = L.n_params(self.n_ion)
        self.scale = np.ones(self.NP)
        self.shift = np.zeros(self.NP)
        self.sigmoid = np.zeros(self.NP, dtype=np.uint8)
        self.active = np.zeros(self.NP, dtype=bool)
        self.ti_same = np.zeros(L.MAX_ION, dtype=np.uint8)
        # (name, slot) in the pytree/ravel order of the reference:
        # electron(normed_Te, normed_ne, distribution_functions[m]), ions[k](normed_Ti, normed_Z, fract), general(...)
        self.leaves: List[tuple] = []

        el = param_cfg["electron"]
        for k, s in (("Te", L.P_TE), ("ne", L.P_NE)):
            self._affine(s, el[k], el[k]["ub"] - el[k]["lb"], el[k]["lb"])
            self.leaves.append((("electron", k), s))
        fe = el.get("fe", {})
        self.fe_type = str(fe.get("type", "dlm")).casefold()
        self.has_m = self.fe_type == "dlm"
        if self.has_m:
            self._affine(L.P_M, {"active": bool(fe.get("active", False))}, 3.0, 2.0)  # base.py:252-253
            self.leaves.append((("electron", "m"), L.P_M))
        # free-form 1-D distribution (Arbitrary1V): nvx trainable values per lineout, raveled right after (Te, ne)
        self.has_fval = self.fe_type == "arbitrary" and int(fe.get("dim", 1)) == 1
        self.fval_active = self.has_fval and bool(fe.get("active", False))
        # 2-D generator with a few trainable scalars (SphericalHarmonics): one leaf [1, n] at the same position
        self.gen2d_active = "sph" in self.fe_type and int(fe.get("dim", 1)) == 2 and bool(fe.get("active", False))
        # free-form 2-D distribution (Arbitrary2V.fval, base.py:457-465): one [nvx, nvx] leaf at the same position
        self.fval2d_active = self.fe_type == "arbitrary" and int(fe.get("dim", 1)) == 2 and bool(fe.get("active", False))
        self.n_electron_leaves = len(self.leaves)
        for i, sp in enumerate(self.species):
            ic = param_cfg[sp]
            o = L.P_ION0 + 4 * i
            for k, off in (("Ti", L.ION_TI), ("Z", L.ION_Z)):
                self._affine(o + off, ic[k], ic[k]["ub"] - ic[k]["lb"], ic[k]["lb"])
                self.leaves.append(((sp, k), o + off))
            self._affine(o + L.ION_FRACT, ic["fract"], 1.0, 0.0)  # ts_params.py:274,321: no scale/shift
            self.active[o + L.ION_FRACT] = False  # `fract` is not a normed_* leaf: never trainable (:648-685)
            self.leaves.append(((sp, "fract"), o + L.ION_FRACT))
            self.ti_same[i] = 1 if (i > 0 and ic["Ti"].get("same", False)) else 0
        g = param_cfg["general"]
        for k in GENERAL_KEYS:
            s = _GENERAL_SLOT[k]
            self._affine(s, g[k], g[k]["ub"] - g[k]["lb"], g[k]["lb"])
            self.leaves.append((("general", k), s))

    def _affine(self, slot, pc, scale, shift):
        self.scale[slot] = scale
        self.shift[slot] = shift
        self.active[slot] = bool(pc.get("active", False))
        self.sigmoid[slot] = 1 if (self.activate and pc.get("active", False)) else 0

    def a_slots(self):
        return [L.P_ION0 + 4 * i + L.ION_A for i in range(self.n_ion)]

    @property
    def active_leaves(self):
        """[(species, key), slot] of the trainable leaves in ravel order."""
        return [(n, s) for (n, s) in self.leaves if self.active[s]]


class ThomsonParams:
    """Normalised parameters of a batch of lineouts.

    Same constructor and accessors as the reference class (``ThomsonParams(param_cfg, num_params,
    batch=True, activate=False)``, ``__call__``, ``get_unnormed_params``, ``get_fitted_params``).
    """

    def __init__(self, param_cfg: Dict, num_params: int, batch: bool = True, activate: bool = False):
        self.param_cfg = param_cfg
        self.num_params = int(num_params)
        self.batch = batch
        self.activate = activate
        self.slots = SlotMap(param_cfg, activate)
        sm = self.slots
        B = self.num_params if batch else 1
        X = np.zeros((B, sm.NP))
        el = param_cfg["electron"]
        X[:, L.P_TE] = self._init(el["Te"]["val"], L.P_TE)
        X[:, L.P_NE] = self._init(el["ne"]["val"], L.P_NE)
        if sm.has_m:
            X[:, L.P_M] = self._init(el["fe"]["params"]["m"]["val"], L.P_M)
        else:
            X[:, L.P_M] = 2.0
        for i, sp in enumerate(sm.species):
            ic = param_cfg[sp]
            o = L.P_ION0 + 4 * i
            X[:, o + L.ION_TI] = self._init(ic["Ti"]["val"], o + L.ION_TI)
            X[:, o + L.ION_Z] = self._init(ic["Z"]["val"], o + L.ION_Z)
            X[:, o + L.ION_A] = ic["A"]["val"]
            X[:, o + L.ION_FRACT] = self._init(ic["fract"]["val"], o + L.ION_FRACT)
        g = param_cfg["general"]
        for k in GENERAL_KEYS:
            X[:, _GENERAL_SLOT[k]] = self._init(g[k]["val"], _GENERAL_SLOT[k])
        self.fval = None
        if sm.has_fval:  # ts_params.py:143-147: one Arbitrary1V per lineout
            from . import distribution as D

            fv = D.arbitrary_1v_init(float(el["fe"]["params"]["init_m"]), int(el["fe"]["nvx"]))
            self.fval = np.tile(fv[None, :], (B, 1))
        # 2-D distribution function (ts_params.py:152-163): one shared table, never batched
        self.fe_dim = int(el.get("fe", {}).get("dim", 1))
        self.fval2d = None
        if self.fe_dim == 2:
            from . import distribution as D

            if batch:
                raise NotImplementedError(
                    "Batch mode not implemented for 2D distributions as a precautionary measure against memory issues")
            fe = el["fe"]
            self.sph = None
            if "sph" in str(fe["type"]).casefold():  # ts_params.py:157-158
                self.sph = D.SphericalHarmonics(fe)
            elif str(fe["type"]).casefold() == "arbitrary":
                self.learn_log = bool(fe["params"]["learn_log"])
                self.fval2d = D.arbitrary_2v_init(float(fe["params"]["init_m"]), int(fe["nvx"]), self.learn_log)
            else:
                raise NotImplementedError(f"Unknown 2D distribution type: {fe['type']}")
        self.X = X  # [B, NP] normalised leaves, engine layout

    def _init(self, val, slot):
        sm = self.slots
        v = (val - sm.shift[slot]) / sm.scale[slot]
        return _inv_act(v) if sm.sigmoid[slot] else v

    # ---- engine-facing ----------------------------------------------------------------------
    def to_matrix(self) -> np.ndarray:
        return self.X

    def copy(self) -> "ThomsonParams":
        other = copy.copy(self)
        other.X = self.X.copy()
        if self.fval is not None:
            other.fval = self.fval.copy()
        if getattr(self, "sph", None) is not None:
            other.sph = copy.deepcopy(self.sph)
        if getattr(self, "fval2d", None) is not None:
            other.fval2d = self.fval2d.copy()
        return other

    # ---- scipy-facing: the reference's ravel_pytree(diff_params) ordering -----------------------
    def ravel_active(self) -> np.ndarray:
        return np.concatenate([self.X[:, s] for _, s in self.slots.active_leaves]) if self.slots.active_leaves else np.zeros(0)

    def with_active(self, flat: np.ndarray) -> "ThomsonParams":
        out = self.copy()
        B = self.X.shape[0]
        for i, (_, s) in enumerate(self.slots.active_leaves):
            out.X[:, s] = flat[i * B : (i + 1) * B]
        return out

    def ravel_grad(self, grad: np.ndarray) -> np.ndarray:
        """[B, NP] engine gradient -> flat vector in ravel order."""
        return np.concatenate([grad[:, s] for _, s in self.slots.active_leaves]) if self.slots.active_leaves else np.zeros(0)

    def grad_mask(self) -> np.ndarray:
        return self.slots.active.astype(np.uint8)

    # ---- reference accessors -------------------------------------------------------------------
    def physical_matrix(self) -> np.ndarray:
        """[B, NP] physical values (host evaluation of ThomsonParams.__call__, reporting only)."""
        sm = self.slots
        P = np.where(sm.sigmoid[None, :].astype(bool), _sigmoid(self.X), self.X) * sm.scale[None, :] + sm.shift[None, :]
        fr = [L.P_ION0 + 4 * i + L.ION_FRACT for i in range(sm.n_ion)]
        for i in range(1, sm.n_ion):
            if sm.ti_same[i]:
                P[:, L.P_ION0 + 4 * i + L.ION_TI] = P[:, L.P_ION0 + L.ION_TI]
        P[:, fr] = P[:, fr] / np.sum(P[:, fr], axis=1, keepdims=True)
        return P

    def get_unnormed_params(self) -> Dict:
        sm = self.slots
        P = self.physical_matrix()
        sq = (lambda a: a) if self.batch else (lambda a: a[0])
        out = {"electron": {"Te": sq(P[:, L.P_TE]), "ne": sq(P[:, L.P_NE])}, "general": {}}
        if sm.has_m:
            out["electron"]["m"] = sq(P[:, L.P_M])
        if self.fval is not None:  # Arbitrary1V.get_unnormed_params: {"f": self()}
            from . import distribution as D

            out["electron"]["f"] = sq(D.arbitrary_1v(self.fval))
        for k in GENERAL_KEYS:
            out["general"][k] = sq(P[:, _GENERAL_SLOT[k]])
        for i, sp in enumerate(sm.species):
            o = L.P_ION0 + 4 * i
            out[sp] = {
                "A": sq(P[:, o + L.ION_A]),
                "fract": sq(P[:, o + L.ION_FRACT]),
                "Ti": sq(P[:, o + L.ION_TI]),
                "Z": sq(P[:, o + L.ION_Z]),
            }
        return out

    def __call__(self) -> Dict:
        """Physical parameters in the reference's layout; "fe"/"v" are added for DLM / Maxwellian."""
        from . import distribution as D

        out = self.get_unnormed_params()
        nvx = self.param_cfg["electron"]["fe"]["nvx"]
        if self.slots.has_m:
            m = np.atleast_1d(out["electron"].pop("m"))
            fe = np.stack([D.dlm(float(mm), nvx) for mm in m])
            vx = np.tile(D.velocity_grid(nvx)[None, :], (len(m), 1))
            out["electron"]["fe"] = fe if self.batch else fe[0]
            out["electron"]["v"] = vx if self.batch else vx[0]
        if self.fval is not None:
            fe = D.arbitrary_1v(self.fval)
            out["electron"].pop("f", None)
            out["electron"]["fe"] = fe if self.batch else fe[0]
            out["electron"]["v"] = np.tile(D.velocity_grid(nvx)[None, :], (fe.shape[0], 1)) if self.batch else D.velocity_grid(nvx)
        if self.fe_dim == 2:
            out["electron"]["fe"] = self.sph() if self.sph is not None else D.arbitrary_2v(self.fval2d, self.learn_log)
            out["electron"]["v"] = D.velocity_grid(nvx)
        return out

    def get_fitted_params(self, param_cfg) -> tuple:
        """ts_params.py:605-645: the active parameters and their count."""
        pd = self.get_unnormed_params()
        fitted, n = {}, 0
        for k in pd:
            fitted[k] = {}
            for k2 in pd[k]:
                if k2 in ("m", "f"):
                    if param_cfg[k]["fe"]["active"]:
                        fitted[k][k2] = pd[k][k2]
                        n += 1
                elif param_cfg[k][k2]["active"]:
                    fitted[k][k2] = pd[k][k2]
                    n += 1
        return fitted, n
