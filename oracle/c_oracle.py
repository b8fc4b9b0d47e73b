"""TEST INFRASTRUCTURE ONLY.  ctypes face of oracle/c/tsadar_oracle.cpp (C++/OpenMP restatement of the reference's 1-D
path with forward-mode dual-number gradients).  Built on demand with g++ into oracle/_build/ (git-ignored; travels to the
GPU box with the snapshot).  Never imported by tsadar_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import tsadar_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "c", "tsadar_oracle.cpp")
OUT = os.path.join(_HERE, "_build", "libtsadar_oracle.so")
_dp = C.POINTER(C.c_double)
_bp = C.POINTER(C.c_uint8)
LOSS = {"l2": 0, "l1": 1, "log-cosh": 2, "poisson": 3}


class Deck(C.Structure):
    _fields_ = [
        ("lamrangE", C.c_double * 2), ("lamrangI", C.c_double * 2), ("npts", C.c_int32), ("load_ele", C.c_int32),
        ("load_ion", C.c_int32), ("ele_lam_shift", C.c_double), ("n_angles", C.c_int32), ("sa_deg", _dp), ("sa_w", _dp),
        ("G", C.c_int32), ("n_ion", C.c_int32), ("nvx", C.c_int32), ("fe", _dp), ("zr", _dp), ("zi", _dp),
        ("sigma_e", C.c_double), ("sigma_i", C.c_double), ("filt_on", C.c_int32), ("filt_od", C.c_double),
        ("filt_w", C.c_double), ("filt_c", C.c_double), ("scale", _dp), ("shift", _dp), ("sig", _bp), ("ti_same", _bp),
        ("loss_method", C.c_int32), ("fit_iaw", C.c_int32), ("fit_blue", C.c_int32), ("fit_red", C.c_int32),
        ("blue_min", C.c_double), ("blue_max", C.c_double), ("red_min", C.c_double), ("red_max", C.c_double),
        ("iaw_min", C.c_double), ("iaw_cf_min", C.c_double), ("iaw_cf_max", C.c_double), ("iaw_max", C.c_double),
    ]


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["g++", "-O3", "-mavx2", "-mfma", "-std=c++17", "-fopenmp", "-shared", "-fPIC", "-o", OUT, SRC], check=True)
    return OUT


_lib = None


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.orc_loss_grad.restype = C.c_int
        lib.orc_loss_grad.argtypes = [C.POINTER(Deck)] + [C.c_void_p] * 7 + [C.c_int32, _dp, C.c_void_p, C.c_int32] + [C.c_void_p] * 4
        lib.orc_chi_table.restype = C.c_int
        lib.orc_chi_table.argtypes = [C.POINTER(Deck), _dp]
        lib.orc_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _arr(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_deck(cfg, sa, fe=None, activate=True):
    """orc_deck + the arrays it points to, from a deck dict (scale / shift / activation flags per slot as
    ThomsonParams builds them, ts_params.py:84-104, 261-306, 422-457)."""
    cfgp, other, data = cfg["parameters"], cfg["other"], cfg["data"]
    species = orc.ion_species(cfgp)
    n_ion = len(species)
    NP = 11 + 4 * n_ion
    scale, shift, sig = np.ones(NP), np.zeros(NP), np.zeros(NP, dtype=np.uint8)

    def put(slot, pc, sc=None, sh=None):
        scale[slot] = pc["ub"] - pc["lb"] if sc is None else sc
        shift[slot] = pc["lb"] if sh is None else sh
        sig[slot] = 1 if (activate and pc.get("active", False)) else 0

    el, g = cfgp["electron"], cfgp["general"]
    put(0, el["Te"]); put(1, el["ne"])
    scale[2], shift[2] = 3.0, 2.0
    for k, slot in (("lam", 3), ("amp1", 4), ("amp2", 5), ("amp3", 6), ("ne_gradient", 7), ("Te_gradient", 8), ("ud", 9), ("Va", 10)):
        put(slot, g[k])
    ti_same = np.zeros(max(n_ion, 1), dtype=np.uint8)
    for s, sp in enumerate(species):
        ic, o = cfgp[sp], 11 + 4 * s
        put(o, ic["Ti"]); put(o + 1, ic["Z"])
        put(o + 3, ic["fract"], 1.0, 0.0)
        ti_same[s] = 1 if (s > 0 and ic["Ti"].get("same", False)) else 0
    nvx = el["fe"]["nvx"]
    if fe is None:
        fe = orc.dlm_fe(float(el["fe"]["params"]["m"]["val"]) if not (activate and el["fe"].get("active")) else
                        float(orc.sigmoid(orc.inv_act((el["fe"]["params"]["m"]["val"] - 2.0) / 3.0)) * 3.0 + 2.0), nvx)
    zr, zi = orc.zprime_tables()
    w0 = np.asarray(sa["weights"])[0]
    keep = dict(sa=_arr(sa["sa"]), w=_arr(np.broadcast_to(np.asarray(w0, dtype=np.float64), np.asarray(sa["sa"]).shape)),
                fe=_arr(fe), zr=_arr(zr), zi=_arr(zi), scale=scale, shift=shift, sig=sig, ti=ti_same)
    d = Deck()
    d.lamrangE[:] = [float(v) for v in other["lamrangE"]]
    d.lamrangI[:] = [float(v) for v in other["lamrangI"]]
    ext = other["extraoptions"]
    d.npts, d.load_ele, d.load_ion = int(other["npts"]), int(bool(ext["load_ele_spec"])), int(bool(ext["load_ion_spec"]))
    d.ele_lam_shift = float(data.get("ele_lam_shift", 0.0))
    d.n_angles = keep["sa"].size
    d.sa_deg, d.sa_w = keep["sa"].ctypes.data_as(_dp), keep["w"].ctypes.data_as(_dp)
    d.G, d.n_ion, d.nvx = int(g["Te_gradient"]["num_grad_points"]), n_ion, int(nvx)
    d.fe, d.zr, d.zi = keep["fe"].ctypes.data_as(_dp), keep["zr"].ctypes.data_as(_dp), keep["zi"].ctypes.data_as(_dp)
    wid = other["PhysParams"]["widIRF"]
    d.sigma_e, d.sigma_i = float(wid["spect_stddev_ele"]), float(wid["spect_stddev_ion"])
    filt = other.get("iawfilter", [0, 0, 0, 0])
    d.filt_on, d.filt_od, d.filt_w, d.filt_c = int(bool(filt[0])), float(filt[1]), float(filt[2]), float(filt[3])
    d.scale, d.shift = scale.ctypes.data_as(_dp), shift.ctypes.data_as(_dp)
    d.sig, d.ti_same = sig.ctypes.data_as(_bp), ti_same.ctypes.data_as(_bp)
    d.loss_method = LOSS[cfg.get("optimizer", {}).get("loss_method", "l2")]
    d.fit_iaw, d.fit_blue, d.fit_red = int(bool(ext.get("fit_IAW"))), int(bool(ext.get("fit_EPWb"))), int(bool(ext.get("fit_EPWr")))
    r = data["fit_rng"]
    for k in ("blue_min", "blue_max", "red_min", "red_max", "iaw_min", "iaw_cf_min", "iaw_cf_max", "iaw_max"):
        setattr(d, k, float(r[k]))
    return d, keep


def loss_grad(cfg, sa, X, batch, w=None, gmask=None, fe=None, nthreads=0, activate=True, want_spectra=True):
    """-> (sums [B, 3], grad [B, NP] or None, ThryE, ThryI).  X [B, NP] normalised leaves in the slot order
    (Te, ne, m, lam, amp1, amp2, amp3, ne_gradient, Te_gradient, ud, Va, then Ti, Z, A, fract per ion)."""
    lib = load()
    d, keep = make_deck(cfg, sa, fe, activate)
    X = _arr(X)
    B, NP = X.shape

    def mat(a):
        if a is None:
            return None
        a = np.asarray(a, dtype=np.float64)
        if a.ndim < 2 or a.shape[0] != B:
            a = np.broadcast_to(a.reshape(1, -1) if a.size > 1 else a.reshape(1, 1), (B, 1024))
        return _arr(a)

    def vec(a):
        return _arr(np.broadcast_to(np.asarray(a, dtype=np.float64).reshape(-1), (B,)))

    ed, idt = mat(batch.get("e_data")), mat(batch.get("i_data"))
    ne_, ni_ = mat(batch.get("noise_e")), mat(batch.get("noise_i"))
    ea, ia = vec(batch["e_amps"]), vec(batch["i_amps"])
    sums = np.zeros((B, 3))
    grad = np.zeros((B, NP)) if gmask is not None else None
    E = np.zeros((B, 1024)) if want_spectra else None
    I = np.zeros((B, 1024)) if want_spectra else None
    wv = _arr(w if w is not None else [1.0, 1.0, 1.0])
    gm = None if gmask is None else _arr(gmask, np.uint8)
    rc = lib.orc_loss_grad(C.byref(d), _ptr(X), _ptr(ed), _ptr(idt), _ptr(ea), _ptr(ia), _ptr(ne_), _ptr(ni_), B,
                           wv.ctypes.data_as(_dp), _ptr(gm), int(nthreads), _ptr(sums), _ptr(grad), _ptr(E), _ptr(I))
    if rc != 0:
        raise RuntimeError(f"orc_loss_grad failed: {rc}")
    return sums, grad, E, I


def chi_table(cfg, sa, fe):
    lib = load()
    d, keep = make_deck(cfg, sa, fe)
    W = np.zeros(1640)
    lib.orc_chi_table(C.byref(d), W.ctypes.data_as(_dp))
    return W


def max_threads() -> int:
    return int(load().orc_max_threads())
