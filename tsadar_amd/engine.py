"""Engine: builds the static ``tsff_config`` from a tsadar input deck and drives libtsff.so.

Device memory is held in torch CUDA tensors (torch is plumbing here: allocation, streams,
torch.distributed); every compute step is a HIP kernel inside libtsff.so reached through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib as L
from . import distribution as D
from .params import SlotMap

_HERE = os.path.dirname(os.path.abspath(__file__))
_DATA = os.path.join(_HERE, "data")

C_LIGHT = 2.99792458e10


def xi_grids():
    """form_factor.py:128-138."""
    minmax, h1 = 8.2, 1024
    xi1 = np.linspace(-minmax - np.sqrt(2.0) / h1, minmax + np.sqrt(2.0) / h1, h1)
    xi2 = np.arange(-minmax, minmax, 0.01)
    return xi1, xi2


def zprime_tables(xi2):
    """form_factor.py:20-45 restricted to |xi| <= 10: linear interpolation of the shipped tables."""
    rd = np.loadtxt(os.path.join(_DATA, "rdWT.txt"))
    im = np.loadtxt(os.path.join(_DATA, "idWT.txt"))
    return np.interp(xi2, rd[:, 0], rd[:, 1]), np.interp(xi2, im[:, 0], im[:, 1])


def log_ratio_table(xi1, xi2):
    """Lg[q][i] = log|(gav + gdif/2) / (gav - gdif/2)| of ratintn.ratcen (ratintn.py:41-49) for
    g = xi1 - xi2[q], i = 0..1021 (the last interval is dropped, :21), zero-padded to 1024 columns.
    With it the Re(chi_e) table of a distribution function is two matrix-vector products."""
    g = xi1[None, :] - xi2[:, None]
    gdif = g[:, 1:-1] - g[:, 0:-2]
    gav = 0.5 * (g[:, 1:-1] + g[:, 0:-2])
    assert np.all(np.abs(gdif) >= 1.0e-4 * np.abs(gav)), "ratcen's small-gdif branch would trigger"
    out = np.zeros((xi2.size, xi1.size))
    out[:, : xi1.size - 2] = np.log(np.abs((gav + 0.5 * gdif) / (gav - 0.5 * gdif)))
    return out


def wavelength_axis_nm(lam_range, npts):
    """The axis the reference hands to the IRF: lamAxis = squeeze(2 pi c / omgs) * 1e7
    (form_factor.py:132-135, 293; generate_spectra.py:163, 191)."""
    lam = np.linspace(lam_range[0], lam_range[1], npts)
    omgs = 2e7 * np.pi * C_LIGHT / lam
    return (2 * np.pi * C_LIGHT / omgs) * 1e7


def gaussian_taps(lam_nm, stddev, cutoff_sigmas):
    """Taps of ``jnp.convolve(x, g, "same")`` with g the Gaussian of irf.py:66-72 / 110-114 sampled
    on the full wavelength axis: y[j] = sum_d g[c + d] x[j - d], c = (n - 1) // 2.  Taps further than
    ``cutoff_sigmas`` standard deviations from the origin are dropped (their relative weight is
    below exp(-cutoff^2 / 2)); ``cutoff_sigmas <= 0`` keeps every non-zero tap."""
    n = lam_nm.size
    origin = (np.amax(lam_nm) + np.amin(lam_nm)) / 2.0
    g = (1.0 / (stddev * np.sqrt(2.0 * np.pi))) * np.exp(-((lam_nm - origin) ** 2.0) / (2.0 * stddev**2.0))
    c = (n - 1) // 2
    if cutoff_sigmas and cutoff_sigmas > 0:
        keep = np.nonzero(np.abs(lam_nm - origin) <= cutoff_sigmas * stddev)[0]
    else:
        keep = np.nonzero(g > 0)[0]
    lo, hi = int(keep[0]), int(keep[-1])
    return np.ascontiguousarray(g[lo : hi + 1]), lo - c


def binned_taps(g, dmin, ppp):
    """Fold the ppp-sample bin average (irf.py:74,124) into the taps of ``gaussian_taps``:
    ybin[p] = (1/ppp) sum_jj y[p ppp + jj],  y[j] = sum_d g[d] x[j - d]  ==>  ybin[p] = sum_s hb[s] x[p ppp + off + s]
    with hb[s] = (1/ppp) sum_jj g[jj - s + nt - 1] and off = -(dmin + nt - 1)."""
    nt = g.size
    hb = np.zeros(nt + ppp - 1)
    for jj in range(ppp):
        # s = jj - t + nt - 1 for t = 0..nt-1
        hb[jj : jj + nt] += g[::-1]
    return np.ascontiguousarray(hb / ppp), -(dmin + nt - 1)


def _as_c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    ptr_t = L.c_double_p if dtype == np.float64 else L.c_uint8_p
    return a, a.ctypes.data_as(ptr_t)


class Engine:
    """One libtsff handle for one (deck, scattering angles) pair on the current CUDA/HIP device."""

    def __init__(self, cfg: Dict, scattering_angles: Dict, activate: bool = True, fe_shared: Optional[np.ndarray] = None,
                 fe_mode: Optional[int] = None, irf_cutoff_sigmas: float = 12.0, irf_allow_cut: bool = True):
        import torch

        if not torch.cuda.is_available():
            raise L.TsffError("tsadar_amd needs a HIP device (no CPU fallback)")
        self.torch = torch
        self.lib = L.load()
        self.cfg = cfg
        other, data = cfg["other"], cfg["data"]
        ext = other["extraoptions"]
        self.slots = SlotMap(cfg["parameters"], activate)
        sm = self.slots
        self.n_ion, self.NP = sm.n_ion, sm.NP
        self.npts = int(other["npts"])
        self.load_ele, self.load_ion = bool(ext["load_ele_spec"]), bool(ext["load_ion_spec"])
        if other.get("iawoff", 0):
            raise NotImplementedError("iawoff cannot run under vmap in the reference (generate_spectra.py:199-208)")
        gen = cfg["parameters"]["general"]
        assert gen["Te_gradient"]["num_grad_points"] == gen["ne_gradient"]["num_grad_points"], \
            "Number of gradient points for Te and ne must be the same"  # generate_spectra.py:70-73
        fecfg = cfg["parameters"]["electron"]["fe"]
        self.fe_dim = int(fecfg.get("dim", 1))
        if self.fe_dim not in (1, 2):
            raise NotImplementedError(f"Not implemented distribution dimension: {self.fe_dim}")  # ts_params.py:164
        if self.fe_dim == 2 and fe_mode is None:
            fe_mode = L.FE_PER_LINEOUT  # the 1-D tables are unused: only form_factor_2d / ats_spectrum run
        self.nvx = int(fecfg["nvx"])

        c = L.TsffConfig()
        keep = []  # numpy arrays referenced by the struct
        c.abi_version = L.ABI_VERSION
        c.lamrangE[:] = [float(v) for v in other["lamrangE"]]
        c.lamrangI[:] = [float(v) for v in other["lamrangI"]]
        c.npts = self.npts
        c.load_ele, c.load_ion = int(self.load_ele), int(self.load_ion)
        c.ele_lam_shift = float(data.get("ele_lam_shift", 0.0))
        sa = np.asarray(scattering_angles["sa"], dtype=np.float64)
        w0 = np.asarray(scattering_angles["weights"])[0]  # generate_spectra.py:165,197
        w = np.broadcast_to(np.asarray(w0, dtype=np.float64), sa.shape)
        c.n_angles = sa.size
        a, c.sa_deg = _as_c(sa, np.float64); keep.append(a)
        a, c.sa_weights = _as_c(w, np.float64); keep.append(a)
        c.num_grad_points = int(gen["Te_gradient"]["num_grad_points"])
        c.n_ion = sm.n_ion
        c.nvx = self.nvx

        # distribution function mode
        if fe_mode is None:
            if fe_shared is not None:
                fe_mode = L.FE_SHARED
            elif sm.fe_type == "dlm":
                fe_mode = L.FE_DLM if fecfg.get("active", False) else L.FE_SHARED
            else:
                fe_mode = L.FE_PER_LINEOUT
        self.fe_mode = fe_mode
        c.fe_mode = fe_mode
        if fe_mode == L.FE_SHARED:
            if fe_shared is None:
                if sm.fe_type != "dlm":
                    raise NotImplementedError(f"Unknown 1D distribution type: {fecfg['type']}")
                fe_shared = D.dlm(float(fecfg["params"]["m"]["val"]), self.nvx)
            a, c.fe_shared = _as_c(fe_shared, np.float64); keep.append(a)
            self.fe_shared = a
        if fe_mode == L.FE_DLM:
            a, c.dlm_table = _as_c(D.dlm_table(self.nvx), np.float64); keep.append(a)

        xi1, xi2 = xi_grids()
        assert xi1.size == L.NXI1 and xi2.size == L.NXI2
        zr, zi = zprime_tables(xi2)
        for name, arr in (("xi1", xi1), ("xi2", xi2), ("zprime_re", zr), ("zprime_im", zi)):
            a, p = _as_c(arr, np.float64); keep.append(a); setattr(c, name, p)
        if fe_mode != L.FE_SHARED:
            a, c.lg_table = _as_c(log_ratio_table(xi1, xi2), np.float64); keep.append(a)

        # instrument response
        phys = other["PhysParams"]
        c.norm = int(phys["norm"])
        lamE = wavelength_axis_nm(other["lamrangE"], self.npts)
        lamI = wavelength_axis_nm(other["lamrangI"], self.npts)
        def set_taps(cutoff):
            if self.load_ele:
                t, d0 = gaussian_taps(lamE, float(phys["widIRF"]["spect_stddev_ele"]), cutoff)
                t, off = binned_taps(t, d0, self.npts // L.NBINS)
                a, c.taps_ele = _as_c(t, np.float64); keep.append(a)
                c.n_taps_ele, c.tap_off_ele = t.size, off
            if self.load_ion and phys["widIRF"]["spect_stddev_ion"]:
                t, d0 = gaussian_taps(lamI, float(phys["widIRF"]["spect_stddev_ion"]), cutoff)
                t, off = binned_taps(t, d0, self.npts // L.NBINS)
                a, c.taps_ion = _as_c(t, np.float64); keep.append(a)
                c.n_taps_ion, c.tap_off_ion = t.size, off

        set_taps(irf_cutoff_sigmas)
        filt = other.get("iawfilter", [0, 0, 0, 0])
        if self.load_ele and filt[0]:
            fb, fr = filt[3] - filt[2] / 2, filt[3] + filt[2] / 2
            if other["lamrangE"][0] < fr and other["lamrangE"][1] > fb:
                mult = np.where((fb < lamE) & (fr > lamE), 10.0 ** (-filt[1]), 1.0)
                a, c.ele_filter = _as_c(mult, np.float64); keep.append(a)

        # parameter transform
        a, c.p_scale = _as_c(sm.scale, np.float64); keep.append(a)
        a, c.p_shift = _as_c(sm.shift, np.float64); keep.append(a)
        a, c.p_sigmoid = _as_c(sm.sigmoid, np.uint8); keep.append(a)
        c.ti_same[:] = [int(v) for v in sm.ti_same]

        # loss masks on the binned axes (loss_function.py:224-259)
        method = cfg.get("optimizer", {}).get("loss_method", "l2")
        if method not in L.LOSS_METHODS:
            raise NotImplementedError(f"loss_method {method}")
        c.loss_method = L.LOSS_METHODS[method]
        ppp = self.npts // L.NBINS
        self.lamE_bin = lamE.reshape(L.NBINS, ppp).mean(axis=1)
        self.lamI_bin = lamI.reshape(L.NBINS, ppp).mean(axis=1)
        r = data["fit_rng"]
        mE = np.zeros(L.NBINS, dtype=np.uint8)
        mI = np.zeros(L.NBINS, dtype=np.uint8)
        if ext.get("fit_EPWb", False):
            mE |= ((self.lamE_bin > r["blue_min"]) & (self.lamE_bin < r["blue_max"])).astype(np.uint8)
        if ext.get("fit_EPWr", False):
            mE |= (((self.lamE_bin > r["red_min"]) & (self.lamE_bin < r["red_max"])).astype(np.uint8) << 1)
        if ext.get("fit_IAW", False):
            mI |= (((self.lamI_bin > r["iaw_min"]) & (self.lamI_bin < r["iaw_cf_min"]))
                   | ((self.lamI_bin > r["iaw_cf_max"]) & (self.lamI_bin < r["iaw_max"]))).astype(np.uint8)
        self.mask_ele, self.mask_ion = mE, mI
        a, c.mask_ele = _as_c(mE, np.uint8); keep.append(a)
        a, c.mask_ion = _as_c(mI, np.uint8); keep.append(a)
        self.n_blue = int(np.count_nonzero(mE & 1))
        self.n_red = int(np.count_nonzero(mE & 2))
        self.n_iaw = int(np.count_nonzero(mI & 1))
        self.fit_blue, self.fit_red, self.fit_iaw = bool(ext.get("fit_EPWb")), bool(ext.get("fit_EPWr")), bool(ext.get("fit_IAW"))

        self._keep = keep
        self._staging = {}
        self._cfg_struct = c
        h = C.c_void_p()
        cutoff = irf_cutoff_sigmas
        while True:
            rc = self.lib.tsff_create(C.byref(c), C.byref(h))
            msg = self.lib.tsff_last_error(None) if rc else b""
            # Wide instrument functions at several points per pixel: spectrum + halo + taps outgrow the LDS of a CU.  The taps
            # beyond ~8 sigma only matter below 1e-14 of a spectrum's maximum (they reproduce the reference's full-length
            # convolution in the 1e-22 tails): drop them step by step down to 7 sigma (2e-11) before giving up, and say so.
            # ``irf_allow_cut=False`` makes the overflow an error instead; the cut-off actually used is ``self.irf_cutoff_sigmas``.
            if rc == L.ERR_LDS and irf_allow_cut and cutoff and cutoff > 7.0:
                cutoff = max(7.0, cutoff - 1.0)
                set_taps(cutoff)
                continue
            break
        L.check(self.lib, None, rc)
        if cutoff != irf_cutoff_sigmas:
            import warnings
            warnings.warn(f"tsadar_amd: IRF taps cut at {cutoff} sigma instead of {irf_cutoff_sigmas} to fit the LDS "
                          f"(relative weight of the dropped taps below {np.exp(-0.5 * cutoff ** 2):.1e})")
        self.irf_cutoff_sigmas = cutoff
        self.h = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        # axes as the library computed them
        ae, ai = np.zeros(L.NBINS), np.zeros(L.NBINS)
        L.check(self.lib, self.h, self.lib.tsff_get_axes(self.h, ae.ctypes.data_as(L.c_double_p), ai.ctypes.data_as(L.c_double_p)))
        self.lamAxisE, self.lamAxisI = ae, ai

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            self.lib.tsff_destroy(h)
            self.h = None

    # ---- helpers ------------------------------------------------------------------------------
    def dev(self, a, dtype=None):
        """numpy / torch -> contiguous float64 CUDA tensor on this engine's device (no copy if it
        already is one)."""
        torch = self.torch
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            t = a
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
        return t.to(device=self.device, dtype=dtype or torch.float64).contiguous()

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)

    # ---- host <-> device staging of the per-step vectors (pinned buffers: a fit step moves ~0.5 MB in and ~0.2 MB out) ----
    def upload(self, a: np.ndarray):
        """NumPy array -> device tensor through a persistent pinned staging buffer (asynchronous copy on the current stream).
        One device tensor per array SHAPE: the result aliases the previous upload of the same shape (a fit step consumes
        its parameters before the next step uploads new ones); copy it if it must outlive the next call.  The pinned buffer is
        not rewritten before its previous copy has completed (event per buffer)."""
        torch = self.torch
        a = np.ascontiguousarray(a, dtype=np.float64)
        key = ("in", a.shape)
        buf = self._staging.get(key)
        if buf is None:
            buf = (torch.empty(a.shape, dtype=torch.float64).pin_memory(), torch.empty(a.shape, dtype=torch.float64, device=self.device),
                   torch.cuda.Event())
            self._staging[key] = buf
        else:
            buf[2].synchronize()   # (the earlier H2D copy out of this pinned buffer)
        pin, dev, ev = buf
        pin.numpy()[...] = a
        dev.copy_(pin, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        return dev

    def download(self, t) -> np.ndarray:
        """Device tensor -> NumPy (pinned staging, one synchronisation of the current stream)."""
        torch = self.torch
        key = ("out", tuple(t.shape))
        pin = self._staging.get(key)
        if pin is None:
            pin = torch.empty(tuple(t.shape), dtype=t.dtype).pin_memory()
            self._staging[key] = pin
        pin.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return pin.numpy().copy()

    def _sync_stream(self):
        s = self.torch.cuda.current_stream(self.device)
        L.check(self.lib, self.h, self.lib.tsff_set_stream(self.h, C.c_void_p(s.cuda_stream)))

    def _vec(self, a, B):
        """amplitude-like input -> [B] device vector."""
        t = self.dev(a).reshape(-1)
        if t.numel() == 1 and B > 1:
            t = t.expand(B).contiguous()
        assert t.numel() == B, f"expected {B} amplitudes, got {t.numel()}"
        return t

    def _mat(self, a, B):
        """noise/data-like input -> [B, 1024] device matrix or None for an all-zero scalar."""
        if a is None:
            return None
        if not isinstance(a, self.torch.Tensor):
            arr = np.asarray(a, dtype=np.float64)
            if arr.size == 1 and float(arr.reshape(-1)[0]) == 0.0:
                return None
            if arr.size != B * L.NBINS:
                arr = np.broadcast_to(arr.reshape(-1, 1) if arr.ndim <= 1 and arr.size in (1, B) else arr, (B, L.NBINS))
            a = arr
        t = self.dev(a)
        return t.reshape(B, L.NBINS).contiguous()

    # ---- entry points ---------------------------------------------------------------------------
    def chi_table(self, fe):
        torch = self.torch
        fe_d = self.dev(fe).reshape(-1, self.nvx)
        W = torch.empty((fe_d.shape[0], L.NXI2), dtype=torch.float64, device=self.device)
        self._sync_stream()
        L.check(self.lib, self.h, self.lib.tsff_chi_table(self.h, self._ptr(fe_d), fe_d.shape[0], self._ptr(W)))
        return W

    def form_factor(self, feature, phys, fe=None):
        """Raw FormFactor.__call__: phys [B, NP] PHYSICAL parameters -> P [B, G, npts, n_angles]."""
        torch = self.torch
        phys_d = self.dev(phys).reshape(-1, self.NP)
        B = phys_d.shape[0]
        fe_d = self.dev(fe)
        G, NA = int(self._cfg_struct.num_grad_points), int(self._cfg_struct.n_angles)
        P = torch.empty((B, G, self.npts, NA), dtype=torch.float64, device=self.device)
        self._sync_stream()
        L.check(self.lib, self.h, self.lib.tsff_form_factor(self.h, int(feature), self._ptr(phys_d), self._ptr(fe_d), B, self._ptr(P)))
        return P

    def form_factor_grad(self, feature, phys, fe, Pbar, want_fe=False):
        """Adjoint of form_factor: Pbar [B, G, npts, n_angles] -> (grad_phys [B, NP], grad_fe [B, nvx] or None)."""
        torch = self.torch
        phys_d = self.dev(phys).reshape(-1, self.NP)
        B = phys_d.shape[0]
        fe_d, Pb = self.dev(fe), self.dev(Pbar)
        gp = torch.empty((B, self.NP), dtype=torch.float64, device=self.device)
        gf = torch.empty((B, int(self._cfg_struct.nvx)), dtype=torch.float64, device=self.device) if want_fe else None
        self._sync_stream()
        L.check(self.lib, self.h, self.lib.tsff_form_factor_grad(self.h, int(feature), self._ptr(phys_d), self._ptr(fe_d), B,
                                                                 self._ptr(Pb), self._ptr(gp), self._ptr(gf)))
        return gp, gf

    def form_factor_2d(self, feature, phys, fe2d, ud_angle=0.0, va_angle=0.0, point_range=None, out=None, save=False):
        """FormFactor.calc_in_2D: phys [B, NP] PHYSICAL parameters, fe2d [nv, nv] (shared) or [B, nv, nv]
        -> P [B, G, npts, n_angles].  ``point_range = (begin, end)`` evaluates that slice of the flat point list only
        (the rest of ``out`` / a zero-filled result is untouched): the sharding unit of the multi-GPU path."""
        torch = self.torch
        phys_d = self.dev(phys).reshape(-1, self.NP)
        B = phys_d.shape[0]
        fe_d = self.dev(fe2d)
        shared = fe_d.dim() == 2
        nv = int(fe_d.shape[-1])
        assert fe_d.shape[-2] == nv and (shared or fe_d.shape[0] == B)
        G, NA = int(self._cfg_struct.num_grad_points), int(self._cfg_struct.n_angles)
        if out is not None:
            P = out
        else:
            P = (torch.zeros if point_range is not None else torch.empty)((B, G, self.npts, NA), dtype=torch.float64, device=self.device)
        lo, hi = point_range if point_range is not None else (0, -1)
        self._sync_stream()
        if save and shared and nv <= 256:   # keep the projection records for form_factor_2d_grad(use_saved=True)
            token = C.c_uint64(0)
            rc = self.lib.tsff_form_factor_2d_save(self.h, int(feature), self._ptr(phys_d), self._ptr(fe_d), nv, float(ud_angle),
                                                   float(va_angle), B, int(lo), int(hi), self._ptr(P), C.byref(token))
            # the library ties the records to the buffers they were made from: keep those buffers (and what the caller
            # passed) so that the adjoint can present the very same ones
            keep = lambda a: a if isinstance(a, torch.Tensor) else np.array(a, dtype=np.float64, copy=True)
            self._saved_2d = dict(phys_in=keep(phys), fe_in=keep(fe2d), phys_d=phys_d, fe_d=fe_d, token=int(token.value))
        else:
            self._saved_2d = None
            rc = self.lib.tsff_form_factor_2d_range(self.h, int(feature), self._ptr(phys_d), self._ptr(fe_d), nv, int(shared),
                                                    float(ud_angle), float(va_angle), B, int(lo), int(hi), self._ptr(P))
        L.check(self.lib, self.h, rc)
        return P

    def form_factor_2d_grad(self, feature, phys, fe2d, Pbar, ud_angle=0.0, va_angle=0.0, want_table=True, point_range=None,
                            use_saved=False, saved_token=None):
        """Adjoint of form_factor_2d (one shared table): Pbar [B, G, npts, n_angles] ->
        (grad_phys [B, NP], grad_fe2d [nv, nv] or None) as device tensors.  ``point_range = (begin, end)``: the
        contributions of that slice of the flat point list only (to be summed over the ranks of a node)."""
        torch = self.torch
        saved = getattr(self, "_saved_2d", None) if use_saved else None
        if saved is not None and self._same_input(phys, saved["phys_in"], saved["phys_d"]) and self._same_input(fe2d, saved["fe_in"], saved["fe_d"]):
            phys_d, fe_d = saved["phys_d"], saved["fe_d"]   # the buffers the projection records were made from
        else:   # other inputs than the saving forward saw (or no records): the adjoint does its own sampling
            saved = None
            phys_d, fe_d = self.dev(phys).reshape(-1, self.NP), self.dev(fe2d)
        B = phys_d.shape[0]
        Pb = self.dev(Pbar)
        assert fe_d.dim() == 2 and fe_d.shape[0] == fe_d.shape[1]
        nv = int(fe_d.shape[0])
        gp = torch.empty((B, self.NP), dtype=torch.float64, device=self.device)
        gf = torch.empty((nv, nv), dtype=torch.float64, device=self.device) if want_table else None
        self._sync_stream()
        lo, hi = point_range if point_range is not None else (0, -1)
        rc = self.lib.tsff_form_factor_2d_grad(self.h, int(feature), self._ptr(phys_d), self._ptr(fe_d), nv, float(ud_angle),
                                               float(va_angle), B, int(lo), int(hi),
                                               C.c_uint64(int(saved_token) if saved_token is not None else (saved["token"] if saved is not None else 0)),
                                               self._ptr(Pb), self._ptr(gp), self._ptr(gf))
        L.check(self.lib, self.h, rc)
        return gp, gf

    def _same_input(self, a, ref, ref_dev) -> bool:
        """Is ``a`` what the saving forward was given?  A host array equal to the copy kept at the save, or a device tensor on
        the same memory (a device tensor rewritten in place between the two calls cannot be told apart: do not do that)."""
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a is ref or (a.is_cuda and a.data_ptr() == ref_dev.data_ptr() and a.numel() == ref_dev.numel())
        if isinstance(ref, torch.Tensor):
            return False
        a = np.asarray(a)
        return a.size == ref.size and bool(np.array_equal(a.reshape(ref.shape), ref))

    def ats_setup(self, weights, ang_axis, stddev_lam, stddev_ang, lam_step=1, ang_step=1, row_start=0, row_end=None,
                  irf_cutoff_sigmas=12.0):
        """Static configuration of the angular (ARTS) instrument chain: weight matrix [n_px, n_angles], the calibrated
        angle axis sas["angAxis"] [n_px] and the two Gaussian widths (FWHM / 2.3548, irf.py:23-24)."""
        W = np.ascontiguousarray(weights, dtype=np.float64)
        n_px = W.shape[0]
        assert W.shape[1] == int(self._cfg_struct.n_angles)
        lamE = wavelength_axis_nm(self.cfg["other"]["lamrangE"], self.npts)
        ta, da = gaussian_taps(np.asarray(ang_axis, dtype=np.float64), float(stddev_ang), irf_cutoff_sigmas)
        ta, offa = binned_taps(ta, da, 1)
        tl, dl = gaussian_taps(lamE, float(stddev_lam), irf_cutoff_sigmas)
        tl, offl = binned_taps(tl, dl, 1)
        c = L.TsffAtsConfig()
        c.n_px = n_px
        keep = []
        a, c.weights = _as_c(W, np.float64); keep.append(a)
        a, c.taps_ang = _as_c(ta, np.float64); keep.append(a)
        a, c.taps_lam = _as_c(tl, np.float64); keep.append(a)
        a, c.lam_axis = _as_c(lamE, np.float64); keep.append(a)
        c.n_taps_ang, c.tap_off_ang, c.n_taps_lam, c.tap_off_lam = ta.size, offa, tl.size, offl
        c.lam_step, c.ang_step = int(lam_step), int(ang_step)
        c.row_start = int(row_start)
        c.row_end = int(row_end if row_end is not None else n_px // ang_step)
        L.check(self.lib, self.h, self.lib.tsff_ats_setup(self.h, C.byref(c)))
        self._ats_shape = (c.row_end - c.row_start, self.npts // int(lam_step))

    def ats_spectrum(self, P, e_amps, lam, amp1, amp2):
        """P [G, npts, n_angles] (one image) -> ThryE [rows, npts / lam_step]."""
        torch = self.torch
        Pd = self.dev(P)
        ea = self.dev(np.broadcast_to(np.asarray(e_amps, dtype=np.float64).reshape(-1), (self._ats_shape[0],)).copy()
                      if not torch.is_tensor(e_amps) else e_amps.reshape(-1))
        assert ea.numel() == self._ats_shape[0]
        out = torch.empty(self._ats_shape, dtype=torch.float64, device=self.device)
        self._sync_stream()
        rc = self.lib.tsff_ats_spectrum(self.h, self._ptr(Pd), self._ptr(ea), float(lam), float(amp1), float(amp2), self._ptr(out))
        L.check(self.lib, self.h, rc)
        return out

    def ats_adjoint(self, P, e_amps, lam, amp1, amp2, Ebar):
        """Reverse of ats_spectrum: Ebar [rows, n_lam] -> (Pbar [G, npts, n_angles] device tensor, (amp1_bar, amp2_bar))."""
        torch = self.torch
        Pd, Eb = self.dev(P), self.dev(Ebar)
        ea = self.dev(np.broadcast_to(np.asarray(e_amps, dtype=np.float64).reshape(-1), (self._ats_shape[0],)).copy()
                      if not torch.is_tensor(e_amps) else e_amps.reshape(-1))
        assert tuple(Eb.shape) == tuple(self._ats_shape)
        Pbar = torch.empty_like(Pd)
        ab = (C.c_double * 2)()
        self._sync_stream()
        rc = self.lib.tsff_ats_adjoint(self.h, self._ptr(Pd), self._ptr(ea), float(lam), float(amp1), float(amp2), self._ptr(Eb),
                                       self._ptr(Pbar), ab)
        L.check(self.lib, self.h, rc)
        return Pbar, (float(ab[0]), float(ab[1]))

    def forward(self, params, e_amps, i_amps, noise_e=None, noise_i=None, fe=None):
        torch = self.torch
        X = self.dev(params).reshape(-1, self.NP)
        B = X.shape[0]
        ea = self._vec(e_amps, B) if self.load_ele else None
        ia = self._vec(i_amps, B) if self.load_ion else None
        ne_, ni_ = self._mat(noise_e, B), self._mat(noise_i, B)
        fe_d = self.dev(fe)
        # (the kernel writes every entry of a loaded feature; a feature that is not loaded comes back as zeros)
        E = (torch.empty if self.load_ele else torch.zeros)((B, L.NBINS), dtype=torch.float64, device=self.device)
        I = (torch.empty if self.load_ion else torch.zeros)((B, L.NBINS), dtype=torch.float64, device=self.device)
        self._sync_stream()
        rc = self.lib.tsff_forward(self.h, self._ptr(X), self._ptr(fe_d), self._ptr(ea), self._ptr(ia), self._ptr(ne_),
                                   self._ptr(ni_), B, self._ptr(E), self._ptr(I))
        L.check(self.lib, self.h, rc)
        return E, I

    def loss_weights(self, B_global, i_norm, e_norm, ion_loss_scale=1.0):
        """The factor each masked sum carries in the total loss (loss_function.py:190-267, 335-338)
        for a nanmean over B_global x 1024 entries and constant denominators i_norm^2 / e_norm^2."""
        c = 0.5 if (self.fit_blue and self.fit_red) else 1.0
        w = np.zeros(3)
        if self.fit_iaw and self.n_iaw:
            w[0] = ion_loss_scale / (B_global * self.n_iaw * i_norm**2)
        if self.fit_blue and self.n_blue:
            w[1] = c / (B_global * self.n_blue * e_norm**2)
        if self.fit_red and self.n_red:
            # the reference halves (blue + red) only when blue is fitted too (:262-264)
            w[2] = c / (B_global * self.n_red * e_norm**2)
        if self.cfg.get("optimizer", {}).get("loss_method", "l2") in ("log-cosh", "poisson"):
            # these functionals ignore the denominator (loss_function.py:414-417)
            w[0] *= i_norm**2
            w[1] *= e_norm**2
            w[2] *= e_norm**2
        return w

    def loss_grad(self, params, batch, weights, grad_mask, fe=None, want_spectra=False, out=None, want_fe_grad=False):
        """-> (loss_terms[3], grad[B, NP], ThryE, ThryI) as CUDA tensors; nothing is synchronised.
        ``want_fe_grad`` (fe_mode PER_LINEOUT): a fifth result, d loss / d fe [B, nvx]."""
        torch = self.torch
        X = self.dev(params).reshape(-1, self.NP)
        B = X.shape[0]
        ea = self._vec(batch["e_amps"], B) if self.load_ele else None
        ia = self._vec(batch["i_amps"], B) if self.load_ion else None
        ed = self._mat(batch["e_data"], B) if self.load_ele else None
        idt = self._mat(batch["i_data"], B) if self.load_ion else None
        ne_, ni_ = self._mat(batch.get("noise_e"), B), self._mat(batch.get("noise_i"), B)
        fe_d = self.dev(fe)
        if out is None:
            terms = torch.empty(3, dtype=torch.float64, device=self.device)
            grad = torch.empty((B, self.NP), dtype=torch.float64, device=self.device)
        else:
            terms, grad = out
        E = torch.zeros((B, L.NBINS), dtype=torch.float64, device=self.device) if want_spectra else None
        I = torch.zeros((B, L.NBINS), dtype=torch.float64, device=self.device) if want_spectra else None
        w = np.ascontiguousarray(weights, dtype=np.float64)
        gm = np.ascontiguousarray(grad_mask, dtype=np.uint8)
        self._sync_stream()
        if want_fe_grad:
            gfe = torch.empty((B, self.nvx), dtype=torch.float64, device=self.device)
            rc = self.lib.tsff_loss_grad_fe(self.h, self._ptr(X), self._ptr(fe_d), self._ptr(ed), self._ptr(idt), self._ptr(ea),
                                            self._ptr(ia), self._ptr(ne_), self._ptr(ni_), B, w.ctypes.data_as(L.c_double_p),
                                            gm.ctypes.data_as(L.c_uint8_p), self._ptr(terms), self._ptr(grad), self._ptr(gfe),
                                            self._ptr(E), self._ptr(I))
            L.check(self.lib, self.h, rc)
            return terms, grad, E, I, gfe
        rc = self.lib.tsff_loss_grad(self.h, self._ptr(X), self._ptr(fe_d), self._ptr(ed), self._ptr(idt), self._ptr(ea),
                                     self._ptr(ia), self._ptr(ne_), self._ptr(ni_), B, w.ctypes.data_as(L.c_double_p),
                                     gm.ctypes.data_as(L.c_uint8_p), self._ptr(terms), self._ptr(grad), self._ptr(E), self._ptr(I))
        L.check(self.lib, self.h, rc)
        return terms, grad, E, I

    def pack_fe_rows(self, terms, grad, gfe, active_slots, B_global=None, b_offset=0, out=None):
        """The packed buffer ``[3 | (P + nvx) x B_global]`` of a free-form f_e step from the outputs of ``loss_grad(want_fe_grad=True)``
        (tsff_pack_fe_rows: one transposing kernel, this rank's columns filled, the others zero)."""
        torch = self.torch
        B = int(grad.shape[0])
        Bg = B if B_global is None else int(B_global)
        act = np.ascontiguousarray(active_slots, dtype=np.int32)
        n = 3 + (act.size + self.nvx) * Bg
        packed = out if out is not None else torch.empty(n, dtype=torch.float64, device=self.device)
        assert packed.numel() == n
        self._sync_stream()
        L.check(self.lib, self.h, self.lib.tsff_pack_fe_rows(self.h, self._ptr(terms), self._ptr(grad), self._ptr(gfe), B,
                                                             act.ctypes.data_as(C.POINTER(C.c_int32)), int(act.size), Bg, int(b_offset),
                                                             self._ptr(packed)))
        return packed

    def loss_grad_packed(self, params, batch, weights, grad_mask, active_slots, B_global=None, b_offset=0, want_spectra=False, out=None):
        """tsff_loss_grad_packed: -> (packed [3 + P * B_global] CUDA tensor, ThryE, ThryI).  packed = [S_iaw, S_blue, S_red |
        gradient, trainable leaves outermost, lineouts of the GLOBAL batch innermost]; this rank's B lineouts fill the
        columns [b_offset, b_offset + B), every other column is written as zero -- the buffer of the step's one in-place
        all-reduce and of the single device-to-host copy.  The buffer is persistent per (P, B_global)."""
        torch = self.torch
        X = self.dev(params).reshape(-1, self.NP)
        B = X.shape[0]
        Bg = int(B_global) if B_global is not None else B
        act = np.ascontiguousarray(active_slots, dtype=np.int32)
        ea = self._vec(batch["e_amps"], B) if self.load_ele else None
        ia = self._vec(batch["i_amps"], B) if self.load_ion else None
        ed = self._mat(batch["e_data"], B) if self.load_ele else None
        idt = self._mat(batch["i_data"], B) if self.load_ion else None
        ne_, ni_ = self._mat(batch.get("noise_e"), B), self._mat(batch.get("noise_i"), B)
        if out is None:
            key = ("packed", act.size, Bg)
            out = self._staging.get(key)
            if out is None:
                out = torch.zeros(3 + act.size * Bg, dtype=torch.float64, device=self.device)
                self._staging[key] = out
        assert out.numel() == 3 + act.size * Bg and out.is_contiguous()
        E = torch.zeros((B, L.NBINS), dtype=torch.float64, device=self.device) if want_spectra else None
        I = torch.zeros((B, L.NBINS), dtype=torch.float64, device=self.device) if want_spectra else None
        w = np.ascontiguousarray(weights, dtype=np.float64)
        gm = np.ascontiguousarray(grad_mask, dtype=np.uint8)
        self._sync_stream()
        rc = self.lib.tsff_loss_grad_packed(self.h, self._ptr(X), None, self._ptr(ed), self._ptr(idt), self._ptr(ea), self._ptr(ia),
                                            self._ptr(ne_), self._ptr(ni_), B, w.ctypes.data_as(L.c_double_p), gm.ctypes.data_as(L.c_uint8_p),
                                            act.ctypes.data_as(C.POINTER(C.c_int32)), int(act.size), Bg, int(b_offset), self._ptr(out),
                                            self._ptr(E), self._ptr(I))
        L.check(self.lib, self.h, rc)
        return out, E, I

    def array_loss(self, params, batch, fe=None):
        torch = self.torch
        X = self.dev(params).reshape(-1, self.NP)
        B = X.shape[0]
        ea = self._vec(batch["e_amps"], B) if self.load_ele else None
        ia = self._vec(batch["i_amps"], B) if self.load_ion else None
        ed = self._mat(batch["e_data"], B) if self.load_ele else None
        idt = self._mat(batch["i_data"], B) if self.load_ion else None
        ne_, ni_ = self._mat(batch.get("noise_e"), B), self._mat(batch.get("noise_i"), B)
        fe_d = self.dev(fe)
        z = lambda: torch.zeros((B, L.NBINS), dtype=torch.float64, device=self.device)
        sums = torch.zeros((B, 3), dtype=torch.float64, device=self.device)
        sqe, sqi, E, I = z(), z(), z(), z()
        self._sync_stream()
        rc = self.lib.tsff_array_loss(self.h, self._ptr(X), self._ptr(fe_d), self._ptr(ed), self._ptr(idt), self._ptr(ea),
                                      self._ptr(ia), self._ptr(ne_), self._ptr(ni_), B, self._ptr(sums), self._ptr(sqe),
                                      self._ptr(sqi), self._ptr(E), self._ptr(I))
        L.check(self.lib, self.h, rc)
        return sums, sqe, sqi, E, I

    def set_denominator_mode(self, mode: int):
        """0: constant denominators (fit loss); 2: |data| + 1e-10 per sample (the reference's Hessian loss)."""
        L.check(self.lib, self.h, self.lib.tsff_set_option(self.h, L.OPT_DENOM_MODE, int(mode)))

    def set_launch_plan(self, plan: int):
        """Bit mask (TSFF_OPT_LAUNCH_PLAN).  0: automatic (one 256-thread workgroup per (lineout, feature) when two fit a
        CU; loss + gradient by the one-sweep kernel where its restrictions hold); bit 0: never interleave the features;
        bit 1: always the two-sweep kernel; bit 2: never three forward-only workgroups per CU."""
        L.check(self.lib, self.h, self.lib.tsff_set_option(self.h, L.OPT_LAUNCH_PLAN, int(plan)))

    def set_dlm_blocks(self, n: int):
        """TSFF_OPT_DLM_BLOCKS: column blocks of the pipelined DLM step (per-lineout tables of block i + 1 on a second stream while
        the one-sweep kernel works on block i).  0 / 1: off (default: measured slower on gfx950), n: n blocks.  Same bits."""
        L.check(self.lib, self.h, self.lib.tsff_set_option(self.h, L.OPT_DLM_BLOCKS, int(n)))

    def fp64_fma_peak_tflops(self) -> float:
        """Measured FP64 vector FMA rate of this device (micro-benchmark, TFLOP/s)."""
        v = C.c_double()
        L.check(self.lib, self.h, self.lib.tsff_fp64_fma_peak(self.h, C.byref(v)))
        return float(v.value)

    def l1_read_peak_tbps(self) -> float:
        """Measured vector-L1 read rate of this device (micro-benchmark, TB/s): the roof of the 2-D sampler for tables read
        through L1/L2."""
        v = C.c_double()
        L.check(self.lib, self.h, self.lib.tsff_l1_read_peak(self.h, C.byref(v)))
        return float(v.value)

    def fp64_mfma_peak_tflops(self) -> float:
        """Measured FP64 matrix-core (v_mfma_f64_16x16x4_f64) rate of this device (micro-benchmark, TFLOP/s)."""
        v = C.c_double()
        L.check(self.lib, self.h, self.lib.tsff_fp64_mfma_peak(self.h, C.byref(v)))
        return float(v.value)

    def enable_timing(self, ring: int = 256):
        """Record one HIP event pair around every main-kernel launch (ring of ``ring`` launches)."""
        L.check(self.lib, self.h, self.lib.tsff_enable_timing(self.h, int(ring)))

    def kernel_times_ms(self, max_n: int = 65536) -> np.ndarray:
        buf = (C.c_float * max_n)()
        n = C.c_int32()
        L.check(self.lib, self.h, self.lib.tsff_kernel_times(self.h, buf, max_n, C.byref(n)))
        return np.array(buf[: n.value], dtype=np.float64)
