"""Drop-in for ``tsadar.core.thomson_diagnostic.ThomsonScatteringDiagnostic`` (reference
core/thomson_diagnostic.py:10-142) backed by the HIP engine: same constructor, same call
signature, same return tuple, same NotImplementedError for unknown spectypes."""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import _lib as L
from .engine import Engine, wavelength_axis_nm
from .params import ThomsonParams


class ThomsonScatteringDiagnostic:
    def __init__(self, cfg: Dict, scattering_angles: Dict, irf_cutoff_sigmas: float = 12.0):
        self.cfg = cfg
        self.scattering_angles = scattering_angles
        self.irf_cutoff_sigmas = irf_cutoff_sigmas
        spectype = cfg["other"]["extraoptions"]["spectype"]
        if "temporal" in spectype or "imaging" in spectype or "1d" in spectype:
            pass
        elif "angular" in spectype:
            if spectype != "angular_full":
                raise NotImplementedError("only the angular_full ARTS model is built on the MI355X path")
            if "angAxis" not in scattering_angles:
                raise KeyError("scattering_angles['angAxis'] (prepare.py:136) is required for angular spectra")
        else:
            raise NotImplementedError(f"Unknown spectype: {spectype}")  # thomson_diagnostic.py:40
        self._engines = {}
        # (world, rank, process group) when the (lambda, theta) point list of a 2-D angular deck is sharded over the GPUs
        # of a node (parallel_calc_all_chi_vals, form_factor.py:431-447); set by LossFunction(distributed=True)
        self.dist = None

    def engine(self, activate: bool) -> Engine:
        """One engine per activation mode (the sigmoid flags are static configuration)."""
        if activate not in self._engines:
            angular = "angular" in self.cfg["other"]["extraoptions"]["spectype"]
            self._engines[activate] = Engine(self.cfg, self.scattering_angles, activate=activate,
                                             irf_cutoff_sigmas=self.irf_cutoff_sigmas,
                                             fe_mode=L.FE_PER_LINEOUT if angular else None)
        return self._engines[activate]

    def __call__(self, ts_params: ThomsonParams, batch: Dict):
        """-> (ThryE, ThryI, lamAxisE, lamAxisI), each [B, 1024] float64 NumPy
        (thomson_diagnostic.py:109-142).  A feature that is not loaded comes back as zeros."""
        eng = self.engine(ts_params.activate)
        X = ts_params.to_matrix()
        B = X.shape[0]
        if self.cfg["other"]["extraoptions"]["spectype"] == "angular_full":
            return self._angular(eng, ts_params, batch)
        E, I = eng.forward(X, batch["e_amps"], batch["i_amps"], batch.get("noise_e"), batch.get("noise_i"))
        lamE = np.tile(eng.lamAxisE[None, :], (B, 1))
        lamI = np.tile(eng.lamAxisI[None, :], (B, 1))
        return E.cpu().numpy(), I.cpu().numpy(), lamE, lamI

    def _angular(self, eng: Engine, ts_params: ThomsonParams, batch: Dict, to_host: bool = True):
        """spectype "angular_full": one plasma condition -> the ARTS image ThryE [rows, n_lam]
        (FitModel.electron_spectrum matmul branch, add_ATS_IRF, reduce_ATS_to_resunit).  The ion feature is not
        measured by ARTS: ThryI = 0 + noise_i like the reference's modlI = 0."""
        cfg, sas = self.cfg, self.scattering_angles
        if ts_params.X.shape[0] != 1:
            raise NotImplementedError("angular spectra are computed for one plasma condition (no vmap in the reference)")
        if cfg["other"]["PhysParams"]["norm"] > 0:
            raise NotImplementedError("PhysParams.norm > 0 is not built for angular spectra")
        e_data = np.asarray(batch["e_data"])
        lam_step = round(eng.npts / e_data.shape[1])
        n_px = np.asarray(sas["weights"]).shape[0]
        ang_step = round(n_px / cfg["other"]["CCDsize"][0])
        key = (lam_step, ang_step)
        if getattr(eng, "_ats_key", None) != key:
            wid = cfg["other"]["PhysParams"]["widIRF"]
            eng.ats_setup(sas["weights"], sas["angAxis"], wid["spect_FWHM_ele"] / 2.3548, wid["ang_FWHM_ele"] / 2.3548,
                          lam_step, ang_step, cfg["data"]["lineouts"]["start"],
                          min(cfg["data"]["lineouts"]["end"], n_px // ang_step), self.irf_cutoff_sigmas)
            eng._ats_key = key
        phys = ts_params.physical_matrix()
        fe2 = fe1 = None
        if eng.fe_dim == 2:
            gen = cfg["parameters"]["general"]
            fe2 = eng.dev(np.ascontiguousarray(ts_params()["electron"]["fe"], dtype=np.float64))
            if self.dist is not None and self.dist[0] > 1:
                from . import distributed as D

                P = D.form_factor_2d_sharded(eng, 0, phys, fe2, gen["ud"]["angle"], gen["Va"]["angle"], *self.dist, save=not to_host)
            else:   # (fit loop, to_host False: keep the projections for the adjoint that follows)
                P = eng.form_factor_2d(0, phys, fe2, gen["ud"]["angle"], gen["Va"]["angle"], save=not to_host)
        else:
            fe1 = np.ascontiguousarray(np.asarray(ts_params()["electron"]["fe"], dtype=np.float64).reshape(1, -1))
            P = eng.form_factor(0, phys, fe1)
        p = phys[0]
        rows = eng._ats_shape[0]
        e_amps = np.broadcast_to(np.asarray(batch["e_amps"], dtype=np.float64).reshape(-1, 1), (rows, 1))
        E_dev = eng.ats_spectrum(P[0], e_amps, p[L.P_LAM], p[L.P_AMP1], p[L.P_AMP2])
        lamE = np.mean(wavelength_axis_nm(cfg["other"]["lamrangE"], eng.npts).reshape(-1, lam_step), axis=1)
        # what the adjoint (LossFunction._vg_angular) needs again: the device-resident P and image, the table, the parameters
        self._angular_ctx = dict(P=P, phys=phys, fe2=fe2, fe1=fe1, e_amps=e_amps, E_dev=E_dev, lamE=lamE)
        if not to_host:   # (the fit loop keeps the image on the device: the loss and its seed are evaluated there)
            return None, None, lamE, []
        E = E_dev.cpu().numpy() + np.asarray(batch["noise_e"])
        return E, 0 + np.asarray(batch["noise_i"]), lamE, []

    def spectrum_breakdown(self, ts_params, batch):
        raise NotImplementedError("spectrum_breakdown (post-processing plots) is outside the hot path")
