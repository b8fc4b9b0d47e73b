"""Drop-in for ``tsadar.core.thomson_diagnostic.ThomsonScatteringDiagnostic`` (reference
core/thomson_diagnostic.py:10-142) backed by the HIP engine: same constructor, same call
signature, same return tuple, same NotImplementedError for unknown spectypes."""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import _lib as L
from .engine import Engine
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
            raise NotImplementedError("angular spectra (2-D f_e / ARTS) are outside the MI355X 1-D form-factor path")
        else:
            raise NotImplementedError(f"Unknown spectype: {spectype}")  # thomson_diagnostic.py:40
        self._engines = {}

    def engine(self, activate: bool) -> Engine:
        """One engine per activation mode (the sigmoid flags are static configuration)."""
        if activate not in self._engines:
            self._engines[activate] = Engine(self.cfg, self.scattering_angles, activate=activate,
                                             irf_cutoff_sigmas=self.irf_cutoff_sigmas)
        return self._engines[activate]

    def __call__(self, ts_params: ThomsonParams, batch: Dict):
        """-> (ThryE, ThryI, lamAxisE, lamAxisI), each [B, 1024] float64 NumPy
        (thomson_diagnostic.py:109-142).  A feature that is not loaded comes back as zeros."""
        eng = self.engine(ts_params.activate)
        X = ts_params.to_matrix()
        B = X.shape[0]
        E, I = eng.forward(X, batch["e_amps"], batch["i_amps"], batch.get("noise_e"), batch.get("noise_i"))
        lamE = np.tile(eng.lamAxisE[None, :], (B, 1))
        lamI = np.tile(eng.lamAxisI[None, :], (B, 1))
        return E.cpu().numpy(), I.cpu().numpy(), lamE, lamI

    def spectrum_breakdown(self, ts_params, batch):
        raise NotImplementedError("spectrum_breakdown (post-processing plots) is outside the hot path")
