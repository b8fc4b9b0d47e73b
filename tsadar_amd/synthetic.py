"""Synthetic Maxwellian workloads of BASELINE.json: the input deck of the
benchmark configs and seeded random plasma conditions.  No network, no shot data: every lineout is
drawn from the parameter ranges of the reference's own round-trip test
(tests/test_inverse/test_1d_random.py:33-39) and decks (tests/configs/1d-defaults.yaml)."""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .params import SlotMap, ThomsonParams

SEED = 20251004
ACTIVE = ("Te", "ne", "Ti", "Va", "lam", "amp1")  # P = 6 free parameters per lineout


def _p(val, active, lb, ub, **kw):
    d = dict(val=val, active=active, lb=lb, ub=ub)
    d.update(kw)
    return d


def baseline_deck(points_per_pixel: int = 1, nvx: int = 128, active=ACTIVE, batch_size: int = 4096) -> dict:
    """EPW [400, 700] nm + IAW [525.75, 527.25] nm, 1024*ppp wavelength points each, P9 geometry,
    sigma_E = 1.3 nm, sigma_I = 0.015 nm, iawfilter [1, 4, 24, 528], l2 loss with y_norm, Maxwellian
    f_e (DLM m = 2, not fitted), one ion species (Z = 8, A = 40)."""
    a = set(active)
    cfg = {
        "parameters": {
            "electron": {
                "Te": _p(0.6, "Te" in a, 0.01, 1.5),
                "ne": _p(0.2, "ne" in a, 0.001, 1.0),
                "fe": {"active": "m" in a, "type": "dlm", "dim": 1, "nvx": nvx,
                       "params": {"m": {"val": 2.0, "lb": 2.0, "ub": 5.0}}},
            },
            "ion-1": {
                "Ti": _p(0.2, "Ti" in a, 0.01, 1.0, same=False),
                "Z": _p(8.0, "Z" in a, 1.0, 25.0),
                "A": {"val": 40.0, "active": False},
                "fract": {"val": 1.0, "active": False},
            },
            "general": {
                "amp1": _p(1.0, "amp1" in a, 0.01, 3.75),
                "amp2": _p(1.0, "amp2" in a, 0.01, 3.75),
                "amp3": _p(1.0, "amp3" in a, 0.01, 3.75),
                "lam": _p(526.5, "lam" in a, 523.0, 528.0),
                "Te_gradient": _p(0.0, False, 0.0, 10.0, num_grad_points=1),
                "ne_gradient": _p(0.0, False, 0.0, 15.0, num_grad_points=1),
                "ud": _p(0.0, "ud" in a, -10.0, 10.0, angle=0.0),
                "Va": _p(0.0, "Va" in a, -20.5, 20.5, angle=0.0),
            },
        },
        "other": {
            "extraoptions": {"spectype": "1d", "load_ion_spec": True, "load_ele_spec": True,
                             "fit_IAW": True, "fit_EPWb": True, "fit_EPWr": True},
            "PhysParams": {"background": [0, 0], "norm": 0,
                           "widIRF": {"spect_stddev_ele": 1.3, "spect_stddev_ion": 0.015}},
            "iawoff": 0,
            "iawfilter": [1, 4, 24, 528],
            "CCDsize": [1024, 1024],
            "points_per_pixel": points_per_pixel,
            "lamrangE": [400, 700],
            "lamrangI": [525.75, 527.25],
            "npts": 1024 * points_per_pixel,
        },
        "data": {
            "fit_rng": {"blue_min": 450, "blue_max": 510, "red_min": 540, "red_max": 625,
                        "iaw_min": 525.5, "iaw_max": 527.5, "iaw_cf_min": 526.49, "iaw_cf_max": 526.51,
                        "forward_epw_start": 400, "forward_epw_end": 700,
                        "forward_iaw_start": 525.75, "forward_iaw_end": 527.25},
            "ion_loss_scale": 1.0, "ele_lam_shift": 0.0, "probe_beam": "P9", "shotnum": 0,
        },
        "optimizer": {"method": "l-bfgs-b", "loss_method": "l2", "y_norm": True, "x_norm": False,
                      "grad_method": "AD", "batch_size": batch_size, "num_epochs": 120},
        "nn": {"use": False},
    }
    return cfg


RANGES = dict(Te=(0.3, 1.5), ne=(0.1, 0.7), Ti=(0.05, 0.5), lam=(525.5, 527.5), amp1=(0.5, 2.5),
              amp2=(0.5, 2.5), amp3=(0.5, 2.5), Va=(-2.0, 2.0))
_SLOT = dict(Te=L.P_TE, ne=L.P_NE, Ti=L.P_ION0 + L.ION_TI, lam=L.P_LAM, amp1=L.P_AMP1, amp2=L.P_AMP2,
             amp3=L.P_AMP3, Va=L.P_VA)


def draw_params(cfg: dict, B: int, rng: np.random.Generator, activate: bool = True, dlm: bool = False) -> ThomsonParams:
    """B lineouts with physical values uniform in RANGES (mapped through the exact inverse of the
    activation, so the physical value is the drawn one)."""
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=activate)
    sm: SlotMap = tp.slots
    for name, (lo, hi) in RANGES.items():
        s = _SLOT[name]
        u = (rng.uniform(lo, hi, B) - sm.shift[s]) / sm.scale[s]
        tp.X[:, s] = np.log(u / (1 - u)) if sm.sigmoid[s] else u
    if dlm:  # per-lineout super-Gaussian order (tests/test_inverse/test_1d_random.py:33)
        u = (rng.uniform(2.0, 3.5, B) - sm.shift[L.P_M]) / sm.scale[L.P_M]
        u = np.clip(u, 1e-6, 1 - 1e-6)
        tp.X[:, L.P_M] = np.log(u / (1 - u)) if sm.sigmoid[L.P_M] else u
    return tp


def make_batch(engine, truth: ThomsonParams, rng: np.random.Generator, noise_level: float = 0.01, fe=None) -> dict:
    """Synthetic 'measured' spectra resident on the GPU: the engine's own forward model at the truth
    parameters with unit amplitudes, times (1 + 1 % Gaussian noise); amplitudes = row maximum inside
    the fit ranges (the reference's lineouts.py:127-150).  noise_e / noise_i are None (= 0)."""
    import torch

    B = truth.X.shape[0]
    ones = np.ones(B)
    E, I = engine.forward(truth.to_matrix(), ones, ones, fe=fe)
    gen = torch.Generator(device=E.device)
    gen.manual_seed(int(rng.integers(1 << 31)))
    E = E * (1 + noise_level * torch.randn(E.shape, dtype=E.dtype, device=E.device, generator=gen))
    I = I * (1 + noise_level * torch.randn(I.shape, dtype=I.dtype, device=I.device, generator=gen))
    mE = torch.from_numpy(engine.mask_ele != 0).to(E.device)
    mI = torch.from_numpy(engine.mask_ion != 0).to(E.device)
    e_amps = E[:, mE].amax(dim=1) if bool(mE.any()) else torch.ones(B, dtype=E.dtype, device=E.device)
    i_amps = I[:, mI].amax(dim=1) if bool(mI.any()) else torch.ones(B, dtype=E.dtype, device=E.device)
    return dict(e_data=E.contiguous(), i_data=I.contiguous(), e_amps=e_amps.contiguous(), i_amps=i_amps.contiguous(),
                noise_e=None, noise_i=None)
