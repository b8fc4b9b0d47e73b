"""Input decks used by the tests, written as Python dicts.

The numbers are those of the reference's test decks (tests/configs/1d-defaults.yaml merged with
tests/configs/1d-inputs.yaml, and tests/configs/epw_*.yaml) -- they are inputs, i.e. fixture data;
the structure is the nested dict the reference's runner produces after its flatten/update/unflatten
merge (tsadar/runner.py:70-72) plus the three derived keys every entry point adds
(tests/test_forward/test_1d.py:42-51).
"""
from __future__ import annotations

import copy


def _p(val, active, lb, ub, **kw):
    d = dict(val=val, active=active, lb=lb, ub=ub)
    d.update(kw)
    return d


def deck_1d(points_per_pixel=5, nvx=128, m=2.5, fe_active=True):
    """1-D EPW-only forward deck of tests/test_forward/test_1d.py."""
    cfg = {
        "parameters": {
            "electron": {
                "Te": _p(0.5, True, 0.001, 1.5),
                "ne": _p(0.2, True, 0.001, 1.0),
                "fe": {
                    "active": fe_active,
                    "type": "dlm",
                    "dim": 1,
                    "nvx": nvx,
                    "params": {"m": {"val": m, "lb": 2.0, "ub": 5.0}},
                },
            },
            "ion-1": {
                "Ti": _p(0.2, False, 0.01, 3.0, same=False),
                "Z": _p(8.0, False, 1.0, 25.0),
                "A": {"val": 40.0, "active": False},
                "fract": {"val": 1.0, "active": False},
            },
            "general": {
                "amp1": _p(1.0, True, 0.01, 3.75),
                "amp2": _p(1.0, True, 0.01, 3.75),
                "amp3": _p(1.0, False, 0.0, 10.0),
                "lam": _p(524.0, True, 523.0, 528.0),
                "Te_gradient": _p(0.0, False, 0.0, 10.0, num_grad_points=1),
                "ne_gradient": _p(0.0, False, 0.0, 15.0, num_grad_points=1),
                "ud": _p(0.0, False, -10.0, 10.0, angle=0.0),
                "Va": _p(0.0, False, -20.5, 20.5, angle=0.0),
            },
        },
        "other": {
            "extraoptions": {
                "spectype": "1d",
                "load_ion_spec": False,
                "load_ele_spec": True,
                "fit_IAW": False,
                "fit_EPWb": True,
                "fit_EPWr": True,
            },
            "PhysParams": {
                "background": [0, 0],
                "norm": 0,
                "widIRF": {"spect_stddev_ele": 1.3, "spect_stddev_ion": 0.015},
            },
            "iawoff": 0,
            "iawfilter": [1, 4, 24, 528],
            "CCDsize": [1024, 1024],
            "points_per_pixel": points_per_pixel,
        },
        "data": {
            "fit_rng": {
                "blue_min": 450,
                "blue_max": 510,
                "red_min": 540,
                "red_max": 625,
                "iaw_min": 525.5,
                "iaw_max": 527.5,
                "iaw_cf_min": 526.49,
                "iaw_cf_max": 526.51,
                "forward_epw_start": 400,
                "forward_epw_end": 700,
                "forward_iaw_start": 525.75,
                "forward_iaw_end": 527.25,
            },
            "ion_loss_scale": 1.0,
            "ele_lam_shift": 0.0,
            "probe_beam": "P9",
            "shotnum": 101675,
        },
        "optimizer": {
            "method": "l-bfgs-b",
            "loss_method": "l2",
            "y_norm": True,
            "x_norm": False,
            "grad_method": "AD",
            "batch_size": 2,
            "num_epochs": 120,
        },
        "nn": {"use": False},
    }
    return finish(cfg)


def finish(cfg):
    """The three derived keys (tests/test_forward/test_1d.py:42-51)."""
    r = cfg["data"]["fit_rng"]
    cfg["other"]["lamrangE"] = [r["forward_epw_start"], r["forward_epw_end"]]
    cfg["other"]["lamrangI"] = [r["forward_iaw_start"], r["forward_iaw_end"]]
    cfg["other"]["npts"] = int(cfg["other"]["CCDsize"][1] * cfg["other"]["points_per_pixel"])
    return cfg


def deck_fit(points_per_pixel=1, nvx=128, m=2.0, active=("Te", "ne", "Ti", "Va", "lam", "amp1"), n_ion=1):
    """EPW+IAW fit deck used for the BASELINE configs: Maxwellian f_e
    (DLM m=2, not fitted), both features loaded, blue+red+IAW fit ranges, l2 loss."""
    cfg = copy.deepcopy(deck_1d(points_per_pixel, nvx, m, fe_active=("m" in active)))
    ext = cfg["other"]["extraoptions"]
    ext["load_ion_spec"] = True
    ext["fit_IAW"] = True
    P = cfg["parameters"]
    P["electron"]["Te"] = _p(0.6, "Te" in active, 0.01, 1.5)
    P["electron"]["ne"] = _p(0.2, "ne" in active, 0.001, 1.0)
    P["ion-1"]["Ti"] = _p(0.2, "Ti" in active, 0.01, 1.0, same=False)
    P["ion-1"]["Z"] = _p(8.0, "Z" in active, 1.0, 25.0)
    g = P["general"]
    g["amp1"] = _p(1.0, "amp1" in active, 0.01, 3.75)
    g["amp2"] = _p(1.0, "amp2" in active, 0.01, 3.75)
    g["amp3"] = _p(1.0, "amp3" in active, 0.01, 3.75)
    g["lam"] = _p(526.5, "lam" in active, 523.0, 528.0)
    g["Va"] = _p(0.0, "Va" in active, -20.5, 20.5, angle=0.0)
    g["ud"] = _p(0.0, "ud" in active, -10.0, 10.0, angle=0.0)
    g["Te_gradient"]["active"] = "Te_gradient" in active
    g["ne_gradient"]["active"] = "ne_gradient" in active
    if n_ion == 2:
        P["ion-1"]["fract"]["val"] = 0.6
        P["ion-2"] = {
            "Ti": _p(0.3, "Ti" in active, 0.01, 1.0, same=False),
            "Z": _p(1.0, False, 0.5, 18.0),
            "A": {"val": 1.0, "active": False},
            "fract": {"val": 0.4, "active": False},
        }
    return finish(cfg)


def deck_kat(kind):
    """Decks of the two dispersion-relation known-answer tests (tests/configs/epw_inputs.yaml and
    the IAW variant used by tests/test_form_factor/test_iaw.py)."""
    cfg = copy.deepcopy(deck_1d(1, 128, 2.0, fe_active=False))
    P = cfg["parameters"]
    P["electron"]["Te"] = _p(0.6, False, 0.01, 1.5)
    P["electron"]["ne"] = _p(0.2, False, 0.001, 1.0)
    P["ion-1"]["Ti"] = _p(0.2, True, 0.01, 1.0, same=False)
    P["ion-1"]["Z"] = _p(1.0, True, 0.5, 18.0)
    P["ion-1"]["A"] = {"val": 1.0, "active": False}
    P["general"]["lam"] = _p(526.5, False, 523.0, 528.0)
    return finish(cfg)


def deck_angular(dim=1, nvx=64, ccd=(1024, 1024), start=90, end=950):
    """ARTS deck: the numbers of tests/configs/arts1v_test_inputs.yaml / arts2v_test_defaults.yaml (Te 1.0, ne 0.4,
    Z 8, A 14, lam 526.5, DLM m 2.5; lineouts 90:950, spect_FWHM_ele 0.9 nm, ang_FWHM_ele 1 deg), spectype
    "angular_full" as tests/test_forward/test_angular_1v.py:58 sets it."""
    cfg = copy.deepcopy(deck_1d(1, nvx, 2.5, fe_active=True))
    P = cfg["parameters"]
    P["electron"]["Te"] = _p(1.0, True, 0.01, 1.5)
    P["electron"]["ne"] = _p(0.4, True, 0.001, 1.0)
    P["ion-1"]["A"]["val"] = 14.0
    P["general"]["lam"] = _p(526.5, True, 525.0, 528.0)
    if dim == 2:
        P["electron"]["fe"] = {"active": True, "type": "arbitrary", "dim": 2, "nvx": nvx,
                               "params": {"init_m": 2.5, "learn_log": True}}
        P["general"]["ud"]["angle"] = 20.0
        P["general"]["Va"]["angle"] = -35.0
    cfg["other"]["extraoptions"]["spectype"] = "angular_full"
    cfg["other"]["PhysParams"]["widIRF"].update(spect_FWHM_ele=0.9, ang_FWHM_ele=1.0, spect_stddev_ele=0.9 / 2.3548)
    cfg["other"]["CCDsize"] = list(ccd)
    cfg["data"]["lineouts"] = {"type": "range", "start": start, "end": end, "skip": 20}
    cfg = finish(cfg)
    cfg["other"]["npts"] = 1024  # prepare.py:202 evaluates it before CCDsize is replaced by the reduced shape
    return cfg
