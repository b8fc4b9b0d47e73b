"""The compute half of ``tsadar.forward.calc_series.forward_pass`` (reference forward/calc_series.py:17-100): derive the
wavelength windows and npts from the deck, look up the scattering geometry, build the dummy batch, evaluate the
diagnostic for every entry of the series and time it.  Plotting, xarray export and MLflow logging (calc_series.py:100-
200) stay in the reference and consume the arrays returned here."""
from __future__ import annotations

from time import time
from typing import Dict

import numpy as np

from .calibration import angular_pixel_axis, get_scattering_angles
from .diagnostic import ThomsonScatteringDiagnostic
from .params import ThomsonParams


def forward_spectra(config: Dict) -> Dict:
    """-> {"ThryE", "ThryI", "lamAxisE", "lamAxisI" (arrays with a leading series axis), "spectrum_calc_time", "ts_params"}."""
    is_angular = "angular" in config["other"]["extraoptions"]["spectype"]
    config["optimizer"]["batch_size"] = 1
    r = config["data"]["fit_rng"]
    config["other"]["lamrangE"] = [r["forward_epw_start"], r["forward_epw_end"]]
    config["other"]["lamrangI"] = [r["forward_iaw_start"], r["forward_iaw_end"]]
    config["other"]["npts"] = int(config["other"]["CCDsize"][1] * config["other"]["points_per_pixel"])
    sas = get_scattering_angles(config)
    dummy_batch = {"i_data": np.array([1]), "e_data": np.array([1]), "noise_e": np.array([0]), "noise_i": np.array([0]),
                   "e_amps": np.array([1]), "i_amps": np.array([1])}
    if is_angular:
        config["other"]["extraoptions"]["spectype"] = "angular_full"
        sas["angAxis"] = angular_pixel_axis()  # get_calibrations(104000, "angular", ...)[0] (calibration.py:456-458)
        shape = (config["other"]["CCDsize"][0], config["other"]["CCDsize"][1])
        dummy_batch["i_data"] = np.ones(shape)
        dummy_batch["e_data"] = np.ones(shape)
    serieslen = len(config["series"]["vals1"]) if "series" in config else 1
    out = {k: [None] * serieslen for k in ("ThryE", "ThryI", "lamAxisE", "lamAxisI")}
    ts_diag = ThomsonScatteringDiagnostic(config, scattering_angles=sas)  # (one engine for the whole series)
    t_start = time()
    ts_params = None
    for i in range(serieslen):
        ts_params = ThomsonParams(config["parameters"], num_params=1, batch=not is_angular)
        E, I, lE, lI = ts_diag(ts_params, dummy_batch)
        out["ThryE"][i], out["ThryI"][i], out["lamAxisE"][i], out["lamAxisI"][i] = E, I, lE, lI
    spectime = time() - t_start
    res = {k: np.array(v) for k, v in out.items() if k != "lamAxisI" or not is_angular}
    if is_angular:
        res["lamAxisI"] = out["lamAxisI"]
    res["spectrum_calc_time"] = spectime
    res["ts_params"] = ts_params
    return res
