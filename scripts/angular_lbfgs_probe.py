"""HIP-driven L-BFGS-B on the angular (ARTS, 1-D DLM) loss from the start of test_angular_vg_loss_finite_difference: the iterate
sequence, to be compared with the oracle-driven one (tests/golden/make_angular_lbfgs.py)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import decks
from oracle import tsadar_oracle as orc
from scipy.optimize import minimize
from tsadar_amd import ThomsonParams, tree, calibration
from tsadar_amd.loss_function import LossFunction
cfg = decks.deck_angular(1, 64, (128, 256), 10, 110)
for k in ("amp1", "amp2", "lam"):
    cfg["parameters"]["general"][k]["active"] = False
cfg["other"]["extraoptions"]["spectype"] = "angular"
sa = calibration.get_scattering_angles(cfg)
cfg["other"]["extraoptions"]["spectype"] = "angular_full"
sa["angAxis"] = calibration.angular_pixel_axis()
vx = orc.velocity_grid(64)
def oracle_image(normed, e_amps):
    phys = orc.physical_params(cfg["parameters"], normed, True)
    p = orc.lineout_params(phys, 0, 1)
    Po, lam_cm = orc.form_factor(cfg["other"]["lamrangE"], 1024, 0.0, sa["sa"], 1, p, vx, orc.dlm_fe(float(p["m"]), 64))
    return orc.ats_spectrum(cfg, sa["weights"], sa["angAxis"], Po, np.squeeze(lam_cm) * 1e7, 256, e_amps, p)
cfg["parameters"]["electron"]["fe"]["active"] = False
truth2 = orc.init_normed_params(cfg["parameters"], 1, True)
truth2["Te"] = truth2["Te"] - 0.2
truth2["ne"] = truth2["ne"] + 0.12
data = oracle_image(truth2, np.ones((100, 1)))[0]
batch2 = dict(e_data=data, i_data=np.zeros((100, 256)), e_amps=np.ones((100, 1)), i_amps=np.zeros(100), noise_e=np.array([0.0]), noise_i=np.array([0.0]))
fit_fn = LossFunction(cfg, sa, batch2)
tp2 = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
diff2, static2 = tree.partition(tp2, tree.get_filter_spec(cfg["parameters"], tp2))
x2, fit_fn.unravel_weights = tree.ravel_pytree(diff2)
its = []
def vg(x, *a):
    v, g = fit_fn.vg_loss(x, *a)
    its.append(np.concatenate([x, [v], g]))
    return v, g
res = minimize(vg, x2, args=(static2, batch2), method="L-BFGS-B", jac=True, options={"maxiter": 60, "ftol": 1e-15, "gtol": 1e-12})
its = np.array(its)
np.save(os.path.join(ROOT, "gpurun_out", "ang_hip_its.npy"), its)
print(res.nit, res.nfev, res.fun, res.fun / its[0, 2], res.x, res.message)
ref = np.load(os.path.join(ROOT, "gpurun_out", "ang_oracle_its.npy"))
n = min(len(ref), len(its))
for k in range(n):
    print(k, "x diff", np.max(np.abs(its[k, :2] - ref[k, :2])), "f rel", abs(its[k, 2] - ref[k, 2]) / ref[k, 2], "g diff rel", np.max(np.abs(its[k, 3:] - ref[k, 3:])) / np.max(np.abs(ref[k, 3:])))
