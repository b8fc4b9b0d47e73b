"""Wall time of one LossFunction.vg_loss evaluation (value + gradient of every trainable leaf) for ARTS decks at the
reference's full size (1024 x 1024 CCD, 860 lineout rows, 241 angles): 1-D DLM f_e (nvx 256) and 2-D Arbitrary2V f_e
(nvx 128: 16384 trainable values of the distribution function + the plasma parameters)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import decks
from tsadar_amd import ThomsonParams, calibration, tree
from tsadar_amd.loss_function import LossFunction

for dim, nvx in ((1, 256), (2, 128)):
    cfg = decks.deck_angular(dim, nvx)
    cfg["other"]["extraoptions"]["spectype"] = "angular"
    sa = calibration.get_scattering_angles(cfg)
    cfg["other"]["extraoptions"]["spectype"] = "angular_full"
    sa["angAxis"] = calibration.angular_pixel_axis()
    batch = dict(e_data=np.ones((860, 1024)), i_data=np.zeros((860, 1024)), e_amps=np.ones((860, 1)), i_amps=np.zeros(860),
                 noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    lf = LossFunction(cfg, sa, batch)
    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    truth = tp.copy(); truth.X[0, 0] -= 0.3
    batch["e_data"] = lf.ts_diag(truth, batch)[0]
    lf = LossFunction(cfg, sa, batch)
    diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
    x0, lf.unravel_weights = tree.ravel_pytree(diff)
    lf.vg_loss(x0, static, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        v, g = lf.vg_loss(x0, static, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        lf.ts_diag(tp, batch)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / n
    print(f"dim {dim} nvx {nvx}: {x0.size} trainable values; forward image {df*1e3:.1f} ms; value+gradient {dt*1e3:.1f} ms "
          f"(loss {v:.4e}, |g| {np.linalg.norm(g):.3e})", flush=True)
