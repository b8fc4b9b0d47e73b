"""Fixed (angle-independent) cost of k_spectrum: time with 10, 5 and 1 scattering angles, narrow IRF."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsadar_amd import synthetic as S
from tsadar_amd.engine import Engine
from tsadar_amd.calibration import sa_lookup

B = 4096
for na in (10, 5, 1):
    cfg = S.baseline_deck(batch_size=B)
    cfg["other"]["PhysParams"]["widIRF"] = {"spect_stddev_ele": 0.15, "spect_stddev_ion": 0.00075}
    sa = sa_lookup("P9"); sa = dict(sa=sa["sa"][:na], weights=(sa["weights"] * np.ones([B, 10]))[:, :na])
    eng = Engine(cfg, sa)
    rng = np.random.default_rng(1)
    truth = S.draw_params(cfg, B, rng); batch = S.make_batch(eng, truth, rng); guess = S.draw_params(cfg, B, rng)
    X = eng.dev(guess.to_matrix()); gm = guess.grad_mask()
    w = eng.loss_weights(B, 1.0, 1.0, 1.0)
    for mode in ("fwd+grad", "fwd"):
        f = (lambda: eng.loss_grad(X, batch, w, gm)) if mode == "fwd+grad" else (lambda: eng.forward(X, batch["e_amps"], batch["i_amps"]))
        for _ in range(3): f()
        torch.cuda.synchronize(); eng.enable_timing(20)
        for _ in range(20): f()
        torch.cuda.synchronize()
        print("angles", na, mode, "kernel ms", round(float(np.mean(eng.kernel_times_ms())), 4))
