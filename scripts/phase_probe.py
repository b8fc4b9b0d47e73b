"""How much of k_spectrum is the IRF convolution (forward + adjoint)?  Times loss_grad for the baseline deck and for the
same deck with IRF widths shrunk to one tap."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsadar_amd import synthetic as S
from tsadar_amd.engine import Engine
from tsadar_amd.calibration import sa_lookup

B = 4096
for label, se, si in (("baseline (107 + 247 taps)", 1.3, 0.015), ("narrow IRF", 0.15, 0.00075)):
    cfg = S.baseline_deck(batch_size=B)
    cfg["other"]["PhysParams"]["widIRF"] = {"spect_stddev_ele": se, "spect_stddev_ion": si}
    sa = sa_lookup("P9"); sa = dict(sa=sa["sa"], weights=sa["weights"] * np.ones([B, 10]))
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
        print(label, mode, "kernel ms", float(np.mean(eng.kernel_times_ms())), "taps", int(eng._cfg_struct.n_taps_ele), int(eng._cfg_struct.n_taps_ion))
