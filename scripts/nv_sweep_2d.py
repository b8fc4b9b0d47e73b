"""The L1/L2 sampler against the oracle at several table sizes (odd and even; the loop is unrolled by two), each several times:
prints the largest relative error per run.  usage: python scripts/nv_sweep_2d.py [nv ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import decks, util
import test_gpu_parity as T
from oracle import tsadar_oracle as orc

cfg = decks.deck_fit()
sa = dict(sa=np.array([35.0, 60.0, 110.0]), weights=np.ones((1, 3)) / 3)
eng = T._engine(cfg, sa)
normed = util.random_lineouts(cfg, 1, seed=61, ranges=dict(ud=(-1.5, 1.5)))
phys = orc.physical_params(cfg["parameters"], normed, True)
phys["ud"] = np.array([0.8])
X = util.normed_to_matrix(phys, 1)
idx = np.array([0, 400, 1023])
for nv in [int(a) for a in sys.argv[1:]] or [132, 133, 134, 135, 137, 160, 161, 255, 256]:
    vx, fe2 = T._fe2d(nv, "anisotropic")
    Po, _ = orc.form_factor_2d(cfg["other"]["lamrangE"], 1024, 0.0, sa["sa"], 1, orc.lineout_params(phys, 0, 1), vx, fe2, 25.0, -40.0, lam_index=idx)
    errs = []
    for rep in range(6):
        P = eng.form_factor_2d(0, X, fe2, 25.0, -40.0).cpu().numpy()
        e = np.abs(P[0][:, idx, :] - Po) / np.abs(Po)
        errs.append(float(np.nanmax(e)) if np.all(np.isfinite(P)) else float("nan"))
    print("nv", nv, "max rel err per run:", " ".join("%.1e" % e for e in errs), flush=True)
