"""Diagnostic: adjoint of the 2-D path vs central differences at several step sizes (kinks of the linear
interpolation at |xi_e| make large steps noisy for stiff parameters such as lam)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import decks, util
from oracle import tsadar_oracle as orc
from tsadar_amd.engine import Engine

nv, n_ion, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = decks.deck_fit(n_ion=n_ion)
if G > 1:
    g = cfg["parameters"]["general"]
    g["Te_gradient"].update(val=6.0, num_grad_points=G)
    g["ne_gradient"].update(val=9.0, num_grad_points=G)
B = 2
sa = dict(sa=np.array([35.0, 60.0, 110.0]), weights=np.ones((B, 3)) / 3)
eng = Engine(cfg, sa)
normed = util.random_lineouts(cfg, B, seed=67, ranges=dict(ud=(-1.5, 1.5)))
phys = orc.physical_params(cfg["parameters"], normed, True)
phys["ud"] = np.array([0.8, -1.1])
X = util.normed_to_matrix(phys, n_ion)
vx = orc.velocity_grid(nv)
XX, YY = np.meshgrid(vx, vx, indexing="ij")
fe2 = np.exp(-((XX / 1.3) ** 2 + (YY / 0.8) ** 2) ** 1.4 / 2) + 0.05 * np.exp(-((XX - 2.0) ** 2 + (YY + 1.0) ** 2))
fe2 /= fe2.sum() * (vx[1] - vx[0]) ** 2
rng = np.random.default_rng(8)
names = ["Te", "ne", "lam", "ud", "Va", "Ti_1", "Z_1"] + (["Te_gradient", "ne_gradient", "Ti_2", "Z_2", "fract_1"] if G > 1 else [])
for feature in (0, 1):
    P0 = eng.form_factor_2d(feature, X, fe2, 25.0, -40.0)
    Pbar = torch.as_tensor(rng.standard_normal(tuple(P0.shape)), device=P0.device) / P0.abs().mean()
    J = lambda Xm, f: float((eng.form_factor_2d(feature, Xm, f, 25.0, -40.0) * Pbar).sum())
    gp, gf = eng.form_factor_2d_grad(feature, X, fe2, Pbar, 25.0, -40.0)
    gp, gf = gp.cpu().numpy(), gf.cpu().numpy()
    for b in range(B):
        for nm in names:
            s = util.slot_of(nm)
            row = []
            for hr in (1e-5, 1e-6, 1e-7, 1e-8):
                h = hr * max(abs(X[b, s]), 1e-2)
                Xp, Xm = X.copy(), X.copy(); Xp[b, s] += h; Xm[b, s] -= h
                row.append((J(Xp, fe2) - J(Xm, fe2)) / (2 * h))
            print(feature, b, f"{nm:12s} adj {gp[b, s]: .8e}  fd " + " ".join(f"{r: .8e}" for r in row), flush=True)
    for (i, j) in [(0, 0), (0, 7), (nv - 1, nv - 1), (nv // 2, nv // 2), (nv // 2 + 3, nv // 2 - 5), (nv - 1, 3), (1, nv - 2)]:
        row = []
        for hr in (1e-4, 1e-6):
            h = hr * fe2.max()
            fp, fm = fe2.copy(), fe2.copy(); fp[i, j] += h; fm[i, j] -= h
            row.append((J(X, fp) - J(X, fm)) / (2 * h))
        print(feature, "table", (i, j), f"adj {gf[i, j]: .8e}  fd " + " ".join(f"{r: .8e}" for r in row), flush=True)
