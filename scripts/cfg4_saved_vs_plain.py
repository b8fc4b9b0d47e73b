"""configs[3] adjoint: the fit-loop form (projection records of the forward) against the plain one -- where and by how much the
table adjoint differs, and how far two PLAIN calls differ from each other (LDS-atomic order).  Run on the GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import decks, util
import test_gpu_parity as T
from oracle import tsadar_oracle as orc

cfg = decks.deck_fit()
na = 512
sa = dict(sa=np.linspace(19.0, 139.0, na), weights=np.ones((1, na)) / na)
eng = T._engine(cfg, sa)
normed = util.random_lineouts(cfg, 1, seed=71, ranges=dict(ud=(-1.5, 1.5)))
phys = orc.physical_params(cfg["parameters"], normed, True)
phys["ud"] = np.array([0.7])
X = util.normed_to_matrix(phys, 1)
vx, fe2 = T._fe2d(256, "anisotropic")
P0 = eng.form_factor_2d(0, X, fe2, 25.0, -40.0)
rng = np.random.default_rng(12)
Pbar = torch.as_tensor(rng.standard_normal(tuple(P0.shape)), device=P0.device) / P0.abs().mean()
gp, gf = eng.form_factor_2d_grad(0, X, fe2, Pbar, 25.0, -40.0)
gpb, gfb = eng.form_factor_2d_grad(0, X, fe2, Pbar, 25.0, -40.0)
P1 = eng.form_factor_2d(0, X, fe2, 25.0, -40.0, save=True)
gp3, gf3 = eng.form_factor_2d_grad(0, X, fe2, Pbar, 25.0, -40.0, use_saved=True)
gp, gf, gpb, gfb, gp3, gf3 = (t.cpu().numpy() for t in (gp, gf, gpb, gfb, gp3, gf3))
m = np.max(np.abs(gf))
print("library", os.environ.get("TSFF_LIBRARY", "in-tree"))
print("P1 == P0:", bool((P1 == P0).all()))
print("plain vs plain   : grad_phys %.3e  grad_fe %.3e (of the largest entry)" % (np.max(np.abs(gpb - gp)) / np.max(np.abs(gp)), np.max(np.abs(gfb - gf)) / m))
print("saved vs plain   : grad_phys %.3e  grad_fe %.3e" % (np.max(np.abs(gp3 - gp)) / np.max(np.abs(gp)), np.max(np.abs(gf3 - gf)) / m))
d = np.abs(gf3 - gf)
i, j = np.unravel_index(np.argmax(d), d.shape)
print("largest difference at", (i, j), "value", gf[i, j], "saved", gf3[i, j], "plain again", gfb[i, j], "| entries above 1e-12 of the max:", int((d > 1e-12 * m).sum()))
