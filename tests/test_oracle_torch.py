"""The torch autograd twin of the oracle: same forward values as the NumPy oracle, gradients
consistent with finite differences.  CPU only."""
import numpy as np

import decks
import util
from oracle import tsadar_oracle as orc
from oracle import tsadar_oracle_torch as ot


def _case(active, n_ion=1, B=2, seed=4, tweak=None):
    cfg = decks.deck_fit(active=active, n_ion=n_ion)
    if tweak:
        tweak(cfg)
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=seed)
    normed = util.random_lineouts(cfg, B, seed=seed + 50)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    return cfg, sa, batch, normed, i_norm, e_norm


def test_twin_forward_equals_numpy_oracle():
    cfg, sa, batch, normed, i_norm, e_norm = _case(("Te", "ne", "Ti", "Va", "lam", "amp1"))
    lo, Eo, Io = orc.loss(cfg, sa, normed, batch, i_norm, e_norm)
    val, g, E, I = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, ["Te"])
    assert abs(val - lo) < 1e-11 * abs(lo)
    assert util.rel_err(E, Eo) < 1e-9 and util.rel_err(I, Io) < 1e-9


def test_twin_gradient_vs_finite_differences():
    """Smooth leaves agree with central differences to ~1e-6; Te/lam move points across kinks of the
    piecewise-linear Z' table, where FD is only first-order accurate -> looser bound."""
    names = ["Te", "ne", "Ti_1", "lam", "amp1", "amp2", "amp3", "Va"]
    cfg, sa, batch, normed, i_norm, e_norm = _case(("Te", "ne", "Ti", "lam", "amp1", "amp2", "amp3", "Va"))
    val, g, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    fd = orc.fd_gradient(cfg, sa, normed, batch, i_norm, e_norm, names, h=1e-6)
    scale = max(np.max(np.abs(v)) for v in fd.values())
    for k in names:
        tol = 1e-4 if k in ("Te", "lam", "ne") else 2e-6
        assert np.max(np.abs(g[k] - fd[k])) / scale < tol, (k, g[k], fd[k])


def test_twin_dlm_order_gradient():
    """d loss / d m through f_e -> {ln f_e Hermite table, W table}: autodiff vs finite differences."""
    cfg, sa, batch, normed, i_norm, e_norm = _case(("Te", "ne", "m", "amp1", "amp2", "lam"), B=1, seed=6)
    normed["m"] = np.array([-0.3])  # m ~ 3.28, inside a table cell
    val, g, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, ["m", "Te"])
    fd = orc.fd_gradient(cfg, sa, normed, batch, i_norm, e_norm, ["m"], h=1e-5)
    assert abs(g["m"][0] - fd["m"][0]) < 1e-4 * max(abs(fd["m"][0]), abs(g["Te"][0]))
