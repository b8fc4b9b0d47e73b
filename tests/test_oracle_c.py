"""The C++/OpenMP oracle (oracle/c/tsadar_oracle.cpp: forward-mode dual numbers) pinned to the reference's golden
vector, to the NumPy restatement and to reverse-mode autodiff of the torch twin -- three independent derivations of the
same numbers before any of them is used to judge the HIP path."""
import os

import numpy as np
import pytest

import decks
import util
from oracle import c_oracle as co
from oracle import tsadar_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_reproduces_reference_golden_vector():
    """tests/test_forward/test_1d.py of the reference: deck 1d-defaults + 1d-inputs, ThryE-1d.npy, rtol 1e-4 there."""
    cfg = decks.deck_1d()
    sa = dict(sa=util.P9["sa"], weights=util.P9["weights"])  # forward tests pass the 1-D weights: [0] is a scalar (Q5)
    normed = orc.init_normed_params(cfg["parameters"], 1, True)
    X = util.normed_to_matrix(normed, 1)
    batch = dict(e_amps=np.array([1.0]), i_amps=np.array([1.0]), noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    _, _, E, _ = co.loss_grad(cfg, sa, X, batch)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_ThryE-1d.npy"))
    assert E.shape == gold.shape == (1, 1024)
    assert np.max(np.abs(E - gold) / np.abs(gold)) < 1e-11


@pytest.mark.parametrize("n_ion,G,ppp", [(1, 1, 1), (2, 3, 2)])
def test_c_oracle_matches_numpy_and_autodiff(n_ion, G, ppp):
    from oracle import tsadar_oracle_torch as ot

    active = ("Te", "ne", "Ti", "Z", "lam", "Va", "ud", "amp1", "amp2", "amp3") + (("Te_gradient", "ne_gradient") if G > 1 else ())
    cfg = decks.deck_fit(active=active, n_ion=n_ion, points_per_pixel=ppp)
    g = cfg["parameters"]["general"]
    g["Te_gradient"].update(val=5.0, num_grad_points=G)
    g["ne_gradient"].update(val=8.0, num_grad_points=G)
    B = 2
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=21)
    batch["noise_e"] = 0.01 * np.random.default_rng(0).random((B, 1024))
    normed = util.random_lineouts(cfg, B, seed=22, ranges=dict(ud=(-1, 1)))
    X = util.normed_to_matrix(normed, n_ion)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    Eo, Io, lamE, lamI = orc.ts_diag(cfg, sa, normed, batch)
    iaw, blue, red = orc.fit_masks(cfg, lamE, lamI)
    w = np.array([cfg["data"]["ion_loss_scale"] / iaw.sum() / i_norm**2, 0.5 / blue.sum() / e_norm**2, 0.5 / red.sum() / e_norm**2])
    sm = util.SlotMap(cfg["parameters"], True)
    names = [k for k in normed if sm.active[util.slot_of(k)]]
    sums, grad, E, I = co.loss_grad(cfg, sa, X, batch, w=w, gmask=sm.active.astype(np.uint8), nthreads=2)
    assert util.rel_err(E, Eo) < 1e-10 and util.rel_err(I, Io) < 1e-9
    val, ref, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    assert abs(float(np.dot(sums.sum(axis=0), w)) - val) < 1e-10 * abs(val)
    scale = max(np.max(np.abs(v)) for v in ref.values())
    for k in names:
        assert np.max(np.abs(grad[:, util.slot_of(k)] - ref[k])) / scale < 1e-8, k


def test_c_oracle_chi_table():
    cfg = decks.deck_fit()
    fe = orc.dlm_fe(3.1, 128)
    W = co.chi_table(cfg, util.sa_fit(1), fe)
    Wo, _ = orc.chi_table(orc.velocity_grid(128), fe)
    assert np.max(np.abs(W - Wo)) / np.max(np.abs(Wo)) < 1e-12
