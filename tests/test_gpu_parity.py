"""GPU parity tests: every call goes through the C ABI of libtsff.so (ctypes) and is compared with
the CPU oracle on the same seeded inputs.  Floating point (float64): tolerance 1e-5 relative as
BASELINE.json's north_star states; most checks are far tighter and say so."""
import copy

import numpy as np
import pytest

import decks
import util
from oracle import tsadar_oracle as orc
from tsadar_amd import _lib as L

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # north_star tolerance


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch


def _engine(cfg, sa, **kw):
    from tsadar_amd.engine import Engine

    return Engine(cfg, sa, **kw)


def test_chi_table_matches_ratintn(torch_mod):
    """a8: W[1640] of k_fe_prepare vs oracle ratintn, Maxwellian and super-Gaussian f_e."""
    cfg = decks.deck_fit()
    eng = _engine(cfg, util.sa_fit(1))
    nvx = cfg["parameters"]["electron"]["fe"]["nvx"]
    vx = orc.velocity_grid(nvx)
    fes = np.stack([orc.dlm_fe(m, nvx) for m in (2.0, 2.5158, 3.7, 5.0)])
    W = eng.chi_table(fes).cpu().numpy()
    for k in range(fes.shape[0]):
        Wo, _ = orc.chi_table(vx, fes[k])
        err = np.max(np.abs(W[k] - Wo)) / np.max(np.abs(Wo))
        assert err < 1e-11, (k, err)


@pytest.mark.parametrize("feature", [0, 1])
def test_form_factor_matches_oracle(torch_mod, feature):
    """a4-a10: raw FormFactor.__call__ output P[G, npts, ntheta] for random plasma conditions."""
    cfg = decks.deck_fit()
    B = 4
    sa = util.sa_fit(B)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, B, seed=11 + feature)
    phys = orc.physical_params(cfg["parameters"], normed, True)
    X = util.normed_to_matrix(phys, 1)
    P = eng.form_factor(feature, X).cpu().numpy()
    nvx = cfg["parameters"]["electron"]["fe"]["nvx"]
    vx, fe = orc.velocity_grid(nvx), orc.dlm_fe(2.0, nvx)
    rng = cfg["other"]["lamrangE"] if feature == 0 else cfg["other"]["lamrangI"]
    for b in range(B):
        p = orc.lineout_params(phys, b, 1)
        Po, _ = orc.form_factor(rng, cfg["other"]["npts"], 0.0, sa["sa"], 1, p, vx, fe)
        err = np.max(np.abs(P[b] - Po) / np.abs(Po))
        assert err < 1e-7, (b, err)  # IAW resonance amplifies rounding (|eps| << 1)


def test_forward_reference_golden(torch_mod):
    """The reference's own golden vector (tests/test_forward/test_1d.py:69,84: rtol 1e-4, atol 0),
    computed by the HIP path: EPW only, DLM m=2.5 (through the Q6 round trip), 5120 -> 1024 samples."""
    from tsadar_amd import ThomsonParams

    cfg = decks.deck_1d()
    eng = _engine(cfg, util.P9, activate=True)
    tp = ThomsonParams(cfg["parameters"], num_params=1, batch=True, activate=True)
    E, _ = eng.forward(tp.to_matrix(), np.array([1]), np.array([1]), np.array([0]), np.array([0]))
    golden = np.load("tests/golden/ref_ThryE-1d.npy")
    np.testing.assert_allclose(E.cpu().numpy(), golden, rtol=RTOL, atol=0)


@pytest.mark.parametrize("ppp", [1, 2])
def test_forward_matches_oracle(torch_mod, ppp):
    """a1-a13: ThryE/ThryI of B random lineouts, EPW + IAW, with noise and amplitudes."""
    cfg = decks.deck_fit(points_per_pixel=ppp)
    B = 6
    sa = util.sa_fit(B)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, B, seed=3)
    rng = np.random.default_rng(5)
    batch = dict(e_amps=rng.uniform(0.5, 2, B), i_amps=rng.uniform(0.5, 2, B),
                 noise_e=0.01 * rng.random((B, 1024)), noise_i=0.01 * rng.random((B, 1024)),
                 e_data=np.ones((B, 1024)), i_data=np.ones((B, 1024)))
    Eo, Io, lE, lI = orc.ts_diag(cfg, sa, normed, batch)
    E, I = eng.forward(util.normed_to_matrix(normed, 1), batch["e_amps"], batch["i_amps"], batch["noise_e"], batch["noise_i"])
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-9
    assert util.rel_err(I.cpu().numpy(), Io) < 1e-8
    np.testing.assert_allclose(eng.lamAxisE, lE[0], rtol=1e-14)
    np.testing.assert_allclose(eng.lamAxisI, lI[0], rtol=1e-14)


def test_forward_two_ions_gradient_points(torch_mod):
    """a4-a10 with n_ion = 2 and num_grad_points = 3 (non-zero Te/ne gradients)."""
    cfg = decks.deck_fit(n_ion=2)
    g = cfg["parameters"]["general"]
    g["Te_gradient"].update(val=6.0, num_grad_points=3)
    g["ne_gradient"].update(val=9.0, num_grad_points=3)
    B = 3
    sa = util.sa_fit(B)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, B, seed=21)
    batch = dict(e_amps=np.ones(B), i_amps=np.ones(B), noise_e=np.zeros((B, 1024)), noise_i=np.zeros((B, 1024)))
    Eo, Io, _, _ = orc.ts_diag(cfg, sa, normed, batch)
    E, I = eng.forward(util.normed_to_matrix(normed, 2), batch["e_amps"], batch["i_amps"])
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-9
    assert util.rel_err(I.cpu().numpy(), Io) < 1e-8


def test_forward_dlm_per_lineout(torch_mod):
    """a2: DLM f_e built on the GPU from a per-lineout m (fe active), then the full chain."""
    cfg = decks.deck_fit(active=("Te", "ne", "m", "amp1", "amp2", "lam"))
    B = 4
    sa = util.sa_fit(B)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, B, seed=8, ranges=dict(m=(2.0, 3.5)))
    batch = dict(e_amps=np.ones(B), i_amps=np.ones(B), noise_e=np.zeros((B, 1024)), noise_i=np.zeros((B, 1024)))
    Eo, Io, _, _ = orc.ts_diag(cfg, sa, normed, batch)
    E, I = eng.forward(util.normed_to_matrix(normed, 1), batch["e_amps"], batch["i_amps"])
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8
    assert util.rel_err(I.cpu().numpy(), Io) < 1e-8


def _loss_setup(cfg, B, seed=2):
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=seed)
    normed = util.random_lineouts(cfg, B, seed=seed + 50)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    return sa, batch, normed, i_norm, e_norm


def test_loss_value_matches_oracle(torch_mod):
    """a14: masked nanmean loss (iaw + blue + red) vs the oracle."""
    cfg = decks.deck_fit()
    B = 5
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B)
    eng = _engine(cfg, sa)
    lo, Eo, Io = orc.loss(cfg, sa, normed, batch, i_norm, e_norm)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms, grad, E, I = eng.loss_grad(util.normed_to_matrix(normed, 1), batch, w, eng.slots.active.astype(np.uint8), want_spectra=True)
    val = float(np.dot(terms.cpu().numpy(), w))
    assert abs(val - lo) / abs(lo) < 1e-9, (val, lo)
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-9
    assert util.rel_err(I.cpu().numpy(), Io) < 1e-8


def _grad_case(torch_mod, active, names, B, seed, n_ion=1, tweak=None, tol=1e-7, ppp=1, plan=0):
    from oracle import tsadar_oracle_torch as ot

    cfg = decks.deck_fit(active=active, n_ion=n_ion, points_per_pixel=ppp)
    if tweak:
        tweak(cfg)
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=seed)
    eng = _engine(cfg, sa)
    eng.set_launch_plan(plan)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms, grad, _, _ = eng.loss_grad(util.normed_to_matrix(normed, n_ion), batch, w, eng.slots.active.astype(np.uint8))
    G = util.matrix_to_named(grad.cpu().numpy(), names)
    val, ref, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    assert abs(float(np.dot(terms.cpu().numpy(), w)) - val) < 1e-9 * abs(val)
    scale = max(np.max(np.abs(v)) for v in ref.values())
    for k in names:
        err = np.max(np.abs(G[k] - ref[k])) / scale
        assert err < tol, (k, G[k], ref[k])
    # leaves that are not trainable get a zero gradient
    g = grad.cpu().numpy()
    for s in range(g.shape[1]):
        if not eng.slots.active[s]:
            assert np.all(g[:, s] == 0.0)


def test_gradient_matches_autodiff_all_leaves(torch_mod):
    """a15: the hand-written adjoint vs reverse-mode autodiff of the torch oracle twin (the role JAX
    autodiff plays in the reference), every differentiable scalar leaf of a 1-ion deck, G = 1."""
    names = ["Te", "ne", "Ti_1", "Z_1", "lam", "amp1", "amp2", "amp3", "ud", "Va", "Te_gradient", "ne_gradient"]

    def tweak(cfg):
        cfg["parameters"]["general"]["Te_gradient"]["val"] = 3.0
        cfg["parameters"]["general"]["ne_gradient"]["val"] = 4.0

    _grad_case(torch_mod, ("Te", "ne", "Ti", "Z", "lam", "amp1", "amp2", "amp3", "ud", "Va", "Te_gradient", "ne_gradient"),
               names, B=2, seed=4, tweak=tweak)


@pytest.mark.parametrize("n_ion", [1, 2])
def test_gradient_two_sweep_kernel(torch_mod, n_ion):
    """a15 by the two-sweep kernel (TSFF_OPT_LAUNCH_PLAN bit 1): the default plan runs the one-sweep kernel
    (k_spectrum_fused) for these decks, so the general kernel is pinned to the autodiff twin separately."""
    if n_ion == 1:
        names = ["Te", "ne", "Ti_1", "Z_1", "lam", "amp1", "amp2", "amp3", "ud", "Va", "Te_gradient", "ne_gradient"]
        active = ("Te", "ne", "Ti", "Z", "lam", "amp1", "amp2", "amp3", "ud", "Va", "Te_gradient", "ne_gradient")
    else:
        names = ["Te", "ne", "Ti_1", "Ti_2", "Z_1", "lam", "Va", "amp1", "amp3"]
        active = ("Te", "ne", "Ti", "Z", "lam", "Va", "amp1", "amp3")

    def tweak(cfg):
        cfg["parameters"]["general"]["Te_gradient"]["val"] = 3.0
        cfg["parameters"]["general"]["ne_gradient"]["val"] = 4.0

    _grad_case(torch_mod, active, names, B=2, seed=4 + n_ion, n_ion=n_ion, tweak=tweak, plan=2)


def test_gradient_one_sweep_two_ions(torch_mod):
    """a15 by the one-sweep kernel with n_ion = 2 and one gradient point (4 x 13 Jacobian-row accumulators per thread)."""
    names = ["Te", "ne", "Ti_1", "Ti_2", "Z_1", "Z_2", "lam", "Va", "ud", "amp1", "amp2", "amp3"]

    def tweak(cfg):
        cfg["parameters"]["ion-2"]["Z"]["active"] = True

    _grad_case(torch_mod, ("Te", "ne", "Ti", "Z", "lam", "Va", "ud", "amp1", "amp2", "amp3"), names, B=3, seed=15, n_ion=2, tweak=tweak)


def test_gradient_baseline_active_set(torch_mod):
    """a15 on the BASELINE active set {Te, ne, Ti, Va, lam, amp1}."""
    _grad_case(torch_mod, ("Te", "ne", "Ti", "Va", "lam", "amp1"), ["Te", "ne", "Ti_1", "Va", "lam", "amp1"], B=3, seed=9)


def test_gradient_production_velocity_grid(torch_mod):
    """a15 with nvx = 320, the velocity grid of the reference's production decks (configs/1d/inputs.yaml:52): larger
    Hermite tables in LDS, the launch planner's budget decisions change."""
    def tweak(cfg):
        cfg["parameters"]["electron"]["fe"]["nvx"] = 320

    _grad_case(torch_mod, ("Te", "ne", "Ti", "Va", "lam", "amp1"), ["Te", "ne", "Ti_1", "Va", "lam", "amp1"], B=2, seed=19, tweak=tweak)


@pytest.mark.parametrize("nvx", [128, 320])
def test_gradient_dlm_order(torch_mod, nvx):
    """the reference's canonical active set {Te, ne, m, amp1, amp2, lam} -- the gradient
    w.r.t. the super-Gaussian order m flows through the ln f_e Hermite table and the W table
    (per-lineout tables by k_fe_vectors + k_wgemm).  nvx = 320: the production velocity grid."""
    def tweak(cfg):
        cfg["parameters"]["electron"]["fe"]["params"]["m"]["val"] = 2.7
        cfg["parameters"]["electron"]["fe"]["nvx"] = nvx

    B = 3
    names = ["Te", "ne", "m", "amp1", "amp2", "lam"]
    from oracle import tsadar_oracle_torch as ot

    cfg = decks.deck_fit(active=("Te", "ne", "m", "amp1", "amp2", "lam"))
    tweak(cfg)
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=41)
    normed = util.random_lineouts(cfg, B, seed=43, ranges=dict(m=(2.05, 4.4)))
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    eng = _engine(cfg, sa)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms, grad, E, I = eng.loss_grad(util.normed_to_matrix(normed, 1), batch, w, eng.slots.active.astype(np.uint8), want_spectra=True)
    val, ref, Eo, Io = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8 and util.rel_err(I.cpu().numpy(), Io) < 1e-8
    assert abs(float(np.dot(terms.cpu().numpy(), w)) - val) < 1e-9 * abs(val)
    G = util.matrix_to_named(grad.cpu().numpy(), names)
    scale = max(np.max(np.abs(v)) for v in ref.values())
    for k in names:
        assert np.max(np.abs(G[k] - ref[k])) / scale < 1e-7, (k, G[k], ref[k])


def test_dlm_step_in_column_blocks_is_bit_identical(torch_mod):
    """TSFF_OPT_DLM_BLOCKS: the per-lineout tables built block by block on the handle's second stream and the one-sweep kernel
    launched per block (offset b0 into batch-global records) give the bits of the one-stream step -- loss sums, gradient, spectra --
    with a ragged last block (B = 600 -> blocks of 256, 256, 88), through both output forms, call after call."""
    B = 600
    cfg = decks.deck_fit(active=("Te", "ne", "m", "amp1", "amp2", "lam"))
    cfg["parameters"]["electron"]["fe"]["params"]["m"]["val"] = 2.7
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=5)
    normed = util.random_lineouts(cfg, B, seed=6, ranges=dict(m=(2.05, 4.4)))
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    eng = _engine(cfg, sa)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    x = util.normed_to_matrix(normed, 1)
    gm = eng.slots.active.astype(np.uint8)
    ref = [t.cpu().numpy() for t in eng.loss_grad(x, batch, w, gm, want_spectra=True)]
    assert np.all(ref[1][:, 2] != 0.0)   # (slot TSFF_P_M: the DLM order is a leaf of every lineout)
    for nblk in (3, 2, 3):
        eng.set_dlm_blocks(nblk)
        got = [t.cpu().numpy() for t in eng.loss_grad(x, batch, w, gm, want_spectra=True)]
        for a, b_ in zip(ref, got):
            assert np.array_equal(a, b_), nblk
    eng.set_dlm_blocks(1)
    got = [t.cpu().numpy() for t in eng.loss_grad(x, batch, w, gm, want_spectra=True)]
    for a, b_ in zip(ref, got):
        assert np.array_equal(a, b_)


def test_per_lineout_tables_match_log_sum(torch_mod):
    """The matrix-vector form of the W table (k_fe_vectors + k_wgemm, per-lineout f_e) equals the direct
    1640 x 1022 logarithm sum (k_fe_prepare) and the oracle: forward with explicit per-lineout f_e."""
    from tsadar_amd import _lib

    cfg = decks.deck_fit()
    B = 5
    sa = util.sa_fit(B)
    eng = _engine(cfg, sa, fe_mode=_lib.FE_PER_LINEOUT)
    nvx = cfg["parameters"]["electron"]["fe"]["nvx"]
    fes = np.stack([orc.dlm_fe(m, nvx) for m in (2.0, 2.3, 3.1, 4.2, 5.0)])
    normed = util.random_lineouts(cfg, B, seed=47)
    batch = dict(e_amps=np.ones(B), i_amps=np.ones(B), noise_e=np.zeros((B, 1024)), noise_i=np.zeros((B, 1024)))
    Eo, Io, _, _ = orc.ts_diag(cfg, sa, normed, batch, fe_batch=fes)
    E, I = eng.forward(util.normed_to_matrix(normed, 1), batch["e_amps"], batch["i_amps"], fe=fes)
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8 and util.rel_err(I.cpu().numpy(), Io) < 1e-8


def test_gradient_two_points_per_pixel(torch_mod):
    """a12/a15 with points_per_pixel = 2 (2048 wavelength samples binned to 1024): the generic
    convolution / binning adjoint."""
    _grad_case(torch_mod, ("Te", "ne", "Ti", "Va", "lam", "amp1", "amp2", "amp3"),
               ["Te", "ne", "Ti_1", "Va", "lam", "amp1", "amp2", "amp3"], B=2, seed=11, ppp=2)


def test_gradient_two_ions_three_gradient_points(torch_mod):
    """a15 with n_ion = 2 (fraction renormalisation, Zbar coupling) and num_grad_points = 3."""
    def tweak(cfg):
        g = cfg["parameters"]["general"]
        g["Te_gradient"].update(val=5.0, num_grad_points=3)
        g["ne_gradient"].update(val=8.0, num_grad_points=3)
        cfg["parameters"]["ion-2"]["Z"]["active"] = True

    names = ["Te", "ne", "Ti_1", "Ti_2", "Z_1", "Z_2", "lam", "Va", "Te_gradient", "ne_gradient", "amp1", "amp2", "amp3"]
    _grad_case(torch_mod, ("Te", "ne", "Ti", "Z", "lam", "Va", "Te_gradient", "ne_gradient", "amp1", "amp2", "amp3"),
               names, B=2, seed=13, n_ion=2, tweak=tweak)


def test_gradient_tied_ion_temperature(torch_mod):
    """ion-2.Ti.same = True ties Ti_2 to Ti_1 (ts_params.py:557-558): its adjoint flows to Ti_1."""
    def tweak(cfg):
        cfg["parameters"]["ion-2"]["Ti"]["same"] = True

    _grad_case(torch_mod, ("Te", "ne", "Ti", "lam"), ["Te", "ne", "Ti_1", "Ti_2", "lam"], B=2, seed=17, n_ion=2, tweak=tweak)


def test_no_ion_irf_passthrough(torch_mod):
    """spect_stddev_ion == 0 (irf.py:82-86) -- no instrument response for the ion feature: ThryI = modlI + noise_i,
    neither convolved nor normalised, amp3 and i_amps without effect.  Forward vs the NumPy oracle, loss + gradient vs autodiff
    of the torch twin, by the one-sweep and by the two-sweep kernel."""
    from oracle import tsadar_oracle_torch as ot

    cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "lam", "amp1", "amp3"))
    cfg["other"]["PhysParams"]["widIRF"]["spect_stddev_ion"] = 0
    B = 3
    sa = util.sa_fit(B)
    truth = util.random_lineouts(cfg, B, seed=301)
    zero = np.zeros((B, 1024))
    unit = dict(e_amps=np.ones(B), i_amps=np.ones(B), noise_e=zero, noise_i=zero, e_data=zero + 1, i_data=zero + 1)
    E, I, lE, lI = orc.ts_diag(cfg, sa, truth, unit)
    rng = np.random.default_rng(302)
    batch = dict(e_data=E * (1 + 0.01 * rng.standard_normal(E.shape)), i_data=I * (1 + 0.01 * rng.standard_normal(I.shape)),
                 e_amps=E.max(axis=1), i_amps=rng.uniform(0.5, 2.0, B), noise_e=0.01 * rng.random((B, 1024)),
                 noise_i=1e-3 * I.max() * rng.random((B, 1024)))
    normed = util.random_lineouts(cfg, B, seed=303)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    names = ["Te", "ne", "Ti_1", "Va", "lam", "amp1", "amp3"]
    val, ref, Eo, Io = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    En, In, _, _ = orc.ts_diag(cfg, sa, normed, batch)
    assert np.max(np.abs(ref["amp3"])) == 0.0
    scale = max(np.max(np.abs(v)) for v in ref.values())
    eng = _engine(cfg, sa)
    X = util.normed_to_matrix(normed, 1)
    Ef, If = eng.forward(X, batch["e_amps"], batch["i_amps"], batch["noise_e"], batch["noise_i"])
    assert util.rel_err(Ef.cpu().numpy(), En) < 1e-9 and util.rel_err(If.cpu().numpy(), In) < 1e-8
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    for plan in (0, 2):
        eng.set_launch_plan(plan)
        terms, grad, Eg, Ig = eng.loss_grad(X, batch, w, eng.slots.active.astype(np.uint8), want_spectra=True)
        assert bool((Ig == If).all()) and bool((Eg == Ef).all())
        assert abs(float(np.dot(terms.cpu().numpy(), w)) - val) < 1e-9 * abs(val)
        G = util.matrix_to_named(grad.cpu().numpy(), names)
        for k in names:
            assert np.max(np.abs(G[k] - ref[k])) < 1e-7 * scale, (plan, k, G[k], ref[k])
    # [npts] samples only match the [1024] data with one point per pixel: refused otherwise, as the reference's shapes would
    cfg2 = decks.deck_fit(points_per_pixel=2)
    cfg2["other"]["PhysParams"]["widIRF"]["spect_stddev_ion"] = 0
    with pytest.raises(L.TsffError, match="spect_stddev_ion == 0"):
        _engine(cfg2, sa)


def test_committed_golden_fixture(torch_mod):
    """tests/golden/oracle_fit_b4.npz (made by tests/golden/make_golden.py from the oracle): spectra,
    loss, gradient and array_loss of four BASELINE lineouts through the drop-in LossFunction API."""
    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd.loss_function import LossFunction

    z = np.load("tests/golden/oracle_fit_b4.npz")
    B = 4
    cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "lam", "amp1"))
    cfg["optimizer"]["batch_size"] = B
    sa = util.sa_fit(B)
    batch = {k: z[k] for k in ("e_data", "i_data", "e_amps", "i_amps", "noise_e", "noise_i")}
    lf = LossFunction(cfg, sa, batch)
    assert lf.i_norm == float(z["i_norm"]) and lf.e_norm == float(z["e_norm"])
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    tp.X[:] = z["X"]
    diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
    x0, lf.unravel_weights = tree.ravel_pytree(diff)
    value, flat = lf.vg_loss(x0, static, batch)  # exactly how loops.py:43-51 calls it
    assert isinstance(value, float) and flat.dtype == np.float64 and flat.shape == (6 * B,)
    assert abs(value - float(z["loss"])) < 1e-9 * abs(float(z["loss"]))
    # fixture columns: Te, ne, Ti_1, Va, lam, amp1; ravel order (pytree field order): Te, ne, Ti_1, lam, amp1, Va
    gref = z["grad"][:, [0, 1, 2, 4, 5, 3]].T.reshape(-1)
    assert np.max(np.abs(flat - gref)) < 1e-7 * np.max(np.abs(gref))
    E, I, lamE, lamI = lf.ts_diag(tp, batch)
    assert util.rel_err(E, z["ThryE"]) < 1e-8 and util.rel_err(I, z["ThryI"]) < 1e-8
    assert lamE.shape == lamI.shape == (B, 1024)
    total, sqdev, E2, I2, params = lf.array_loss(tp, batch)
    np.testing.assert_allclose(total, z["array_loss"], rtol=1e-7)
    np.testing.assert_allclose(sqdev["ele"], z["sqdev_ele"], rtol=1e-6, atol=1e-12 * z["sqdev_ele"].max())
    np.testing.assert_allclose(sqdev["ion"], z["sqdev_ion"], rtol=1e-6, atol=1e-12 * z["sqdev_ion"].max())
    assert set(params) == {"electron", "general", "ion-1"}


def test_full_size_properties(torch_mod):
    """BASELINE size (B = 4096, configs[2]) through size-independent properties: batch invariance
    (a lineout's spectrum and gradient do not depend on its batch mates), linearity in the amplitudes,
    additivity of the loss sums, 1/N scaling of the gradient; plus an oracle spot check of 3 of the 4096."""
    from tsadar_amd import synthetic as S
    from tsadar_amd.engine import Engine

    B = 4096
    cfg = S.baseline_deck(batch_size=B)
    sa = util.sa_fit(B)
    eng = Engine(cfg, sa)
    rng = np.random.default_rng(S.SEED)
    truth = S.draw_params(cfg, B, rng)
    batch = S.make_batch(eng, truth, rng)
    guess = S.draw_params(cfg, B, rng)
    X = guess.to_matrix()
    gm = guess.grad_mask()
    e_norm, i_norm = float(batch["e_data"].max()), float(batch["i_data"].max())
    w = eng.loss_weights(B, i_norm, e_norm)
    terms, grad, E, I = eng.loss_grad(X, batch, w, gm, want_spectra=True)
    terms, grad, E, I = terms.cpu().numpy(), grad.cpu().numpy(), E.cpu().numpy(), I.cpu().numpy()
    assert np.all(np.isfinite(grad)) and np.all(np.isfinite(E)) and np.all(np.isfinite(I))
    # batch invariance + 1/N scaling: re-evaluate a scattered subset on its own
    idx = np.array([0, 1, 257, 1023, 2048, 4095])
    sub = {k: (v[torch_mod.as_tensor(idx, device=v.device)] if v is not None else None) for k, v in batch.items()}
    w_sub = eng.loss_weights(len(idx), i_norm, e_norm)
    t2, g2, E2, I2 = eng.loss_grad(X[idx], sub, w_sub, gm, want_spectra=True)
    np.testing.assert_array_equal(E2.cpu().numpy(), E[idx])
    np.testing.assert_array_equal(I2.cpu().numpy(), I[idx])
    np.testing.assert_allclose(g2.cpu().numpy() * (len(idx) / B), grad[idx], rtol=1e-10, atol=1e-14 * np.abs(grad).max())
    # additivity: the batch sums are the sum of the per-lineout sums
    per = np.stack([eng.loss_grad(X[i:i + 1], {k: (v[i:i + 1] if v is not None else None) for k, v in batch.items()},
                                  w, gm)[0].cpu().numpy() for i in idx])
    np.testing.assert_allclose(per.sum(axis=0), t2.cpu().numpy(), rtol=1e-12)
    # linearity in the amplitudes (noise is zero): doubling e_amps doubles ThryE
    E3, I3 = eng.forward(X[idx], 2 * sub["e_amps"], 3 * sub["i_amps"])
    np.testing.assert_allclose(E3.cpu().numpy(), 2 * E[idx], rtol=1e-14)
    np.testing.assert_allclose(I3.cpu().numpy(), 3 * I[idx], rtol=1e-14)
    # oracle spot check
    for b in (5, 1999, 4000):
        normed = {k: X[b:b + 1, util.slot_of(k)] for k in
                  ["Te", "ne", "m", "Ti_1", "Z_1", "A_1", "fract_1", "lam", "amp1", "amp2", "amp3", "ne_gradient", "Te_gradient", "ud", "Va"]}
        bt = dict(e_amps=batch["e_amps"][b:b + 1].cpu().numpy(), i_amps=batch["i_amps"][b:b + 1].cpu().numpy(),
                  noise_e=np.zeros((1, 1024)), noise_i=np.zeros((1, 1024)))
        Eo, Io, _, _ = orc.ts_diag(cfg, util.sa_fit(1), normed, bt)
        assert util.rel_err(E[b:b + 1], Eo) < 1e-8 and util.rel_err(I[b:b + 1], Io) < 1e-8


def test_edge_cases(torch_mod):
    """Single lineout; EPW-only and IAW-only decks; parameters at the edge of the sigmoid range;
    every loss functional; non-zero drift and flow; zero-weight masks."""
    from tsadar_amd.engine import Engine
    from oracle import tsadar_oracle_torch as ot

    for method in ("l2", "l1", "log-cosh", "poisson"):
        cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "ud", "lam", "amp1", "amp2", "amp3"))
        cfg["optimizer"]["loss_method"] = method
        B = 2
        sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=23)
        normed["ud"] = np.array([0.45, 0.55])  # identity-free: 'ud' is active -> sigmoid -> [-10, 10]
        eng = Engine(cfg, sa)
        w = eng.loss_weights(B, i_norm, e_norm)
        names = ["Te", "ne", "Ti_1", "Va", "ud", "lam", "amp1", "amp2", "amp3"]
        terms, grad, _, _ = eng.loss_grad(util.normed_to_matrix(normed, 1), batch, w, eng.slots.active.astype(np.uint8))
        val, ref, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
        assert abs(float(np.dot(terms.cpu().numpy(), w)) - val) < 1e-9 * abs(val), method
        G = util.matrix_to_named(grad.cpu().numpy(), names)
        scale = max(np.max(np.abs(v)) for v in ref.values())
        for k in names:
            assert np.max(np.abs(G[k] - ref[k])) < 1e-7 * scale, (method, k)
    # EPW only / IAW only
    for ele, ion in ((True, False), (False, True)):
        cfg = decks.deck_fit()
        ext = cfg["other"]["extraoptions"]
        ext["load_ele_spec"], ext["load_ion_spec"] = ele, ion
        ext["fit_EPWb"] = ext["fit_EPWr"] = ele
        ext["fit_IAW"] = ion
        B = 1
        sa, batch, normed, i_norm, e_norm = _loss_setup(decks.deck_fit(), B, seed=29)
        eng = Engine(cfg, sa)
        E, I = eng.forward(util.normed_to_matrix(normed, 1), batch["e_amps"], batch["i_amps"], batch["noise_e"], batch["noise_i"])
        Eo, Io, _, _ = orc.ts_diag(cfg, sa, normed, batch)
        if ele:
            assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8 and float(I.abs().max()) == 0.0
        else:
            assert util.rel_err(I.cpu().numpy(), Io) < 1e-8 and float(E.abs().max()) == 0.0
        w = eng.loss_weights(B, i_norm, e_norm)
        terms, grad, _, _ = eng.loss_grad(util.normed_to_matrix(normed, 1), batch, w, eng.slots.active.astype(np.uint8))
        lo, _, _ = orc.loss(cfg, sa, normed, batch, i_norm, e_norm)
        assert abs(float(np.dot(terms.cpu().numpy(), w)) - lo) < 1e-9 * abs(lo)
    # saturated sigmoids: physical values pinned at a bound, gradient finite and ~0
    cfg = decks.deck_fit()
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, 1, seed=31)
    normed["amp1"] = np.array([40.0])
    normed["Va"] = np.array([-40.0])
    normed["Te"] = np.array([3.0])
    eng = Engine(cfg, sa)
    w = eng.loss_weights(1, i_norm, e_norm)
    terms, grad, _, _ = eng.loss_grad(util.normed_to_matrix(normed, 1), batch, w, eng.slots.active.astype(np.uint8))
    g = grad.cpu().numpy()
    assert np.all(np.isfinite(g)) and abs(g[0, 4]) < 1e-12


def test_errors_are_reported_not_swallowed(torch_mod):
    from tsadar_amd import _lib
    from tsadar_amd.engine import Engine

    cfg = decks.deck_fit()
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, 1, seed=37)
    eng = Engine(cfg, sa)  # shared Maxwellian: the DLM order is not a leaf of this engine
    gm = eng.slots.active.astype(np.uint8)
    gm[_lib.P_M] = 1
    with pytest.raises(_lib.TsffError, match="DLM order m"):
        eng.loss_grad(util.normed_to_matrix(normed, 1), batch, eng.loss_weights(1, i_norm, e_norm), gm)
    cfg = decks.deck_fit()
    cfg["other"]["PhysParams"]["norm"] = 1
    with pytest.raises(_lib.TsffError, match="norm"):
        Engine(cfg, sa)
    # the engine mirrors the Z' table about xi = 0 when LDS is short: a table that is not even / odd is refused
    from tsadar_amd import engine as eng_mod

    real = eng_mod.zprime_tables
    try:
        eng_mod.zprime_tables = lambda xi2: (real(xi2)[0] + 1e-3 * np.arange(xi2.size), real(xi2)[1])
        with pytest.raises(_lib.TsffError, match="not even"):
            Engine(decks.deck_fit(), sa)
    finally:
        eng_mod.zprime_tables = real
    cfg = decks.deck_fit()
    cfg["other"]["extraoptions"]["spectype"] = "bogus"
    from tsadar_amd.diagnostic import ThomsonScatteringDiagnostic

    with pytest.raises(NotImplementedError, match="Unknown spectype"):
        ThomsonScatteringDiagnostic(cfg, sa)


def _dist_rank(rank, world, port, out):
    """One rank of the 2-process rehearsal: real engine on the (shared) GPU, gloo collectives."""
    import os
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TSFF_DIST_BACKEND="gloo", TSFF_FORCE_DEVICE="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    from tsadar_amd import ThomsonParams, distributed as D, tree
    from tsadar_amd.loss_function import LossFunction

    z = np.load(os.path.join(root, "tests/golden/oracle_fit_b4.npz"))
    B = 4
    D.init_from_env()
    cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "lam", "amp1"))
    lo, hi = D.shard_bounds(B, world, rank)
    full = {k: z[k] for k in ("e_data", "i_data", "e_amps", "i_amps", "noise_e", "noise_i")}
    local = {k: v[lo:hi] for k, v in full.items()}
    lf = LossFunction(cfg, util.sa_fit(hi - lo), full, distributed=True)  # norms from the global sample, as loops.py:133
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    tp.X[:] = z["X"]
    diff, _ = tree.partition(tp)
    x0, lf.unravel_weights = tree.ravel_pytree(diff)
    tpl = ThomsonParams(cfg["parameters"], hi - lo, batch=True, activate=True)
    tpl.X[:] = tp.X[lo:hi]
    value, flat = lf.vg_loss(x0, tree.StaticParams(tpl), local)
    np.save(os.path.join(out, f"r{rank}.npy"), np.concatenate([[value], flat]))
    if rank == 0:   # the same step on one rank, same process, same device
        single = LossFunction(cfg, util.sa_fit(B), full)
        single.unravel_weights = lf.unravel_weights
        v1, g1 = single.vg_loss(x0, tree.StaticParams(tp), full)
        np.save(os.path.join(out, "single.npy"), np.concatenate([[v1], g1]))
    dist.destroy_process_group()


def test_two_rank_sharded_fit_step_on_gpu(torch_mod, tmp_path):
    """two ranks (sharing this box's single GPU, gloo transport) evaluate their shards with the
    HIP engine and all-reduce [loss sums | gradient]; both hold the full-batch loss and gradient of the
    committed fixture."""
    import socket
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dist_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(r0, r1)
    # 1 rank == 2 ranks: a lineout's gradient is reduced inside its own workgroups in a fixed order and the all-reduce only adds
    # zeros to it (bit-identical); the three loss sums are folded per shard first (1e-14)
    one = np.load(tmp_path / "single.npy")
    np.testing.assert_array_equal(r0[1:], one[1:])
    assert abs(r0[0] - one[0]) <= 1e-14 * abs(one[0])
    z = np.load("tests/golden/oracle_fit_b4.npz")
    gref = z["grad"][:, [0, 1, 2, 4, 5, 3]].T.reshape(-1)
    assert abs(r0[0] - float(z["loss"])) < 1e-9 * abs(float(z["loss"]))
    assert np.max(np.abs(r0[1:] - gref)) < 1e-7 * np.max(np.abs(gref))


def test_inverse_round_trip_like_reference(torch_mod):
    """The reference's round-trip test (tests/test_inverse/test_1d_random.py:58-174): synthesise an EPW
    spectrum from random (m, Te, ne, amp1, amp2, lam) (seed 42, ranges :33-39), start from a second random
    draw and refit with scipy L-BFGS-B on value-and-gradient; every parameter must come back within
    rtol 0.1 (:173-174).  Same deck (1-D, DLM f_e, 5 points per pixel), the loss mean((ThryE - truth)^2) is
    expressed through LossFunction by fitting the whole EPW window with unit normalisation."""
    from scipy.optimize import minimize

    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd.loss_function import LossFunction

    def perturb(rng, P):
        P["electron"]["fe"]["params"]["m"]["val"] = float(rng.uniform(2.0, 3.5))
        P["electron"]["Te"]["val"] = float(rng.uniform(0.5, 1.5))
        P["electron"]["ne"]["val"] = float(rng.uniform(0.1, 0.7))
        P["general"]["amp1"]["val"] = float(rng.uniform(0.5, 2.5))
        P["general"]["amp2"]["val"] = float(rng.uniform(0.5, 2.5))
        P["general"]["lam"]["val"] = float(rng.uniform(523, 527))

    cfg = decks.deck_1d()
    ext = cfg["other"]["extraoptions"]
    ext["fit_EPWb"], ext["fit_EPWr"], ext["fit_IAW"] = True, False, False
    cfg["data"]["fit_rng"].update(blue_min=0.0, blue_max=1e4)  # every sample of the EPW window
    cfg["optimizer"].update(y_norm=False, batch_size=1)
    dummy = dict(i_data=np.array([1]), e_data=np.array([1]), noise_e=np.array([0]), noise_i=np.array([0]),
                 e_amps=np.array([1]), i_amps=np.array([1]))
    rng = np.random.default_rng(42)
    perturb(rng, cfg["parameters"])
    gt = ThomsonParams(cfg["parameters"], num_params=1, batch=True, activate=True)
    lf = LossFunction(cfg, util.P9, dummy)
    ThryE, _, lamE, _ = lf.ts_diag(gt, dummy)
    batch = dict(dummy, e_data=ThryE, i_data=np.zeros((1, 1024)))
    perturb(rng, cfg["parameters"])
    fit = ThomsonParams(cfg["parameters"], num_params=1, batch=True, activate=True)
    diff, static = tree.partition(fit, tree.get_filter_spec(cfg["parameters"], fit))
    x0, lf.unravel_weights = tree.ravel_pytree(diff)
    assert x0.shape == (6,)
    res = minimize(lf.vg_loss, x0, args=(static, batch), method="L-BFGS-B", jac=True)
    learned = tree.combine(lf.unravel_weights(res["x"]), static).get_unnormed_params()
    truth = gt.get_unnormed_params()
    for sp, k in (("electron", "Te"), ("electron", "ne"), ("electron", "m"), ("general", "amp1"), ("general", "amp2"), ("general", "lam")):
        np.testing.assert_allclose(learned[sp][k], truth[sp][k], atol=0, rtol=0.1, err_msg=f"{sp}.{k} (loss {res['fun']:.3e})")
    assert res["fun"] < 1e-4  # (the reference asserts only the parameters)


def test_hessian_for_sigmas(torch_mod):
    """LossFunction.h_loss_wrt_params (central differences of the HIP gradient of the
    reference's Hessian loss) vs the double-backward Hessian of the oracle twin, in the nested layout
    postprocess.get_sigmas reads."""
    from oracle import tsadar_oracle_torch as ot
    from tsadar_amd import ThomsonParams
    from tsadar_amd.loss_function import LossFunction

    active = ("Te", "ne", "Ti", "lam", "amp1")
    names = ["Te", "ne", "Ti_1", "lam", "amp1"]  # ravel order
    keys = [("electron", "Te"), ("electron", "ne"), ("ion-1", "Ti"), ("general", "lam"), ("general", "amp1")]
    cfg = decks.deck_fit(active=active)
    B = 2
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=53)
    lf = LossFunction(cfg, sa, batch)
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    tp.X[:] = util.normed_to_matrix(normed, 1)
    hess = lf.h_loss_wrt_params(tp, batch)
    assert set(hess) == {"electron", "ion-1", "general"} and hess["electron"]["Te"]["general"]["lam"].shape == (B, B)
    for b in range(B):
        nb = {k: v[b:b + 1] for k, v in normed.items()}
        bb = {k: np.asarray(v)[b:b + 1] for k, v in batch.items()}
        Ho = ot.hessian(cfg, util.sa_fit(1), nb, bb, names)
        Hg = np.array([[hess[s1][k1][s2][k2][b, b] for (s2, k2) in keys] for (s1, k1) in keys])
        assert np.max(np.abs(Hg - Ho)) < 2e-4 * np.max(np.abs(Ho)), (b, Hg, Ho)
        # the quantity postprocess.get_sigmas derives from it
        sg = np.sign(np.diag(np.linalg.inv(Hg))) * np.sqrt(np.abs(np.diag(np.linalg.inv(Hg))))
        so = np.sign(np.diag(np.linalg.inv(Ho))) * np.sqrt(np.abs(np.diag(np.linalg.inv(Ho))))
        np.testing.assert_allclose(sg, so, rtol=5e-3)


def _fe2d(nv, kind):
    vx = orc.velocity_grid(nv)
    X, Y = np.meshgrid(vx, vx, indexing="ij")
    if kind == "maxwellian":
        f = np.exp(-(X**2 + Y**2) / 2)
    else:  # anisotropic super-Gaussian with a drifting bump: no symmetry left for a transposed index to hide behind
        f = np.exp(-((X / 1.3) ** 2 + (Y / 0.8) ** 2) ** 1.4 / 2) + 0.05 * np.exp(-((X - 2.0) ** 2 + (Y + 1.0) ** 2))
    return vx, f / (f.sum() * (vx[1] - vx[0]) ** 2)


@pytest.mark.parametrize("kind,nv", [("maxwellian", 48), ("anisotropic", 48), ("anisotropic", 132), ("anisotropic", 133)])
def test_form_factor_2d_matches_oracle(torch_mod, kind, nv):
    """a16: FormFactor.calc_in_2D (rotate + project + ratintn per (lambda, theta) point) vs the oracle's
    restatement on a subset of wavelengths; non-zero drift and flow at oblique angles.  (Parity with the reference
    itself is unpinned for this path: its goldens are not in the reference tree.)"""
    cfg = decks.deck_fit()
    B = 2  # nv = 48: table resident in LDS; nv = 132 (> 128): table read through L1/L2; 133: an odd number of samples per line (the sampler's loop is unrolled by two)
    sa = dict(sa=np.array([35.0, 60.0, 110.0]), weights=np.ones((B, 3)) / 3)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, B, seed=61, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    phys["ud"] = np.array([0.8, -1.1])
    X = util.normed_to_matrix(phys, 1)
    vx, fe2 = _fe2d(nv, kind)
    ud_ang, va_ang = 25.0, -40.0
    idx = np.array([0, 1, 100, 333, 511, 512, 700, 1023]) if nv < 100 else np.array([0, 400, 1023])
    for feature, rng in ((0, cfg["other"]["lamrangE"]), (1, cfg["other"]["lamrangI"])):
        P = eng.form_factor_2d(feature, X, fe2, ud_ang, va_ang).cpu().numpy()
        assert P.shape == (B, 1, 1024, 3) and np.all(np.isfinite(P))
        for b in range(B):
            p = orc.lineout_params(phys, b, 1)
            Po, _ = orc.form_factor_2d(rng, 1024, 0.0, sa["sa"], 1, p, vx, fe2, ud_ang, va_ang, lam_index=idx)
            err = np.max(np.abs(P[b][:, idx, :] - Po) / np.abs(Po))
            assert err < 1e-7, (feature, b, err)


def test_rolling_sampler_every_walk_direction(torch_mod):
    """The sampler for tables read through L1/L2 (nv = 132 > 128: project_rolling) is compiled in eight forms -- direction of the cell
    walk on each axis x orientation of the table copy it reads -- chosen per point from the rotation angle beta.  Away from the laser line
    beta is the direction of +-k (second / fourth quadrant for every scattering angle); within ~0.02 nm of it the drift term of xi_e wins,
    so drift directions in all four quadrants and samples of the ION window around the line make the points of this test cover all eight
    forms (asserted from the oracle's own beta); every point against the oracle's restatement of calc_in_2D, 1e-7."""
    cfg = decks.deck_fit()
    nv = 132
    sa = dict(sa=np.array([25.0, 40.0, 62.0, 88.0, 115.0, 150.0]), weights=np.ones((1, 6)) / 6)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, 1, seed=67, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    vx, fe2 = _fe2d(nv, "anisotropic")
    idx = np.array([3, 226, 231, 234, 235, 236, 239, 244, 1020])   # (the laser line, 526.094 nm, lies at sample 234.7 of the ion window)
    seen = set()
    for ud, ud_ang, va_ang in ((1.4, 25.0, -40.0), (1.4, 115.0, 200.0), (-1.2, 60.0, 10.0), (0.9, 290.0, 135.0)):
        phys["ud"] = np.array([ud])
        X = util.normed_to_matrix(phys, 1)
        P = eng.form_factor_2d(1, X, fe2, ud_ang, va_ang).cpu().numpy()
        dbg = {}
        Po, _ = orc.form_factor_2d(cfg["other"]["lamrangI"], 1024, 0.0, sa["sa"], 1, orc.lineout_params(phys, 0, 1), vx, fe2, ud_ang, va_ang,
                                   lam_index=idx, debug=dbg)
        err = np.max(np.abs(P[0][:, idx, :] - Po) / np.abs(Po))
        assert err < 1e-7, (ud, ud_ang, va_ang, err)
        cb, sb = np.cos(dbg["beta"]).ravel(), np.sin(dbg["beta"]).ravel()
        seen |= set(zip((cb >= 0).tolist(), (sb >= 0).tolist(), (np.abs(sb) > np.abs(cb)).tolist()))
    assert len(seen) == 8, sorted(seen)


@pytest.mark.parametrize("nv", [129, 133, 161, 191, 253])
def test_rolling_sampler_odd_table_sizes(torch_mod, nv):
    """The L1/L2 sampler issues its requests one sample ahead of the window, in a loop unrolled by two; an odd number of samples per line
    takes one more sample past the end of the line with weight zero.  (A first version finished odd lines in a separate tail: the values
    leaving the loop were copied before their data had arrived, and odd tables gave wrong projections two runs out of three -- an
    intermittent fault, so every size is run several times here.)  Forward and forward-with-records: the same bits every time, and the
    oracle's values."""
    cfg = decks.deck_fit()
    sa = dict(sa=np.array([35.0, 60.0, 110.0]), weights=np.ones((1, 3)) / 3)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, 1, seed=61, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    phys["ud"] = np.array([0.8])
    X = util.normed_to_matrix(phys, 1)
    vx, fe2 = _fe2d(nv, "anisotropic")
    idx = np.array([0, 400, 1023])
    Po, _ = orc.form_factor_2d(cfg["other"]["lamrangE"], 1024, 0.0, sa["sa"], 1, orc.lineout_params(phys, 0, 1), vx, fe2, 25.0, -40.0, lam_index=idx)
    P0 = None
    for rep in range(5):
        P = eng.form_factor_2d(0, X, fe2, 25.0, -40.0, save=bool(rep & 1))
        err = np.max(np.abs(P.cpu().numpy()[0][:, idx, :] - Po) / np.abs(Po))
        assert err < 1e-7, (nv, rep, err)
        if P0 is None:
            P0 = P.clone()
        assert bool((P == P0).all()), (nv, rep)


def _angular_sa(cfg):
    """tests/test_forward/test_angular_1v.py:53-60: the geometry is looked up as spectype "angular", then the deck is
    switched to "angular_full"."""
    from tsadar_amd import calibration

    cfg["other"]["extraoptions"]["spectype"] = "angular"
    sa = calibration.get_scattering_angles(cfg)
    cfg["other"]["extraoptions"]["spectype"] = "angular_full"
    sa["angAxis"] = calibration.angular_pixel_axis()
    return sa


@pytest.mark.parametrize("ccd,n_lam,start,end", [((1024, 1024), 1024, 90, 950), ((128, 256), 256, 10, 110)])
def test_ats_instrument_chain_matches_oracle(torch_mod, ccd, n_lam, start, end):
    """a16: angular_full weight-matrix product, add_ATS_IRF and reduce_ATS_to_resunit on the reference's own
    calibration data (1024 x 241 weight matrix, non-uniform pixel angle axis) vs the oracle's line-by-line
    restatement, full size and with 8 x 4 resolution units.  Both consume the same P[1, 1024, 241]."""
    from tsadar_amd import calibration

    cfg = decks.deck_angular(1, 64, ccd, start, end)
    sa = _angular_sa(cfg)
    assert sa["weights"].shape == (1024, 241) and sa["angAxis"].shape == (1024,)
    eng = _engine(cfg, sa, fe_mode=L.FE_PER_LINEOUT)
    normed = util.random_lineouts(cfg, 1, seed=5)
    phys = orc.physical_params(cfg["parameters"], normed, True)
    X = util.normed_to_matrix(phys, 1)
    P = eng.form_factor(0, X, orc.dlm_fe(2.7, 64)[None, :])[0]
    assert P.shape == (1, 1024, 241)
    lam_step, ang_step = 1024 // n_lam, 1024 // ccd[0]
    wid = cfg["other"]["PhysParams"]["widIRF"]
    eng.ats_setup(sa["weights"], sa["angAxis"], wid["spect_FWHM_ele"] / 2.3548, wid["ang_FWHM_ele"] / 2.3548,
                  lam_step, ang_step, start, end)
    rows = end - start
    e_amps = np.random.default_rng(3).uniform(0.5, 2.0, (rows, 1))
    p = orc.lineout_params(phys, 0, 1)
    E = eng.ats_spectrum(P, e_amps, p["lam"], p["amp1"], p["amp2"]).cpu().numpy()
    lam_nm = np.linspace(*cfg["other"]["lamrangE"], 1024)
    Eo, lam_o = orc.ats_spectrum(cfg, sa["weights"], sa["angAxis"], P.cpu().numpy(), lam_nm, n_lam, e_amps, p)
    assert E.shape == Eo.shape == (rows, n_lam)
    err = np.max(np.abs(E - Eo)) / np.max(np.abs(Eo))
    assert err < 1e-10, err  # 12-sigma tap cut-off: 5e-32 relative
    assert np.all(np.abs(E - Eo) <= 1e-9 * np.abs(Eo) + 1e-12)


@pytest.mark.parametrize("dim,fe_type", [(1, "dlm"), (2, "arbitrary"), (2, "sphericalharmonic")])
def test_angular_diagnostic_end_to_end(torch_mod, dim, fe_type):
    """ThomsonScatteringDiagnostic with spectype angular_full (tests/test_forward/test_angular_1v.py / _2v.py call
    pattern: batch=False parameters, e_amps = [1]) for a 1-D DLM and a 2-D Arbitrary2V distribution function; the
    1-D case is checked end to end against the oracle (form factor at 241 angles + instrument chain)."""
    from tsadar_amd import ThomsonParams, calibration
    from tsadar_amd.diagnostic import ThomsonScatteringDiagnostic

    cfg = decks.deck_angular(dim, 256 if dim == 1 else 64)
    if fe_type == "sphericalharmonic":  # tests/configs/arts2d_test_inputs.yaml:85-97
        cfg["parameters"]["electron"]["fe"] = {"active": True, "dim": 2, "type": "sphericalharmonic", "nvx": 64, "params": {
            "flm_type": "mora-yahi", "init_m": 2.2, "LTx": 225000.0, "LTy": 400000.0, "Nl": 1, "nvr": 64}}
    sa = _angular_sa(cfg)
    batch = dict(e_data=np.ones((1024, 1024)), i_data=np.ones((1024, 1024)), noise_e=np.array([0]), noise_i=np.array([0]),
                 e_amps=np.array([1]), i_amps=np.array([1]))
    diag = ThomsonScatteringDiagnostic(cfg, sa)
    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    E, I, lamE, lamI = diag(tp, batch)
    assert E.shape == (860, 1024) and np.all(np.isfinite(E)) and lamE.shape == (1024,)
    amp = tp.physical_matrix()[0, [L.P_AMP1, L.P_AMP2]]
    assert np.all(np.max(E, axis=1) <= np.max(amp) * (1 + 1e-12)) and np.all(np.max(E, axis=1) >= np.min(amp) * 0.5)
    if dim == 1:
        phys = orc.physical_params(cfg["parameters"], orc.init_normed_params(cfg["parameters"], 1, True), True)
        p = orc.lineout_params(phys, 0, 1)
        vx = orc.velocity_grid(256)
        fe = orc.dlm_fe(float(p["m"]), 256)
        Po, lam_cm = orc.form_factor(cfg["other"]["lamrangE"], 1024, 0.0, sa["sa"], 1, p, vx, fe)
        Eo, lam_o = orc.ats_spectrum(cfg, sa["weights"], sa["angAxis"], Po, np.squeeze(lam_cm) * 1e7, 1024,
                                     np.ones((860, 1)), p)
        assert np.max(np.abs(E - Eo)) / np.max(np.abs(Eo)) < 1e-7
        np.testing.assert_allclose(lamE, lam_o, rtol=1e-13)
    else:
        fe2 = tp()["electron"]["fe"]
        vx = tp()["electron"]["v"]
        assert fe2.shape == (64, 64) and abs(np.sum(fe2) * (vx[1] - vx[0]) ** 2 - 1.0) < 1e-12
        eng = diag.engine(True)
        P = eng.form_factor_2d(0, tp.physical_matrix(), fe2, 20.0, -35.0)[0].cpu().numpy()
        p = orc.lineout_params(orc.physical_params(cfg["parameters"], orc.init_normed_params(cfg["parameters"], 1, True), True), 0, 1)
        Eo, _ = orc.ats_spectrum(cfg, sa["weights"], sa["angAxis"], P, np.linspace(400, 700, 1024), 1024, np.ones((860, 1)), p)
        assert np.max(np.abs(E - Eo)) / np.max(np.abs(Eo)) < 1e-10


def _free_form_fe(B, nvx, seed):
    """Free-form distribution functions: super-Gaussians modulated by a smooth asymmetric factor, renormalised."""
    rng = np.random.default_rng(seed)
    vx = orc.velocity_grid(nvx)
    fes = []
    for b in range(B):
        f = orc.dlm_fe(rng.uniform(2.0, 3.5), nvx) * np.exp(0.15 * np.sin(1.3 * vx + rng.uniform(0, 6.28)) + 0.05 * np.tanh(vx))
        fes.append(f / np.sum(f) / (vx[1] - vx[0]))
    return np.stack(fes)


@pytest.mark.parametrize("nvx", [128, 64, 320])
def test_gradient_wrt_distribution_function(torch_mod, nvx):
    """SURVEY 8(f1), free-form f_e: d loss / d fe[b, i] from tsff_loss_grad_fe (table adjoints scattered in k_spectrum,
    transposed MFMA GEMM with the log-ratio table, k_fe_adjoint) vs reverse-mode autodiff of the oracle twin through
    both uses of f_e (Hermite ln f_e lookup and the 1640 x 1022 ratintn table), together with the plasma parameters."""
    from oracle import tsadar_oracle_torch as ot

    B = 2
    names = ["Te", "ne", "Ti_1", "Va", "lam", "amp1"]
    cfg = decks.deck_fit(nvx=nvx)
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=71)
    normed = util.random_lineouts(cfg, B, seed=73)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    fe = _free_form_fe(B, nvx, 5)
    eng = _engine(cfg, sa, fe_mode=L.FE_PER_LINEOUT)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms, grad, E, I, gfe = eng.loss_grad(util.normed_to_matrix(normed, 1), batch, w, eng.slots.active.astype(np.uint8),
                                           fe=fe, want_spectra=True, want_fe_grad=True)
    val, ref, ref_fe, Eo, Io = ot.value_and_grad_fe(cfg, sa, normed, batch, i_norm, e_norm, names, fe)
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8 and util.rel_err(I.cpu().numpy(), Io) < 1e-8
    assert abs(float(np.dot(terms.cpu().numpy(), w)) - val) < 1e-9 * abs(val)
    G = util.matrix_to_named(grad.cpu().numpy(), names)
    scale = max(np.max(np.abs(v)) for v in ref.values())
    for k in names:
        assert np.max(np.abs(G[k] - ref[k])) / scale < 1e-7, (k, G[k], ref[k])
    gfe = gfe.cpu().numpy()
    assert gfe.shape == (B, nvx) and np.all(np.isfinite(gfe))
    # compare as d loss / d ln fe (= fe * d loss / d fe): the tails of fe span 20 decades
    a, r = gfe * fe, ref_fe * fe
    err = np.max(np.abs(a - r)) / np.max(np.abs(r))
    assert err < 1e-7, err
    # and the plain gradient where fe is not negligible
    big = fe > 1e-6 * fe.max()
    assert np.max(np.abs(gfe[big] - ref_fe[big])) / np.max(np.abs(ref_fe[big])) < 1e-6


def test_vg_loss_free_form_distribution(torch_mod):
    """LossFunction.vg_loss with an Arbitrary1V distribution function (base.py:157-204; filter spec :462-471): the flat
    gradient carries nvx values per lineout right after (Te, ne), in the reference's ravel order, chained through the
    generator (Butterworth smoothing, 10 ** -(7 u)^2, normalisation) on the host."""
    from oracle import tsadar_oracle_torch as ot
    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd import distribution as D
    from tsadar_amd.loss_function import LossFunction

    B, nvx = 2, 64
    cfg = decks.deck_fit(nvx=nvx, active=("Te", "ne", "amp1", "lam"))
    cfg["parameters"]["electron"]["fe"] = {"active": True, "type": "arbitrary", "dim": 1, "nvx": nvx, "params": {"init_m": 2.4}}
    sa = util.sa_fit(B)
    cfg_data = decks.deck_fit(nvx=nvx)
    batch = util.synthetic_batch(cfg_data, sa, B, seed=81)
    loss_fn = LossFunction(cfg, sa, batch)
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    rng = np.random.default_rng(4)
    tp.fval = tp.fval * (1 + 0.02 * rng.normal(size=tp.fval.shape))
    tp.X[:, L.P_TE] += rng.normal(size=B) * 0.1
    spec = tree.get_filter_spec(cfg["parameters"], tp)
    assert [n for n, _ in spec] == [("electron", "Te"), ("electron", "ne"), ("electron", "fval"), ("general", "lam"), ("general", "amp1")]
    diff, static = tree.partition(tp, spec)
    x0, loss_fn.unravel_weights = tree.ravel_pytree(diff)
    assert x0.size == 4 * B + B * nvx
    val, g = loss_fn.vg_loss(x0, static, batch)
    # oracle: autodiff w.r.t. (Te, ne, lam, amp1) and f_e, then the generator's chain rule
    fe = D.arbitrary_1v(tp.fval)
    names = ["Te", "ne", "lam", "amp1"]
    cfg_o = decks.deck_fit(nvx=nvx, active=("Te", "ne", "amp1", "lam"))  # same activation flags, explicit f_e
    normed = {k: tp.X[:, util.slot_of(k)].copy() for k in orc.init_normed_params(cfg_o["parameters"], B, True)}
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    vo, ref, ref_fe, _, _ = ot.value_and_grad_fe(cfg_o, sa, normed, batch, i_norm, e_norm, names, fe)
    assert abs(val - vo) < 1e-9 * abs(vo)
    expect = np.concatenate([ref["Te"], ref["ne"], D.arbitrary_1v_vjp(tp.fval, ref_fe).ravel(), ref["lam"], ref["amp1"]])
    assert np.max(np.abs(g - expect)) / np.max(np.abs(expect)) < 1e-7
    # self-consistency: directional derivative of vg_loss's own value
    d = rng.normal(size=x0.size)
    d /= np.linalg.norm(d)
    h = 1e-6
    vp, _ = loss_fn.vg_loss(x0 + h * d, static, batch)
    vm, _ = loss_fn.vg_loss(x0 - h * d, static, batch)
    assert abs((vp - vm) / (2 * h) - np.dot(g, d)) < 1e-5 * np.linalg.norm(g)
    # one L-BFGS-B iteration lowers the loss
    import scipy.optimize as so

    res = so.minimize(loss_fn.vg_loss, x0, args=(static, batch), method="L-BFGS-B", jac=True, options={"maxiter": 3})
    assert res.fun < val


def test_adam_loop_like_reference(torch_mod):
    """The reference's optax branch (loops.py:59-95): vg_loss with a non-l-bfgs method returns ((value, aux), grad
    pytree); an Adam loop over DiffParams lowers the loss and the pytree gradient equals the flat one."""
    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd.loss_function import LossFunction

    B = 4
    cfg = decks.deck_fit()
    cfg["optimizer"]["method"] = "adam"
    cfg["optimizer"]["learning_rate"] = 0.02
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=17)
    loss_fn = LossFunction(cfg, sa, batch)
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
    opt = tree.Adam(cfg["optimizer"]["learning_rate"])
    state = opt.init(diff)
    losses = []
    for _ in range(25):
        (val, aux), grad = loss_fn.vg_loss(diff, static, batch)
        assert isinstance(grad, tree.DiffParams) and aux[0].shape == (B, 1024) and "electron" in aux[1]
        updates, state = opt.update(grad, state)
        diff = tree.apply_updates(diff, updates)
        losses.append(val)
    assert losses[-1] < 0.85 * losses[0] and all(b < a * 1.02 for a, b in zip(losses, losses[1:])), losses
    # same numbers through the l-bfgs-b calling convention
    cfg2 = copy.deepcopy(cfg)
    cfg2["optimizer"]["method"] = "l-bfgs-b"
    lf2 = LossFunction(cfg2, sa, batch)
    x0, lf2.unravel_weights = tree.ravel_pytree(diff)
    v2, g2 = lf2.vg_loss(x0, static, batch)
    (v1, _), g1 = loss_fn.vg_loss(diff, static, batch)
    assert v1 == v2 and np.array_equal(g1.ravel(), g2)


def _dist_rank_2d(rank, world, port, out):
    """One rank of the 2-process rehearsal of the sharded 2-D form factor (shared GPU, gloo)."""
    import os
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TSFF_DIST_BACKEND="gloo", TSFF_FORCE_DEVICE="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    from tsadar_amd import ThomsonParams, distributed as D
    from tsadar_amd.engine import Engine

    D.init_from_env()
    cfg = decks.deck_fit()
    sa = dict(sa=np.array([35.0, 60.0, 110.0]), weights=np.ones((1, 3)) / 3)
    eng = Engine(cfg, sa, activate=False)
    tp = ThomsonParams(cfg["parameters"], 1, batch=True, activate=False)
    vx, fe2 = _fe2d(48, "anisotropic")
    P = D.form_factor_2d_sharded(eng, 0, tp.physical_matrix(), fe2, 25.0, -40.0, world, rank)
    ref = eng.form_factor_2d(0, tp.physical_matrix(), fe2, 25.0, -40.0)
    assert P.shape == ref.shape == (1, 1, 1024, 3)
    np.save(os.path.join(out, f"p{rank}.npy"), np.stack([P.cpu().numpy(), ref.cpu().numpy()]))
    # the adjoint over the same split: each rank reverses its slice, one all-reduce of [grad_phys | grad_fe2d]
    import torch

    Pbar = torch.as_tensor(np.random.default_rng(12).standard_normal(tuple(ref.shape)), device=ref.device) / ref.abs().mean()
    gp, gf = D.form_factor_2d_grad_sharded(eng, 0, tp.physical_matrix(), fe2, Pbar, 25.0, -40.0, world, rank)
    gp1, gf1 = eng.form_factor_2d_grad(0, tp.physical_matrix(), fe2, Pbar, 25.0, -40.0)
    np.savez(os.path.join(out, f"g{rank}.npz"), gp=gp.cpu().numpy(), gf=gf.cpu().numpy(), gp1=gp1.cpu().numpy(), gf1=gf1.cpu().numpy())
    # the whole angular fit step on two ranks: LossFunction(distributed=True) shards the point list of the 2-D form factor
    # (all-gather) and of its adjoint (all-reduce); value and gradient equal the single-rank ones
    from tsadar_amd import tree
    from tsadar_amd.loss_function import LossFunction

    acfg = decks.deck_angular(2, 48, (128, 256), 10, 110)
    asa = _angular_sa(acfg)
    atp = ThomsonParams(acfg["parameters"], 1, batch=False, activate=True)
    abatch = dict(e_data=np.ones((100, 256)), i_data=np.zeros((100, 256)), e_amps=np.ones((100, 1)), i_amps=np.zeros(100),
                  noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    single = LossFunction(acfg, asa, abatch)
    truth = atp.copy()
    truth.X[0, 0] -= 0.3
    abatch["e_data"] = single.ts_diag(truth, abatch)[0]
    single = LossFunction(acfg, asa, abatch)
    multi = LossFunction(acfg, asa, abatch, distributed=True)
    adiff, astatic = tree.partition(atp, tree.get_filter_spec(acfg["parameters"], atp))
    x0, single.unravel_weights = tree.ravel_pytree(adiff)
    multi.unravel_weights = single.unravel_weights
    v1, g1 = single.vg_loss(x0, astatic, abatch)
    v2, g2 = multi.vg_loss(x0, astatic, abatch)
    np.savez(os.path.join(out, f"a{rank}.npz"), v1=v1, g1=g1, v2=v2, g2=g2)
    dist.destroy_process_group()


def test_two_rank_sharded_form_factor_2d(torch_mod, tmp_path):
    """SURVEY 8(e), 2-D angular row: the flat (lambda, theta) point list split over two ranks
    (tsff_form_factor_2d_range) and all-gathered equals the single-rank image bit for bit, on both ranks; odd point
    counts exercise the padded last chunk (point_range)."""
    import socket
    import torch.multiprocessing as mp

    from tsadar_amd import distributed as D

    assert D.point_range(3073, 2, 0) == (0, 1537, 1537) and D.point_range(3073, 2, 1) == (1537, 3073, 1537)
    assert D.point_range(5, 8, 7) == (5, 5, 1)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dist_rank_2d, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    np.testing.assert_array_equal(p0[0], p0[1])
    np.testing.assert_array_equal(p1[0], p1[1])
    np.testing.assert_array_equal(p0[0], p1[0])
    g0, g1 = np.load(tmp_path / "g0.npz"), np.load(tmp_path / "g1.npz")
    a0, a1 = np.load(tmp_path / "a0.npz"), np.load(tmp_path / "a1.npz")
    assert a0["v2"] == a1["v2"] == a0["v1"] and np.array_equal(a0["g2"], a1["g2"])   # forward bit-identical, ranks agree
    assert a0["g1"].size == 48 * 48 + 5 and np.max(np.abs(a0["g2"] - a0["g1"])) < 1e-12 * np.max(np.abs(a0["g1"]))
    for k in ("gp", "gf"):   # sums in a different order: equal to rounding (1e-13 of the largest entry), identical on both ranks
        np.testing.assert_array_equal(g0[k], g1[k])
        assert np.max(np.abs(g0[k] - g0[k + "1"])) < 1e-13 * np.max(np.abs(g0[k + "1"])), k


def _random_deck(seed):
    """A randomly configured deck: ion species, gradient points, points per pixel, velocity grid, wavelength windows,
    notch filter, IRF widths, fit ranges, loss functional, drifts -- everything the static configuration can vary."""
    rng = np.random.default_rng(1000 + seed)
    n_ion = int(rng.integers(1, 3))
    active = ["Te", "ne", "Ti", "lam", "amp1", "amp2", "amp3", "Va", "ud"]
    if rng.random() < 0.5:
        active += ["Z"]
    G = int(rng.integers(1, 4))
    if G > 1:
        active += ["Te_gradient", "ne_gradient"]
    cfg = decks.deck_fit(points_per_pixel=int(rng.integers(1, 3)), nvx=int(rng.choice([64, 96, 128, 200])),
                         m=float(rng.uniform(2.0, 4.5)), active=tuple(active), n_ion=n_ion)
    g = cfg["parameters"]["general"]
    g["Te_gradient"].update(val=float(rng.uniform(0, 8)), num_grad_points=G)
    g["ne_gradient"].update(val=float(rng.uniform(0, 12)), num_grad_points=G)
    g["ud"]["val"] = float(rng.uniform(-2, 2))
    g["Va"]["val"] = float(rng.uniform(-3, 3))
    r = cfg["data"]["fit_rng"]
    r["forward_epw_start"], r["forward_epw_end"] = float(rng.uniform(380, 430)), float(rng.uniform(640, 720))
    r["blue_min"], r["blue_max"] = float(rng.uniform(440, 470)), float(rng.uniform(495, 515))
    r["red_min"], r["red_max"] = float(rng.uniform(535, 560)), float(rng.uniform(600, 630))
    other = cfg["other"]
    other["iawfilter"] = [int(rng.integers(0, 2)), float(rng.uniform(2, 5)), float(rng.uniform(10, 30)), float(rng.uniform(524, 530))]
    other["PhysParams"]["widIRF"] = {"spect_stddev_ele": float(rng.uniform(0.6, 2.0)), "spect_stddev_ion": float(rng.uniform(0.008, 0.03))}
    ext = other["extraoptions"]
    ext["fit_EPWb"], ext["fit_EPWr"] = bool(rng.random() < 0.8), True
    cfg["data"]["ion_loss_scale"] = float(rng.uniform(0.3, 3.0))
    cfg["data"]["ele_lam_shift"] = float(rng.uniform(-0.5, 0.5))
    cfg["optimizer"]["loss_method"] = str(rng.choice(["l2", "l1", "log-cosh", "poisson"]))
    cfg["optimizer"]["y_norm"] = bool(rng.random() < 0.7)
    if n_ion == 2:
        cfg["parameters"]["ion-2"]["Ti"]["same"] = bool(rng.random() < 0.5)
    return decks.finish(cfg), n_ion


@pytest.mark.parametrize("seed", range(10))
def test_random_decks_forward_loss_gradient(torch_mod, seed):
    """Randomly configured decks (species, gradient points, points per pixel, nvx, windows, notch filter, IRF widths, fit
    ranges, loss functional, noise arrays): spectra vs the NumPy oracle, loss and gradient vs autodiff of its twin."""
    from oracle import tsadar_oracle_torch as ot

    cfg, n_ion = _random_deck(seed)
    B = 2
    sa = util.sa_fit(B)
    rng = np.random.default_rng(seed)
    if seed >= 5:   # not the P9 fibre bundle: a random set of scattering angles and weights
        na = int(np.random.default_rng(900 + seed).integers(3, 15))
        ang = np.sort(np.random.default_rng(901 + seed).uniform(25.0, 140.0, na))
        wts = np.random.default_rng(902 + seed).uniform(0.2, 1.0, na)
        sa = dict(sa=ang, weights=(wts / wts.sum()) * np.ones([B, na]))
    batch = util.synthetic_batch(cfg, sa, B, seed=300 + seed)
    batch["noise_e"] = 0.02 * rng.random((B, 1024))
    batch["noise_i"] = 0.02 * rng.random((B, 1024))
    normed = util.random_lineouts(cfg, B, seed=400 + seed, ranges=dict(ud=(-2, 2), Ti_2=(0.05, 0.5)))
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    eng = _engine(cfg, sa)
    X = util.normed_to_matrix(normed, n_ion)
    Eo, Io, _, _ = orc.ts_diag(cfg, sa, normed, batch)
    E, I = eng.forward(X, batch["e_amps"], batch["i_amps"], batch["noise_e"], batch["noise_i"])
    # (1e-8 on the P9 fibre bundle; the random angle sets reach down to 25 degrees, where k lambda_De is small and the resonances
    #  are sharp enough for last-bit differences in k to show at 3e-8: seeds 150, 154, 159, 183 of a 200-seed soak)
    tol, ltol = (1e-8, 1e-9) if seed < 5 else (5e-8, 1e-8)
    assert util.rel_err(E.cpu().numpy(), Eo) < tol and util.rel_err(I.cpu().numpy(), Io) < tol, cfg["optimizer"]
    names = [k for k in normed if eng.slots.active[util.slot_of(k)]]
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms, grad, _, _ = eng.loss_grad(X, batch, w, eng.slots.active.astype(np.uint8))
    val, ref, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    assert abs(float(np.dot(terms.cpu().numpy(), w)) - val) < ltol * abs(val), (cfg["optimizer"]["loss_method"], val)
    Gd = util.matrix_to_named(grad.cpu().numpy(), names)
    scale = max(np.max(np.abs(v)) for v in ref.values())
    for k in names:
        assert np.max(np.abs(Gd[k] - ref[k])) / scale < 1e-6, (k, cfg["optimizer"]["loss_method"], Gd[k], ref[k])


def test_angular_vg_loss_finite_difference(torch_mod):
    """LossFunction for spectype angular_full with a 1-D DLM distribution function: value = the reference's calc_ei_error
    on the ARTS image (oracle chain end to end), gradient = the hand-written adjoint (loss seed -> tsff_ats_adjoint ->
    tsff_form_factor_grad -> parameter transform; the DLM order through d loss / d f_e) vs central differences of the
    ORACLE's loss; the finite-difference fallback agrees; an L-BFGS-B run recovers perturbed parameters."""
    from scipy.optimize import minimize

    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd.loss_function import LossFunction

    cfg = decks.deck_angular(1, 64, (128, 256), 10, 110)
    for k in ("amp1", "amp2", "lam"):
        cfg["parameters"]["general"][k]["active"] = False
    sa = _angular_sa(cfg)
    vx = orc.velocity_grid(64)

    def oracle_image(normed, e_amps):
        phys = orc.physical_params(cfg["parameters"], normed, True)
        p = orc.lineout_params(phys, 0, 1)
        Po, lam_cm = orc.form_factor(cfg["other"]["lamrangE"], 1024, 0.0, sa["sa"], 1, p, vx, orc.dlm_fe(float(p["m"]), 64))
        return orc.ats_spectrum(cfg, sa["weights"], sa["angAxis"], Po, np.squeeze(lam_cm) * 1e7, 256, e_amps, p)

    truth = orc.init_normed_params(cfg["parameters"], 1, True)
    truth["Te"] = truth["Te"] - 0.4
    truth["ne"] = truth["ne"] + 0.3
    truth["m"] = truth["m"] + 0.5
    data, lam = oracle_image(truth, np.ones((100, 1)))
    batch = dict(e_data=data, i_data=np.zeros((100, 256)), e_amps=np.ones((100, 1)), i_amps=np.zeros(100),
                 noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    loss_fn = LossFunction(cfg, sa, batch)
    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    spec = tree.get_filter_spec(cfg["parameters"], tp)
    assert [n for n, _ in spec] == [("electron", "Te"), ("electron", "ne"), ("electron", "m")]
    diff, static = tree.partition(tp, spec)
    x0, loss_fn.unravel_weights = tree.ravel_pytree(diff)
    val, g = loss_fn.vg_loss(x0, static, batch)

    def oracle_loss(x):
        n = orc.init_normed_params(cfg["parameters"], 1, True)
        n["Te"], n["ne"], n["m"] = np.array([x[0]]), np.array([x[1]]), np.array([x[2]])
        E, lam_o = oracle_image(n, batch["e_amps"])
        err = np.square(data - E) / loss_fn.e_norm**2
        r = cfg["data"]["fit_rng"]
        blue = (lam_o > r["blue_min"]) & (lam_o < r["blue_max"])
        red = (lam_o > r["red_min"]) & (lam_o < r["red_max"])
        return 0.5 * (np.mean(err[:, blue]) + np.mean(err[:, red]))

    vo = oracle_loss(x0)
    assert abs(val - vo) < 1e-7 * abs(vo), (val, vo)
    h = 1e-6
    go = np.array([(oracle_loss(x0 + h * e) - oracle_loss(x0 - h * e)) / (2 * h) for e in np.eye(3)])
    assert np.max(np.abs(g - go)) < 1e-4 * np.max(np.abs(go)), (g, go)
    loss_fn.force_fd = True
    v2, g2 = loss_fn.vg_loss(x0, static, batch)
    loss_fn.force_fd = False
    # (step 1e-5 of the fallback straddles kinks of the table lookups: it is the less accurate of the two)
    # (value: host NumPy sums on the fallback path, device sums on the adjoint path)
    assert abs(v2 - val) < 1e-13 * abs(val) and np.max(np.abs(g2 - g)) < 1e-2 * np.max(np.abs(g)), (g2, g)
    # ---- L-BFGS-B on two parameters (Te, ne; the DLM order held at its true value), noise-free data ----
    # The image loss of the reference's angular path (row-max normalisation, resolution-unit means, table lookups) is
    # extremely ill-conditioned: its gradient changes by 0.6 % over 1e-6 in x a few steps from the start (difference
    # quotients with h = 1e-5 are off by 8 % there), the (Te, ne) plane holds several local minima within 0.02 of the truth,
    # and the line search jumps across them -- driven by the ORACLE's loss with central differences the same start ends at
    # 0.16 of the initial loss (h = 1e-6) while the adjoint-driven run ends at 0.035: which minimum is reached depends on the
    # last bits (tests/golden/make_angular_lbfgs.py has the study).  So the check is on what IS determined: (i) the first
    # iterates equal the oracle-driven ones (committed fixture, central differences with h = 1e-7) to 1e-6, before the
    # sensitivity amplifies rounding by ~500x per iteration; (ii) the run ends at a stationary point of the ORACLE's loss
    # with a lower value -- i.e. it stopped because the model has a minimum there, not because of a wrong gradient.
    cfg["parameters"]["electron"]["fe"]["active"] = False
    fix = np.load("tests/golden/angular_lbfgs_oracle.npz")
    truth2 = orc.init_normed_params(cfg["parameters"], 1, True)
    truth2["Te"] = truth2["Te"] - 0.2
    truth2["ne"] = truth2["ne"] + 0.12
    np.testing.assert_allclose([truth2["Te"][0], truth2["ne"][0]], fix["x_truth"], rtol=0, atol=1e-14)
    data2 = oracle_image(truth2, np.ones((100, 1)))[0]
    batch2 = dict(batch, e_data=data2)
    fit_fn = LossFunction(cfg, sa, batch2)
    tp2 = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    diff2, static2 = tree.partition(tp2, tree.get_filter_spec(cfg["parameters"], tp2))
    x2, fit_fn.unravel_weights = tree.ravel_pytree(diff2)
    assert x2.size == 2 and np.allclose(x2, fix["x_start"], rtol=0, atol=1e-14)
    its = []

    def vg(x, *a):
        v, g = fit_fn.vg_loss(x, *a)
        its.append(np.concatenate([x, [v], g]))
        return v, g

    res = minimize(vg, x2, args=(static2, batch2), method="L-BFGS-B", jac=True, options={"maxiter": 60, "ftol": 1e-15, "gtol": 1e-12})
    its, ref = np.array(its), fix["iterates"]
    v0, g0 = its[0, 2], np.max(np.abs(its[0, 3:]))
    for k in range(3):   # (i)
        assert np.max(np.abs(its[k, :2] - ref[k, :2])) < 1e-6, (k, its[k], ref[k])
        assert abs(its[k, 2] - ref[k, 2]) < 1e-3 * ref[k, 2], (k, its[k], ref[k])
    assert np.max(np.abs(its[0, 3:] - ref[0, 3:])) < 1e-6 * g0 and np.max(np.abs(its[1, 3:] - ref[1, 3:])) < 1e-3 * np.max(np.abs(ref[1, 3:]))

    def oracle_loss2(x):
        n = orc.init_normed_params(cfg["parameters"], 1, True)
        n["Te"], n["ne"] = np.array([x[0]]), np.array([x[1]])
        E, lam_o = oracle_image(n, batch["e_amps"])
        err = np.square(data2 - E) / fit_fn.e_norm**2
        r = cfg["data"]["fit_rng"]
        return 0.5 * (np.mean(err[:, (lam_o > r["blue_min"]) & (lam_o < r["blue_max"])]) + np.mean(err[:, (lam_o > r["red_min"]) & (lam_o < r["red_max"])]))

    h = 1e-7   # (ii)
    g_end = np.array([(oracle_loss2(res.x + h * e) - oracle_loss2(res.x - h * e)) / (2 * h) for e in np.eye(2)])
    assert abs(oracle_loss2(res.x) - res.fun) < 1e-7 * res.fun
    assert np.max(np.abs(g_end)) < 1e-4 * g0 and np.max(np.abs(res.jac)) < 1e-4 * g0, (res.x, g_end, res.jac, g0)
    assert res.fun < 0.5 * v0, (res.x, res.fun, v0, res.nit)


@pytest.mark.parametrize("B", [37, 256, 4096])
def test_full_batch_against_cpp_oracle(torch_mod, B):
    """BASELINE config 2's and config 3's batches (256 / 4096 lineouts) checked one by one -- spectra, loss sums and all six gradient columns --
    against the C++/OpenMP oracle (forward-mode dual numbers, pinned to the reference golden vector in
    tests/test_oracle_c.py): no sampling, every lineout of the batch."""
    from oracle import c_oracle as co
    from tsadar_amd import synthetic as S

    cfg = S.baseline_deck(batch_size=B)
    sa = util.sa_fit(B)
    eng = _engine(cfg, sa)
    rng = np.random.default_rng(S.SEED)
    truth = S.draw_params(cfg, B, rng)
    batch = S.make_batch(eng, truth, rng)
    guess = S.draw_params(cfg, B, rng)
    X = guess.to_matrix()
    hb = {k: (v.cpu().numpy() if v is not None else None) for k, v in batch.items()}
    w = eng.loss_weights(B, float(hb["i_data"].max()), float(hb["e_data"].max()), cfg["data"]["ion_loss_scale"])
    gm = guess.grad_mask()
    terms, grad, E, I = eng.loss_grad(X, batch, w, gm, want_spectra=True)
    sums, gref, Eo, Io = co.loss_grad(cfg, sa, X, hb, w=w, gmask=gm)
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8 and util.rel_err(I.cpu().numpy(), Io) < 1e-7
    np.testing.assert_allclose(terms.cpu().numpy(), sums.sum(axis=0), rtol=1e-9)
    g = grad.cpu().numpy()
    act = np.nonzero(gm)[0]
    for s in act:  # per column: relative to the column's largest entry, every lineout
        assert np.max(np.abs(g[:, s] - gref[:, s])) < 1e-6 * np.max(np.abs(gref[:, s])), s
    assert np.all(g[:, gm == 0] == 0.0)
    # tsff_forward on the same batch (BASELINE configs[1] is B = 256 forward-only): the pair-sweep forward kernel in its 512-thread
    # one-round form (B = 37, 256) and with three workgroups per CU (B = 4096) -- the bits of the loss kernel's spectra, hence the
    # oracle's to the same 1e-8 / 1e-7, every lineout
    Ef, If = eng.forward(X, batch["e_amps"], batch["i_amps"], noise_e=batch.get("noise_e"), noise_i=batch.get("noise_i"))
    assert bool((Ef == E).all()) and bool((If == I).all())


def test_forward_pass_like_calc_series(torch_mod):
    """tsadar.forward.calc_series.forward_pass, compute half: the 1-D deck of the reference's forward test gives the
    golden vector (ThomsonParams without activation here, as forward_pass builds them: activate defaults to False)."""
    from tsadar_amd.forward import forward_spectra

    cfg = decks.deck_1d()
    for k in ("lamrangE", "lamrangI", "npts"):
        cfg["other"].pop(k, None)
    res = forward_spectra(cfg)
    assert res["ThryE"].shape == (1, 1, 1024) and res["lamAxisE"].shape == (1, 1, 1024) and res["spectrum_calc_time"] > 0
    assert cfg["other"]["npts"] == 5120 and cfg["other"]["lamrangE"] == [400, 700]
    # oracle with the same (non-activated) parameters
    normed = orc.init_normed_params(cfg["parameters"], 1, False)
    sa = dict(sa=util.P9["sa"], weights=util.P9["weights"])
    unit = dict(e_amps=np.array([1.0]), i_amps=np.array([1.0]), noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    Eo, _, lamE, _ = orc.ts_diag(cfg, sa, normed, unit, activate=False)
    assert util.rel_err(res["ThryE"][0], Eo) < 1e-8
    np.testing.assert_allclose(res["lamAxisE"][0], lamE, rtol=1e-12)


def test_nan_outside_fit_ranges_is_ignored(torch_mod):
    """Measured data outside every fit range never enter the loss (loss_function.py:224-259 masks them to NaN and
    nanmean drops them); NaNs stored there must not leak into the loss or the gradient."""
    cfg = decks.deck_fit()
    B = 3
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=5)
    eng = _engine(cfg, sa)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    X = util.normed_to_matrix(normed, 1)
    gm = eng.slots.active.astype(np.uint8)
    t0, g0, _, _ = eng.loss_grad(X, batch, w, gm)
    bad = {k: (np.array(v, dtype=np.float64, copy=True) if k in ("e_data", "i_data") else v) for k, v in batch.items()}
    bad["e_data"][:, eng.mask_ele == 0] = np.nan
    bad["i_data"][:, eng.mask_ion == 0] = np.nan
    t1, g1, _, _ = eng.loss_grad(X, bad, w, gm)
    np.testing.assert_array_equal(t0.cpu().numpy(), t1.cpu().numpy())
    np.testing.assert_array_equal(g0.cpu().numpy(), g1.cpu().numpy())


def test_launch_plans_agree(torch_mod):
    """The launch plans of tsff_loss_grad give the same numbers: interleaved (one 256-thread workgroup per lineout and
    feature, default; the one-sweep kernel k_spectrum_fused), both features in one 512-thread workgroup (two-sweep
    k_spectrum), the two-sweep kernel interleaved, and -- with 5 points per pixel, where two
    features do not fit the LDS of a CU -- one launch per feature accumulating into the gradient (checked against the
    C++ oracle)."""
    from oracle import c_oracle as co

    cfg = decks.deck_fit()
    B = 5
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=77)
    eng = _engine(cfg, sa)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    X = util.normed_to_matrix(normed, 1)
    gm = eng.slots.active.astype(np.uint8)
    out = {}
    for plan in (0, 1, 2, 3):   # bit 0: never interleave the features; bit 1: two-sweep kernel instead of the one-sweep one
        eng.set_launch_plan(plan)
        t, g, E, I = eng.loss_grad(X, batch, w, gm, want_spectra=True)
        out[plan] = [a.cpu().numpy() for a in (t, g, E, I)]
    eng.set_launch_plan(0)
    Ef, If = eng.forward(X, batch["e_amps"], batch["i_amps"], batch["noise_e"], batch["noise_i"])
    for plan in (1, 2, 3):
        np.testing.assert_array_equal(out[0][2], out[plan][2])   # spectra: the same bits from every kernel
        np.testing.assert_array_equal(out[0][3], out[plan][3])
        np.testing.assert_array_equal(out[0][0], out[plan][0])
        np.testing.assert_allclose(out[0][1], out[plan][1], rtol=1e-12, atol=1e-15 * np.abs(out[0][1]).max())
    np.testing.assert_array_equal(out[0][2], Ef.cpu().numpy())   # ... and from tsff_forward
    np.testing.assert_array_equal(out[0][3], If.cpu().numpy())
    # 5 points per pixel, both features: split launches
    cfg5 = decks.deck_fit(points_per_pixel=5)
    sa5, batch5, normed5, i5, e5 = _loss_setup(cfg5, 2, seed=78)
    eng5 = _engine(cfg5, sa5)
    w5 = eng5.loss_weights(2, i5, e5, cfg5["data"]["ion_loss_scale"])
    X5 = util.normed_to_matrix(normed5, 1)
    t5, g5, E5, I5 = eng5.loss_grad(X5, batch5, w5, gm, want_spectra=True)
    sums, gref, Eo, Io = co.loss_grad(cfg5, sa5, X5, batch5, w=w5, gmask=gm)
    assert util.rel_err(E5.cpu().numpy(), Eo) < 1e-8 and util.rel_err(I5.cpu().numpy(), Io) < 1e-7
    np.testing.assert_allclose(t5.cpu().numpy(), sums.sum(axis=0), rtol=1e-9)
    g5 = g5.cpu().numpy()
    for s in np.nonzero(gm)[0]:
        assert np.max(np.abs(g5[:, s] - gref[:, s])) < 1e-6 * np.max(np.abs(gref[:, s])), s


@pytest.mark.parametrize("ppp,n_ion,active", [(5, 1, ("Te", "ne", "Ti", "Va", "lam", "amp1")), (2, 2, ("Te", "ne", "Ti", "Z", "Va", "ud", "lam", "amp1", "amp2", "amp3")),
                                              (3, 1, ("Te", "ne", "m", "amp1", "amp2", "lam")), (4, 1, ("Te", "ne", "Ti", "lam", "amp3")),
                                              (6, 1, ("Te", "ne", "Ti", "Va", "lam", "amp1"))])
def test_rows_kernel_points_per_pixel(torch_mod, ppp, n_ion, active):
    """k_spectrum_rows (points_per_pixel > 1: the one-sweep kernel with its Jacobian rows in a global scratch array) against
    the two-sweep kernel on the same deck -- spectra and loss sums the same bits, gradient to 1e-11 -- and against the C++
    oracle; 5 points per pixel is the reference's default deck shape (tests/configs/1d-defaults.yaml:100); the third case
    fits the DLM order per lineout (tangent tables, GM = 1); 6 points per pixel takes the kernel's generic convolution forms
    (the fast ones are instantiated for 2 to 5) if two workgroups still fit a CU, else the two-sweep kernel on both sides."""
    from oracle import c_oracle as co

    dlm = "m" in active
    cfg = decks.deck_fit(points_per_pixel=ppp, active=active, n_ion=n_ion, m=2.6 if dlm else 2.0)
    B = 3
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=500 + ppp)
    rng = np.random.default_rng(510 + ppp)
    batch["noise_e"] = 0.02 * rng.random((B, 1024))
    batch["noise_i"] = 0.02 * rng.random((B, 1024))
    eng = _engine(cfg, sa)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    X = util.normed_to_matrix(normed, n_ion)
    gm = eng.slots.active.astype(np.uint8)
    out = {}
    # rows kernel (small batch: one workgroup per round, the last to finish runs the chain) / without the base-point exchange
    # between lanes / one workgroup for all rounds (the large-batch form) / two-sweep kernel
    for plan in (0, 8, 1, 2):
        eng.set_launch_plan(plan)
        out[plan] = [a.cpu().numpy() for a in eng.loss_grad(X, batch, w, gm, want_spectra=True)]
    eng.set_launch_plan(0)
    for k in range(4):
        np.testing.assert_array_equal(out[0][k], out[8][k])
        np.testing.assert_array_equal(out[0][k], out[1][k])
    for rep in range(3):   # (the split form hands rows between workgroups: the same bits every time)
        again = [a.cpu().numpy() for a in eng.loss_grad(X, batch, w, gm, want_spectra=True)]
        for k in range(4):
            np.testing.assert_array_equal(out[0][k], again[k])
    for k in (2, 3):
        np.testing.assert_array_equal(out[0][k], out[2][k])
    # tsff_forward at several points per pixel: the forward form of the rounds kernel (k_spectrum_rows<..., FWD>) in its small-batch
    # form (plan 0: one workgroup per round), as one workgroup for all rounds (plan 1) and k_spectrum MODE 0 (plan 2) -- the bits of the
    # loss kernels' spectra, every time
    for plan in (0, 0, 1, 2):
        eng.set_launch_plan(plan)
        Ef, If = eng.forward(X, batch["e_amps"], batch["i_amps"], noise_e=batch["noise_e"], noise_i=batch["noise_i"], fe=None)
        np.testing.assert_array_equal(Ef.cpu().numpy(), out[0][2])
        np.testing.assert_array_equal(If.cpu().numpy(), out[0][3])
    eng.set_launch_plan(0)
    # (the loss sums are folded over 256 threads x 4 bins here, over 512 x 2 by the two-sweep kernel's one-feature workgroups)
    np.testing.assert_allclose(out[0][0], out[2][0], rtol=1e-14)
    np.testing.assert_allclose(out[0][1], out[2][1], rtol=1e-10, atol=1e-12 * np.abs(out[0][1]).max())   # (sums of 10^4 terms in another order)
    assert np.abs(out[0][1]).max() > 0.0
    if not dlm:
        sums, gref, Eo, Io = co.loss_grad(cfg, sa, X, batch, w=w, gmask=gm)
        assert util.rel_err(out[0][2], Eo) < 1e-8 and util.rel_err(out[0][3], Io) < 1e-7
        np.testing.assert_allclose(out[0][0], sums.sum(axis=0), rtol=1e-9)
        for sl in np.nonzero(gm)[0]:
            assert np.max(np.abs(out[0][1][:, sl] - gref[:, sl])) <= 1e-6 * np.max(np.abs(gref)), sl


@pytest.mark.parametrize("ele,ion", [(True, False), (False, True)])
def test_rows_kernel_single_feature(torch_mod, ele, ion):
    """k_spectrum_rows with ONE loaded feature (grid B, gradient written directly instead of through the per-feature parts),
    3 points per pixel: loss and gradient against the two-sweep kernel and the oracle's loss."""
    cfg = decks.deck_fit(points_per_pixel=3)
    ext = cfg["other"]["extraoptions"]
    ext["load_ele_spec"], ext["load_ion_spec"] = ele, ion
    ext["fit_EPWb"] = ext["fit_EPWr"] = ele
    ext["fit_IAW"] = ion
    B = 3
    sa, batch, normed, i_norm, e_norm = _loss_setup(decks.deck_fit(points_per_pixel=3), B, seed=61)
    eng = _engine(cfg, sa)
    X = util.normed_to_matrix(normed, 1)
    w = eng.loss_weights(B, i_norm, e_norm)
    gm = eng.slots.active.astype(np.uint8)
    out = {}
    for plan in (0, 2):
        eng.set_launch_plan(plan)
        out[plan] = [a.cpu().numpy() for a in eng.loss_grad(X, batch, w, gm, want_spectra=True)]
    eng.set_launch_plan(0)
    for k in (2, 3):
        np.testing.assert_array_equal(out[0][k], out[2][k])
    np.testing.assert_allclose(out[0][0], out[2][0], rtol=1e-14)
    np.testing.assert_allclose(out[0][1], out[2][1], rtol=1e-10, atol=1e-12 * np.abs(out[0][1]).max())
    assert np.abs(out[0][1]).max() > 0.0
    lo, _, _ = orc.loss(cfg, sa, normed, batch, i_norm, e_norm)
    assert abs(float(np.dot(out[0][0], w)) - lo) < 1e-9 * abs(lo)


def test_wide_irf_cuts_taps_instead_of_failing(torch_mod):
    """5 points per pixel with an ion IRF five times as wide as the shipped decks': at 12 sigma the spectrum + halo + taps outgrow the LDS
    of a CU.  The engine drops the outermost taps step by step (never below 7 sigma) with a warning instead of refusing the deck;
    the result equals the oracle's full-length convolution to the weight of the dropped taps."""
    import warnings

    cfg = decks.deck_fit(points_per_pixel=5)
    cfg["other"]["PhysParams"]["widIRF"]["spect_stddev_ion"] = 0.075
    B = 2
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=71)
    with warnings.catch_warnings(record=True) as wrn:
        warnings.simplefilter("always")
        eng = _engine(cfg, sa)
    assert 7.0 <= eng.irf_cutoff_sigmas < 12.0 and any("IRF taps cut" in str(w.message) for w in wrn)
    X = util.normed_to_matrix(normed, 1)
    E, I = eng.forward(X, batch["e_amps"], batch["i_amps"], batch["noise_e"], batch["noise_i"])
    Eo, Io, _, _ = orc.ts_diag(cfg, sa, normed, batch)
    assert util.rel_err(E.cpu().numpy(), Eo) < 1e-8 and util.rel_err(I.cpu().numpy(), Io) < 1e-8
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms, grad, _, _ = eng.loss_grad(X, batch, w, eng.slots.active.astype(np.uint8))
    lo, _, _ = orc.loss(cfg, sa, normed, batch, i_norm, e_norm)
    assert abs(float(np.dot(terms.cpu().numpy(), w)) - lo) < 1e-8 * abs(lo) and bool(torch_mod.isfinite(grad).all())


@pytest.mark.parametrize("seed", range(8))
def test_one_sweep_kernel_random_geometry(torch_mod, seed):
    """k_spectrum_fused / k_spectrum_rows (1, 2, 3, 5 points per pixel) on randomly drawn geometry: 3 to 24 scattering angles (the base-point exchange between lanes is taken
    up to 16 angles, the plain form above), EPW windows that contain the laser line, end beside it or lie wholly on one side (all
    128-sample units asymptotic / none), IAW windows of different width, cold and hot ions (|xi_i| on both sides of the 28 that
    switches the asymptotic ion terms), one or two species, drifts.  Against the two-sweep kernel (spectra and loss sums the
    same bits, gradient 1e-11) and against the C++ oracle (spectra 1e-8, gradient 1e-6)."""
    from oracle import c_oracle as co

    rng = np.random.default_rng(7000 + seed)
    n_ion = 1 + seed % 2
    ppp = (1, 1, 2, 1, 5, 1, 3, 2)[seed % 8]   # (> 1: k_spectrum_rows, the one-sweep kernel in rounds with its rows in global memory)
    cfg = decks.deck_fit(points_per_pixel=ppp, active=("Te", "ne", "Ti", "Va", "lam", "amp1", "amp2", "amp3", "ud"), n_ion=n_ion)
    other = cfg["other"]
    lo = float(rng.choice([400.0, 450.0, 500.0, 528.5, 540.0]))
    hi = float(rng.choice([520.0, 526.0, 560.0, 700.0])) if lo < 520.0 else float(rng.choice([600.0, 700.0]))
    c = float(rng.uniform(526.0, 527.0))
    hw = float(rng.choice([0.4, 0.75, 2.5]))
    r0 = cfg["data"]["fit_rng"]   # (decks.finish derives lamrangE / lamrangI from these)
    r0["forward_epw_start"], r0["forward_epw_end"] = lo, hi
    r0["forward_iaw_start"], r0["forward_iaw_end"] = c - hw, c + hw
    # (IRF widths in units of the sample spacing like the shipped decks', up to twice as wide; with several points per pixel the
    #  ion IRF stays below 1.4 x the shipped width: beyond ~1000 taps the spectrum + halo no longer fit the LDS and tsff_create says so)
    other["PhysParams"]["widIRF"] = {"spect_stddev_ele": float(rng.uniform(0.5, 2.0)) * (hi - lo) / 300.0,
                                     "spect_stddev_ion": float(rng.uniform(0.008, 0.03 if ppp == 1 else 0.02)) * hw / 0.75}
    r = cfg["data"]["fit_rng"]
    r["blue_min"], r["blue_max"] = lo + 0.1 * (hi - lo), lo + 0.45 * (hi - lo)
    r["red_min"], r["red_max"] = lo + 0.55 * (hi - lo), lo + 0.9 * (hi - lo)
    r["iaw_min"], r["iaw_max"] = c - 0.8 * hw, c + 0.8 * hw
    r["iaw_cf_min"], r["iaw_cf_max"] = c - 0.05 * hw, c + 0.05 * hw
    cfg = decks.finish(cfg)
    B = 3
    na = int(rng.integers(3, 25))
    ang = np.sort(rng.uniform(20.0, 150.0, na))
    wts = rng.uniform(0.2, 1.0, na)
    sa = dict(sa=ang, weights=(wts / wts.sum()) * np.ones([B, na]))
    batch = util.synthetic_batch(cfg, sa, B, seed=7100 + seed)
    normed = util.random_lineouts(cfg, B, seed=7200 + seed, ranges=dict(ud=(-2, 2), Va=(-4, 4), Ti_1=(0.012, 0.9), Ti_2=(0.012, 0.9), Te=(0.05, 1.4)))
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    eng = _engine(cfg, sa)
    X = util.normed_to_matrix(normed, n_ion)
    w = eng.loss_weights(B, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    gm = eng.slots.active.astype(np.uint8)
    out = {}
    for plan in (0, 8, 2):   # one-sweep (exchange where it applies), one-sweep without the exchange, two-sweep
        eng.set_launch_plan(plan)
        out[plan] = [a.cpu().numpy() for a in eng.loss_grad(X, batch, w, gm, want_spectra=True)]
    eng.set_launch_plan(0)
    assert np.isfinite(out[0][1]).all() and np.isfinite(out[0][2]).all()
    for plan in (8, 2):
        for k in (2, 3):
            np.testing.assert_array_equal(out[0][k], out[plan][k])
        if ppp == 1 or plan == 8: np.testing.assert_array_equal(out[0][0], out[plan][0])
        else: np.testing.assert_allclose(out[0][0], out[plan][0], rtol=1e-14)   # (loss sums folded over another thread count)
        np.testing.assert_allclose(out[0][1], out[plan][1], rtol=1e-10, atol=1e-12 * np.abs(out[0][1]).max())
    np.testing.assert_array_equal(out[0][1], out[8][1])   # the exchanged base points are the ones the lane would have computed
    sums, gref, Eo, Io = co.loss_grad(cfg, sa, X, batch, w=w, gmask=gm)
    # (against the oracle the bound is set by the conditioning of the decks drawn here, not by the kernels: Te down to 50 eV and Ti
    #  down to 12 eV give resonances so narrow that last-bit differences in k and omega_pe move the peak by 1e-8 of its height --
    #  seeds 14 and 22 of a 50-seed soak; the kernels agree with each other to the last bit above)
    assert util.rel_err(out[0][2], Eo) < 1e-7 and util.rel_err(out[0][3], Io) < 1e-7, (lo, hi, na)
    np.testing.assert_allclose(out[0][0], sums.sum(axis=0), rtol=1e-8)
    for sl in np.nonzero(gm)[0]:
        assert np.max(np.abs(out[0][1][:, sl] - gref[:, sl])) <= 1e-6 * np.max(np.abs(gref)) , (sl, lo, hi, na)


@pytest.mark.parametrize("ccd,n_lam,start,end", [((1024, 1024), 1024, 90, 950), ((128, 256), 256, 10, 110)])
def test_ats_adjoint_directional_derivatives(torch_mod, ccd, n_lam, start, end):
    """Reverse of the ARTS instrument chain (tsff_ats_adjoint): for a random linear functional <Ebar, ThryE(P)> the
    adjoint image Pbar and the amplitude adjoints against central differences of the forward chain along random
    directions (the chain has kinks only at ties of the row maxima)."""
    cfg = decks.deck_angular(1, 64, ccd, start, end)
    sa = _angular_sa(cfg)
    eng = _engine(cfg, sa, fe_mode=L.FE_PER_LINEOUT)
    normed = util.random_lineouts(cfg, 1, seed=5)
    phys = orc.physical_params(cfg["parameters"], normed, True)
    X = util.normed_to_matrix(phys, 1)
    P = eng.form_factor(0, X, orc.dlm_fe(2.7, 64)[None, :])[0]
    wid = cfg["other"]["PhysParams"]["widIRF"]
    eng.ats_setup(sa["weights"], sa["angAxis"], wid["spect_FWHM_ele"] / 2.3548, wid["ang_FWHM_ele"] / 2.3548,
                  1024 // n_lam, 1024 // ccd[0], start, end)
    rows = end - start
    rng = np.random.default_rng(11)
    e_amps = rng.uniform(0.5, 2.0, rows)
    Ebar = rng.normal(size=(rows, n_lam))
    lam, a1, a2 = 526.5, 0.8, 1.3
    Pbar, (a1b, a2b) = eng.ats_adjoint(P, e_amps, lam, a1, a2, Ebar)
    Pbar = Pbar.cpu().numpy()
    Pn = P.cpu().numpy()

    def func(Pm, b1=a1, b2=a2):
        return float(np.sum(Ebar * eng.ats_spectrum(Pm, e_amps, lam, b1, b2).cpu().numpy()))

    for k in range(3):
        d = Pn * rng.normal(size=Pn.shape)
        h = 1e-6
        fd = (func(Pn + h * d) - func(Pn - h * d)) / (2 * h)
        an = float(np.sum(Pbar * d))
        assert abs(fd - an) < 2e-5 * max(abs(an), abs(fd), 1e-30), (k, fd, an)
    h = 1e-6
    assert abs((func(Pn, a1 + h) - func(Pn, a1 - h)) / (2 * h) - a1b) < 1e-7 * abs(a1b)
    assert abs((func(Pn, a1, a2 + h) - func(Pn, a1, a2 - h)) / (2 * h) - a2b) < 1e-7 * abs(a2b)


@pytest.mark.parametrize("nv,n_ion,G", [(48, 2, 3), (132, 1, 1)])
def test_form_factor_2d_grad_finite_differences(torch_mod, nv, n_ion, G):
    """Adjoint of the 2-D path (tsff_form_factor_2d_grad): J = <Pbar, P(phys, fe2d)>.  d J / d phys and d J / d fe2d[i][j]
    against central differences of the (oracle-checked) forward tsff_form_factor_2d.  nv = 48: tables in LDS, two ion
    species, three gradient points; nv = 132: table and its adjoint through L2 / global atomics."""
    torch = torch_mod
    cfg = decks.deck_fit(n_ion=n_ion)
    if G > 1:
        g = cfg["parameters"]["general"]
        g["Te_gradient"].update(val=6.0, num_grad_points=G)
        g["ne_gradient"].update(val=9.0, num_grad_points=G)
    B = 2
    sa = dict(sa=np.array([35.0, 60.0, 110.0]), weights=np.ones((B, 3)) / 3)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, B, seed=67, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    phys["ud"] = np.array([0.8, -1.1])
    X = util.normed_to_matrix(phys, n_ion)
    _, fe2 = _fe2d(nv, "anisotropic")
    ud_ang, va_ang = 25.0, -40.0
    rng = np.random.default_rng(8)
    names = ["Te", "ne", "lam", "ud", "Va", "Ti_1", "Z_1"] + (["Te_gradient", "ne_gradient", "Ti_2", "Z_2", "fract_1"] if G > 1 else [])
    entries = [(0, 0), (0, 7), (nv - 1, nv - 1), (nv // 2, nv // 2), (nv // 2 + 3, nv // 2 - 5), (nv - 1, 3), (1, nv - 2)]
    for feature in (0, 1):
        P0 = eng.form_factor_2d(feature, X, fe2, ud_ang, va_ang)
        Pbar = torch.as_tensor(rng.standard_normal(tuple(P0.shape)), device=P0.device) / P0.abs().mean()

        Jabs = float((P0 * Pbar).abs().sum())

        def J(Xm, f):
            return float((eng.form_factor_2d(feature, Xm, f, ud_ang, va_ang) * Pbar).sum())

        gp, gf = eng.form_factor_2d_grad(feature, X, fe2, Pbar, ud_ang, va_ang)
        gp, gf = gp.cpu().numpy(), gf.cpu().numpy()
        assert np.all(np.isfinite(gp)) and np.all(np.isfinite(gf))
        for b in range(B):
            for nm in names:
                s = util.slot_of(nm)
                # the forward has kinks (linear interpolation of f1 at |xi_e| and of the Z' table): the step must keep
                # the samples inside their cells, and lam moves omega - omega_L a million times faster than the others
                # (a sample that crosses a table node inside the step spoils the quotient: the best of three steps counts)
                tried = []
                for hr in (1e-8, 1e-9, 1e-10, 1e-11) if nm == "lam" else (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
                    h = hr * max(abs(X[b, s]), 1e-2)
                    Xp, Xm = X.copy(), X.copy()
                    Xp[b, s] += h
                    Xm[b, s] -= h
                    fd = (J(Xp, fe2) - J(Xm, fe2)) / (2 * h)
                    scale = max(abs(fd), 1e-3 * np.max(np.abs(gp[:, s])))
                    # + the rounding floor of the quotient: J is a sum of Jabs worth of terms in float64
                    tried.append((abs(gp[b, s] - fd) - 2e-15 * Jabs / h) / scale)
                assert min(tried) < (2e-3 if nm == "lam" else 1e-4), (feature, b, nm, gp[b, s], fd, tried)
        for (i, j) in entries:
            h = 1e-6 * fe2.max()
            fp, fm = fe2.copy(), fe2.copy()
            fp[i, j] += h
            fm[i, j] -= h
            fd = (J(X, fp) - J(X, fm)) / (2 * h)
            assert abs(gf[i, j] - fd) < 2e-5 * max(abs(fd), 1e-4 * np.max(np.abs(gf))), (feature, i, j, gf[i, j], fd)
        # without the table adjoint the parameter gradient is the same
        gp2, none = eng.form_factor_2d_grad(feature, X, fe2, Pbar, ud_ang, va_ang, want_table=False)
        assert none is None and np.allclose(gp2.cpu().numpy(), gp, rtol=1e-12, atol=0)
        # fit-loop form: the forward keeps the projection records (tsff_form_factor_2d_save), the adjoint does no sampling
        if nv <= 256:
            P1 = eng.form_factor_2d(feature, X, fe2, ud_ang, va_ang, save=True)
            assert torch.allclose(P1, P0, rtol=1e-13, atol=0)
            gp3, gf3 = eng.form_factor_2d_grad(feature, X, fe2, Pbar, ud_ang, va_ang, use_saved=True)
            assert np.max(np.abs(gp3.cpu().numpy() - gp)) < 1e-11 * np.max(np.abs(gp))
            assert np.max(np.abs(gf3.cpu().numpy() - gf)) < 1e-11 * np.max(np.abs(gf))
            # records are tied to what they were made from: another table (or parameters) -> the adjoint samples for itself
            fe_other = fe2 * (1.0 + 0.1 * np.cos(np.arange(nv))[:, None])
            gp4, gf4 = eng.form_factor_2d_grad(feature, X, fe_other, Pbar, ud_ang, va_ang, use_saved=True)
            gp5, gf5 = eng.form_factor_2d_grad(feature, X, fe_other, Pbar, ud_ang, va_ang)
            assert np.array_equal(gp4.cpu().numpy(), gp5.cpu().numpy())   # (fixed-order reductions: the same bits)
            assert np.max(np.abs(gf4.cpu().numpy() - gf5.cpu().numpy())) <= 1e-13 * np.max(np.abs(gf5.cpu().numpy()))
            eng.form_factor_2d(feature, X, fe2, ud_ang, va_ang)   # a plain forward invalidates the records
            gp6, _ = eng.form_factor_2d_grad(feature, X, fe2, Pbar, ud_ang, va_ang, use_saved=True, want_table=False)
            assert np.allclose(gp6.cpu().numpy(), gp, rtol=1e-12, atol=0)
            # the library itself refuses records that do not belong to the call (stale range, other buffers)
            P1 = eng.form_factor_2d(feature, X, fe2, ud_ang, va_ang, save=True)
            sv = eng._saved_2d
            other = eng.dev(fe2.copy())
            gpx = torch.empty((B, eng.NP), dtype=torch.float64, device=eng.device)
            rc = eng.lib.tsff_form_factor_2d_grad(eng.h, feature, eng._ptr(sv["phys_d"]), eng._ptr(other), nv, ud_ang, va_ang, B, 0, -1, sv["token"],
                                                  eng._ptr(eng.dev(Pbar)), eng._ptr(gpx), None)
            assert rc == -2 and b"other inputs" in eng.lib.tsff_last_error(eng.h)
            # ... and a token that is not the one of the LAST save: a made-up one, and the previous generation's after a new save
            assert sv["token"] != 0
            rc = eng.lib.tsff_form_factor_2d_grad(eng.h, feature, eng._ptr(sv["phys_d"]), eng._ptr(sv["fe_d"]), nv, ud_ang, va_ang, B, 0, -1,
                                                  sv["token"] ^ 0x10000, eng._ptr(eng.dev(Pbar)), eng._ptr(gpx), None)
            assert rc == -22 and b"stale or foreign token" in eng.lib.tsff_last_error(eng.h)
            old_token = sv["token"]
            eng.form_factor_2d(feature, X, fe2, ud_ang, va_ang, save=True)
            sv2 = eng._saved_2d
            assert sv2["token"] not in (0, old_token)
            rc = eng.lib.tsff_form_factor_2d_grad(eng.h, feature, eng._ptr(sv2["phys_d"]), eng._ptr(sv2["fe_d"]), nv, ud_ang, va_ang, B, 0, -1,
                                                  old_token, eng._ptr(eng.dev(Pbar)), eng._ptr(gpx), None)
            assert rc == -22
            rc = eng.lib.tsff_form_factor_2d_grad(eng.h, feature, eng._ptr(sv2["phys_d"]), eng._ptr(sv2["fe_d"]), nv, ud_ang, va_ang, B, 0, -1,
                                                  sv2["token"], eng._ptr(eng.dev(Pbar)), eng._ptr(gpx), None)
            assert rc == 0 and np.allclose(gpx.cpu().numpy(), gp, rtol=1e-10, atol=0)


@pytest.mark.parametrize("fe_type", ["arbitrary", "sphericalharmonic", "sphericalharmonic-nn"])
def test_angular_2d_vg_loss_adjoint(torch_mod, fe_type):
    """LossFunction.vg_loss for an ARTS deck with a 2-D distribution function: the gradient comes from the hand-written
    adjoint (loss seed -> tsff_ats_adjoint -> tsff_form_factor_2d_grad -> parameter transform / table generator) and is
    compared leaf by leaf with central differences of the loss value (full GPU forwards): plasma parameters, the two
    drift speeds, amplitudes, and the distribution function itself -- all nvx^2 values of Arbitrary2V.fval in one
    evaluation, or the three Mora-Yahi generator parameters."""
    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd.loss_function import LossFunction

    nvx = 48
    cfg = decks.deck_angular(2, nvx, (128, 256), 10, 110)
    n_gen = 3
    if fe_type == "sphericalharmonic":
        cfg["parameters"]["electron"]["fe"] = {"active": True, "dim": 2, "type": "sphericalharmonic", "nvx": nvx, "params": {
            "flm_type": "mora-yahi", "init_m": 2.2, "LTx": 225000.0, "LTy": 400000.0, "Nl": 1, "nvr": 64}}
    if fe_type == "sphericalharmonic-nn":   # FLM_NN radial functions (spherical_harmonics.py:14-50) with caller-supplied layer weights
        rw = np.random.default_rng(8)
        sizes = [1, 6, 6, 6, 1]
        mk = lambda sc, b0: {"weights": [rw.normal(0, sc, (sizes[j + 1], sizes[j])) for j in range(4)],
                             "biases": [rw.normal(0, 0.2, sizes[j + 1]) + (b0 if j == 3 else 0.0) for j in range(4)]}
        cfg["parameters"]["electron"]["fe"] = {"active": True, "dim": 2, "type": "sphericalharmonic", "nvx": nvx, "params": {
            "flm_type": "nn", "init_m": 2.2, "Nl": 1, "nvr": 64,
            "nn_weights": {f"1,{m}": {"flm_mag": mk(0.6, 1.5), "flm_sign": mk(0.8, 0.0)} for m in (0, 1)}}}
        n_gen = 2 * 2 * (6 + 36 + 36 + 6) + 1
    g = cfg["parameters"]["general"]
    for k, val in (("ud", 0.6), ("Va", -0.8)):
        g[k]["val"], g[k]["active"] = val, True
    g["amp2"]["active"] = True
    sa = _angular_sa(cfg)
    # data: the model itself at a different plasma condition
    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    truth = tp.copy()
    truth.X[0, L.P_TE] -= 0.3
    truth.X[0, L.P_NE] += 0.25
    batch = dict(e_data=np.ones((100, 256)), i_data=np.zeros((100, 256)), e_amps=np.ones((100, 1)), i_amps=np.zeros(100),
                 noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    loss_fn = LossFunction(cfg, sa, batch)
    data = loss_fn.ts_diag(truth, batch)[0]
    batch["e_data"] = data
    loss_fn = LossFunction(cfg, sa, batch)
    spec = tree.get_filter_spec(cfg["parameters"], tp)
    names = [n for n, _ in spec]
    assert ("electron", "fval" if fe_type == "arbitrary" else "fe") in names and ("general", "ud") in names
    diff, static = tree.partition(tp, spec)
    x0, loss_fn.unravel_weights = tree.ravel_pytree(diff)
    assert x0.size == len(spec) - 1 + (nvx * nvx if fe_type == "arbitrary" else n_gen)
    val, gflat = loss_fn.vg_loss(x0, static, batch)
    assert np.isfinite(val) and val > 0 and gflat.shape == x0.shape and np.all(np.isfinite(gflat))
    grads = diff.like(gflat)

    def value(x):
        return loss_fn._angular_value(tree.combine(static, diff.like(x)), batch)[0]

    assert abs(value(x0) - val) < 1e-13 * val
    o = 0
    rng = np.random.default_rng(4)
    for (name, s), v in zip(diff.slots, diff.values):
        gl = gflat[o:o + v.size]
        if v.size > 8:   # the free-form table: a few single entries and one random direction through all of them
            fv = v.ravel()
            cand = np.argsort(-np.abs(gl))[:3].tolist() + [int(rng.integers(v.size)) for _ in range(2)]
            for i in cand:
                h = 1e-6 * max(abs(fv[i]), 1.0)
                e = np.zeros_like(x0)
                e[o + i] = h
                fd = (value(x0 + e) - value(x0 - e)) / (2 * h)
                assert abs(gl[i] - fd) < 1e-4 * max(abs(fd), 1e-3 * np.max(np.abs(gl))), (name, i, gl[i], fd)
            d = np.zeros_like(x0)
            d[o:o + v.size] = rng.standard_normal(v.size)
            h = 1e-7
            fd = (value(x0 + h * d) - value(x0 - h * d)) / (2 * h)
            assert abs(np.dot(gflat, d) - fd) < 1e-4 * abs(fd), (name, np.dot(gflat, d), fd)
        else:
            for i in range(v.size):
                h = 1e-9 if name[1] == "lam" else 1e-6   # (lam: see test_form_factor_2d_grad_finite_differences)
                e = np.zeros_like(x0)
                e[o + i] = h
                fd = (value(x0 + e) - value(x0 - e)) / (2 * h)
                tol = 5e-3 if name[1] == "lam" else 2e-4
                assert abs(gl[i] - fd) < tol * max(abs(fd), 1e-4 * np.max(np.abs(gflat))), (name, i, gl[i], fd)
        o += v.size
    # the finite-difference fallback (force_fd) agrees on the scalar leaves
    if fe_type == "sphericalharmonic":
        loss_fn.force_fd = True
        v2, g2 = loss_fn.vg_loss(x0, static, batch)
        loss_fn.force_fd = False
        assert abs(v2 - val) < 1e-13 * abs(val)
        lam_i = names.index(("general", "lam"))
        keep = np.ones(x0.size, bool)
        keep[lam_i if lam_i < names.index(("electron", "fe")) else lam_i + 2] = False
        assert np.max(np.abs(g2 - gflat)[keep]) < 2e-3 * np.max(np.abs(gflat[keep])), (g2, gflat)


@pytest.mark.parametrize("n_ion,G", [(1, 1), (2, 3)])
def test_form_factor_grad_finite_differences(torch_mod, n_ion, G):
    """Adjoint of the raw 1-D form factor for an arbitrary seed (tsff_form_factor_grad, what the angular instrument chain
    feeds): J = <Pbar, P(phys, fe)>; d J / d phys and d J / d fe[b][i] (through the Hermite ln f_e lookup and the ratintn
    table) against central differences of the oracle-checked forward tsff_form_factor."""
    torch = torch_mod
    cfg = decks.deck_fit(n_ion=n_ion)
    if G > 1:
        g = cfg["parameters"]["general"]
        g["Te_gradient"].update(val=6.0, num_grad_points=G)
        g["ne_gradient"].update(val=9.0, num_grad_points=G)
    B = 2
    sa = dict(sa=np.array([35.0, 60.0, 85.0, 110.0, 135.0]), weights=np.ones((B, 5)) / 5)
    eng = _engine(cfg, sa, fe_mode=L.FE_PER_LINEOUT)
    normed = util.random_lineouts(cfg, B, seed=91, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    X = util.normed_to_matrix(phys, n_ion)
    nvx = cfg["parameters"]["electron"]["fe"]["nvx"]
    fe = _free_form_fe(B, nvx, 9)
    rng = np.random.default_rng(10)
    names = ["Te", "ne", "lam", "ud", "Va", "Ti_1", "Z_1"] + (["Te_gradient", "ne_gradient", "Ti_2", "Z_2", "fract_1"] if G > 1 else [])
    for feature in (0, 1):
        P0 = eng.form_factor(feature, X, fe)
        Pbar = torch.as_tensor(rng.standard_normal(tuple(P0.shape)), device=P0.device) / P0.abs().mean()

        Jabs = float((P0 * Pbar).abs().sum())

        def J(Xm, f):
            return float((eng.form_factor(feature, Xm, f) * Pbar).sum())

        gp, gf = eng.form_factor_grad(feature, X, fe, Pbar, want_fe=True)
        gp, gf = gp.cpu().numpy(), gf.cpu().numpy()
        assert np.all(np.isfinite(gp)) and np.all(np.isfinite(gf)) and gf.shape == (B, nvx)
        gp0, none = eng.form_factor_grad(feature, X, fe, Pbar)
        assert none is None and np.allclose(gp0.cpu().numpy(), gp, rtol=1e-11, atol=1e-11 * np.abs(gp).max())
        for b in range(B):
            for nm in names:
                s = util.slot_of(nm)
                # kinks of the table lookups (see the 2-D test): a sample that crosses a table node inside the step spoils
                # the quotient, so three step sizes are tried and the best one counts
                tried = []
                for hr in (1e-8, 1e-9, 1e-10, 1e-11) if nm == "lam" else (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
                    h = hr * max(abs(X[b, s]), 1e-2)
                    Xp, Xm = X.copy(), X.copy()
                    Xp[b, s] += h
                    Xm[b, s] -= h
                    fd = (J(Xp, fe) - J(Xm, fe)) / (2 * h)
                    scale = max(abs(fd), 1e-3 * np.max(np.abs(gp[:, s])))
                    # + the rounding floor of the quotient: J is a sum of Jabs worth of terms in float64
                    tried.append((abs(gp[b, s] - fd) - 2e-15 * Jabs / h) / scale)
                assert min(tried) < (2e-3 if nm == "lam" else 1e-4), (feature, b, nm, gp[b, s], fd, tried)
            # d J / d ln fe[i] = fe[i] gf[i] (the far tail, fe < 1e-6 of the peak, cannot be resolved by a difference quotient)
            ok = np.flatnonzero(fe[b] > 1e-6 * fe[b].max())
            glog = fe[b] * gf[b]
            for i in [int(ok[np.argmax(np.abs(glog[ok]))]), nvx // 2, nvx // 2 + 7, int(ok[0]), int(ok[-1])]:
                h = 1e-6 * fe[b, i]
                fp, fm = fe.copy(), fe.copy()
                fp[b, i] += h
                fm[b, i] -= h
                fd = (J(X, fp) - J(X, fm)) / (2 * h)
                assert abs(gf[b, i] - fd) * fe[b, i] < 1e-4 * max(abs(fd) * fe[b, i], 1e-3 * np.max(np.abs(glog[ok]))), (feature, b, i, gf[b, i], fd)


def test_committed_golden_fixture_2d(torch_mod):
    """tests/golden/oracle_2d.npz (made by tests/golden/make_golden_2d.py from the CPU oracle alone): the 2-D form factor on
    a 48 x 48 anisotropic table (both features, oblique drift and flow) and the ARTS image of a 1-D DLM deck through the
    drop-in ThomsonScatteringDiagnostic -- the committed regression pin of the angular path."""
    from tsadar_amd import ThomsonParams
    from tsadar_amd.diagnostic import ThomsonScatteringDiagnostic

    z = np.load("tests/golden/oracle_2d.npz")
    cfg = decks.deck_fit()
    sa = dict(sa=z["ff_sa"], weights=np.ones((2, 3)) / 3)
    eng = _engine(cfg, sa)
    for feature in (0, 1):
        P = eng.form_factor_2d(feature, z["ff_X"], z["ff_fe2d"], float(z["ff_ud_angle"]), float(z["ff_va_angle"])).cpu().numpy()
        ref = z[f"ff_P{feature}"]
        err = np.max(np.abs(P[:, :, z["ff_idx"], :] - ref) / np.abs(ref))
        assert err < 1e-7, (feature, err)
    acfg = decks.deck_angular(1, 64, (128, 256), 10, 110)
    asa = _angular_sa(acfg)
    diag = ThomsonScatteringDiagnostic(acfg, asa)
    tp = ThomsonParams(acfg["parameters"], 1, batch=False, activate=True)
    batch = dict(e_data=np.ones((100, 256)), i_data=np.zeros((100, 256)), e_amps=z["ats_e_amps"], i_amps=np.zeros(100),
                 noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    E, _, lamE, _ = diag(tp, batch)
    assert E.shape == z["ats_E"].shape and np.max(np.abs(E - z["ats_E"])) < 1e-8 * np.max(np.abs(z["ats_E"]))
    np.testing.assert_allclose(lamE, z["ats_lam"], rtol=1e-13)


def test_angular_optax_loop_like_reference(torch_mod):
    """The reference's angular fit loop (inverse/loops.py:167-275: batch=False parameters, optax Adam on the partitioned
    pytree, ``(val, aux), grad = loss_fn.vg_loss(diff_params, static_params, data)``) on an ARTS deck whose distribution
    function is a free-form 48 x 48 table: data from a different table (another super-Gaussian order) and temperature;
    the loop -- all 2304 table values and the plasma parameters trained together through the adjoint -- lowers the loss."""
    from tsadar_amd import ThomsonParams, tree
    from tsadar_amd import distribution as Dist
    from tsadar_amd.loss_function import LossFunction

    nvx = 48
    cfg = decks.deck_angular(2, nvx, (128, 256), 10, 110)
    cfg["optimizer"]["method"] = "adam"
    cfg["optimizer"]["learning_rate"] = 0.001   # (0.004: 22 x lower loss in 60 steps; 0.01 oscillates)
    sa = _angular_sa(cfg)
    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    truth = tp.copy()
    truth.X[0, L.P_TE] -= 0.3
    truth.fval2d = Dist.arbitrary_2v_init(3.2, nvx, truth.learn_log)
    batch = dict(e_data=np.ones((100, 256)), i_data=np.zeros((100, 256)), e_amps=np.ones((100, 1)), i_amps=np.zeros(100),
                 noise_e=np.array([0.0]), noise_i=np.array([0.0]))
    batch["e_data"] = LossFunction(cfg, sa, batch).ts_diag(truth, batch)[0]
    loss_fn = LossFunction(cfg, sa, batch)
    diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
    assert any(v.size == nvx * nvx for v in diff.values)
    opt = tree.Adam(cfg["optimizer"]["learning_rate"])
    state = opt.init(diff)
    losses = []
    for _ in range(40):
        (val, aux), grad = loss_fn.vg_loss(diff, static, batch)
        assert isinstance(grad, tree.DiffParams) and aux[0].shape == (100, 256) and aux[1]["electron"]["fe"].shape == (nvx, nvx)
        updates, state = opt.update(grad, state)
        diff = tree.apply_updates(diff, updates)
        losses.append(val)
    assert losses[-1] < 0.4 * losses[0] and max(losses) <= 1.05 * losses[0], losses


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 at its stated size: non-Maxwellian f_e on a 256 x 256 velocity grid, 512 scattering angles x 1024
# wavelengths (524 288 points, table read through L1/L2 from the padded copy of k_pad2d, table adjoint cut in 2 x 2
# tiles).  Reference: FormFactor.calc_in_2D / rotate / calc_chi_vals (form_factor.py:449-587, 300-324, 349-388).
# Parity with the reference itself is UNPINNED for this path (its ARTS goldens are not in the reference tree): the
# comparison is with the oracle's restatement.
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config4(torch_mod):
    cfg = decks.deck_fit()
    na = 512
    sa = dict(sa=np.linspace(19.0, 139.0, na), weights=np.ones((1, na)) / na)
    eng = _engine(cfg, sa)
    normed = util.random_lineouts(cfg, 1, seed=71, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    phys["ud"] = np.array([0.7])
    X = util.normed_to_matrix(phys, 1)
    vx, fe2 = _fe2d(256, "anisotropic")
    ud_ang, va_ang = 25.0, -40.0
    P = eng.form_factor_2d(0, X, fe2, ud_ang, va_ang)
    assert tuple(P.shape) == (1, 1, 1024, na) and bool(torch_mod.isfinite(P).all())
    return dict(cfg=cfg, sa=sa, eng=eng, phys=phys, X=X, vx=vx, fe2=fe2, ud=ud_ang, va=va_ang, P=P)


def test_config4_forward_matches_oracle(torch_mod, config4):
    """(i) forward at nv = 256, 512 angles: 8 scattered wavelength indices x 66 angles (every 8th, the first and the last)
    against the oracle's restatement (every point is independent of the others in the 2-D path), 1e-7 relative."""
    c = config4
    lam_idx = np.array([0, 1, 137, 400, 511, 512, 777, 1023])
    ang_idx = np.unique(np.concatenate([np.arange(0, 512, 8), [1, 511]]))
    p = orc.lineout_params(c["phys"], 0, 1)
    Po, _ = orc.form_factor_2d(c["cfg"]["other"]["lamrangE"], 1024, 0.0, c["sa"]["sa"][ang_idx], 1, p, c["vx"], c["fe2"], c["ud"], c["va"],
                               lam_index=lam_idx)
    Pg = c["P"].cpu().numpy()[0][:, lam_idx][:, :, ang_idx]
    err = np.max(np.abs(Pg - Po) / np.abs(Po))
    assert err < 1e-7, err


def test_config4_point_ranges_are_bit_identical(torch_mod, config4):
    """(ii) tsff_form_factor_2d_range over three uneven slices of the flat point list == one call, bit for bit (the unit the
    multi-GPU path shards, form_factor.py:431-447)."""
    c = config4
    eng, n = c["eng"], 1024 * 512
    out = torch_mod.zeros_like(c["P"])
    for lo, hi in ((0, 100003), (100003, 100004), (100004, n)):
        eng.form_factor_2d(0, c["X"], c["fe2"], c["ud"], c["va"], point_range=(lo, hi), out=out)
    assert bool((out == c["P"]).all())


def test_config4_adjoint_saved_records_and_finite_differences(torch_mod, config4):
    """(iii) the fit-loop form (forward keeps the projection records, adjoint does no sampling) == the plain adjoint to
    1e-13 (parameters 1e-14) with the table adjoint cut in 4 tiles of 128 x 128 cells; (iv) d <Pbar, P> / d fe2d[i][j] against central
    differences of the forward on 5 entries (2e-5, as at the smaller sizes)."""
    torch = torch_mod
    c = config4
    eng, X, fe2 = c["eng"], c["X"], c["fe2"]
    rng = np.random.default_rng(12)
    P0 = c["P"]
    Pbar = torch.as_tensor(rng.standard_normal(tuple(P0.shape)), device=P0.device) / P0.abs().mean()
    gp, gf = eng.form_factor_2d_grad(0, X, fe2, Pbar, c["ud"], c["va"])
    P1 = eng.form_factor_2d(0, X, fe2, c["ud"], c["va"], save=True)
    assert bool((P1 == P0).all())
    gp3, gf3 = eng.form_factor_2d_grad(0, X, fe2, Pbar, c["ud"], c["va"], use_saved=True)
    gp, gf, gp3, gf3 = (t.cpu().numpy() for t in (gp, gf, gp3, gf3))
    assert np.all(np.isfinite(gp)) and np.all(np.isfinite(gf))
    # (round 3: both kernels take the point's rotation from one function with pinned roundings, so the projection is the same bits
    #  whoever samples it: the parameter gradient agrees exactly, the table adjoint to the order of its LDS atomics -- 4e-16 measured;
    #  with cos(atan(y / x)) in one kernel and x / |xi_e| in the other the bound had to be 1e-11)
    assert np.max(np.abs(gp3 - gp)) <= 1e-14 * np.max(np.abs(gp)), np.max(np.abs(gp3 - gp)) / np.max(np.abs(gp))
    assert np.max(np.abs(gf3 - gf)) < 1e-13 * np.max(np.abs(gf)), np.max(np.abs(gf3 - gf)) / np.max(np.abs(gf))

    def J(f):
        return float((eng.form_factor_2d(0, X, f, c["ud"], c["va"]) * Pbar).sum())

    nv = 256
    for (i, j) in [(0, 0), (nv - 1, nv - 1), (nv // 2, nv // 2), (nv // 2 + 3, nv // 2 - 5), (127, 128)]:   # corners, centre, a tile seam
        h = 1e-6 * fe2.max()
        fp, fm = fe2.copy(), fe2.copy()
        fp[i, j] += h
        fm[i, j] -= h
        fd = (J(fp) - J(fm)) / (2 * h)
        assert abs(gf[i, j] - fd) < 2e-5 * max(abs(fd), 1e-4 * np.max(np.abs(gf))), (i, j, gf[i, j], fd)


def test_adjoints_are_run_to_run_reproducible(torch_mod):
    """SURVEY section 5 (fixed-order reductions): the lineout-scalar adjoints of tsff_form_factor_2d_grad and tsff_form_factor_grad
    are sums of per-workgroup partials folded in a fixed order (they used to be global atomics) -> bit-identical run to
    run; so are the loss sums and the gradient of tsff_loss_grad.  The table adjoints (d loss / d fe2d, d loss / d fe) are
    gathered with LDS atomics whose order is not fixed: reproducible to rounding (1e-13 of the largest entry)."""
    torch = torch_mod
    cfg = decks.deck_fit(n_ion=2)
    g = cfg["parameters"]["general"]
    g["Te_gradient"].update(val=6.0, num_grad_points=3)
    g["ne_gradient"].update(val=9.0, num_grad_points=3)
    B = 2
    sa = dict(sa=np.linspace(30.0, 120.0, 37), weights=np.ones((B, 37)) / 37)
    eng = _engine(cfg, sa, fe_mode=L.FE_PER_LINEOUT)
    normed = util.random_lineouts(cfg, B, seed=91, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    X = util.normed_to_matrix(phys, 2)
    rng = np.random.default_rng(92)
    for nv in (48, 132):   # table in LDS / through L2 with a tiled table adjoint
        _, fe2 = _fe2d(nv, "anisotropic")
        P0 = eng.form_factor_2d(0, X, fe2, 25.0, -40.0)
        Pbar = torch.as_tensor(rng.standard_normal(tuple(P0.shape)), device=P0.device) / P0.abs().mean()
        runs = [eng.form_factor_2d_grad(0, X, fe2, Pbar, 25.0, -40.0) for _ in range(3)]
        gp = [r[0].cpu().numpy() for r in runs]
        gf = [r[1].cpu().numpy() for r in runs]
        assert np.array_equal(gp[0], gp[1]) and np.array_equal(gp[0], gp[2]), nv
        assert max(np.max(np.abs(gf[0] - gf[k])) for k in (1, 2)) <= 1e-13 * np.max(np.abs(gf[0])), nv
    nvx = cfg["parameters"]["electron"]["fe"]["nvx"]
    fe = np.stack([orc.dlm_fe(m, nvx) for m in (2.3, 3.4)])
    P1 = eng.form_factor(0, X, fe)
    Pb1 = torch.as_tensor(rng.standard_normal(tuple(P1.shape)), device=P1.device) / P1.abs().mean()
    runs = [eng.form_factor_grad(0, X, fe, Pb1, want_fe=True) for _ in range(3)]
    assert np.array_equal(runs[0][0].cpu().numpy(), runs[1][0].cpu().numpy()) and np.array_equal(runs[0][0].cpu().numpy(), runs[2][0].cpu().numpy())
    gfe = [r[1].cpu().numpy() for r in runs]
    assert max(np.max(np.abs(gfe[0] - gfe[k])) for k in (1, 2)) <= 1e-13 * np.max(np.abs(gfe[0]))
    # the fit path: loss sums and gradient, one-sweep and two-sweep kernels
    cfg1 = decks.deck_fit()
    sa1, batch, normed1, i_norm, e_norm = _loss_setup(cfg1, 64, seed=93)
    eng1 = _engine(cfg1, sa1)
    w = eng1.loss_weights(64, i_norm, e_norm)
    X1 = util.normed_to_matrix(normed1, 1)
    for plan in (0, 2):
        eng1.set_launch_plan(plan)
        r = [eng1.loss_grad(X1, batch, w, eng1.slots.active.astype(np.uint8)) for _ in range(3)]
        for k in (1, 2):
            assert bool((r[0][0] == r[k][0]).all()) and bool((r[0][1] == r[k][1]).all()), plan


@pytest.mark.parametrize("tag", ["arts1v", "arts2v"])
def test_reference_angular_decks_against_oracle_fixture(torch_mod, tag):
    """The reference's own ARTS forward tests (tests/test_forward/test_angular_1v.py / _2v.py) on their own decks -- merged
    verbatim into tests/golden/<tag>_deck.json by tests/golden/make_golden_arts.py -- through the drop-in
    ThomsonScatteringDiagnostic, against the oracle's images for the same decks (tests/golden/oracle_arts.npz, every 10th
    row of [860, 1024]).  The reference's goldens ThryE-arts1v.npy / ThryE-arts2v.npy are absent from its tree, so this is
    parity with the ORACLE (unpinned against the reference); with the blobs at hand the check becomes
    ``assert_allclose(np.load("ThryE-<tag>.npy")[z["rows"]], E[z["rows"]], rtol=1e-4)``.
    arts1v: 1-D DLM f_e (nvx 256, m through the activation round trip), 2 points per pixel (2048 wavelengths -> 1024
    resolution units); arts2v: SphericalHarmonics / Mora-Yahi f_e on a 128 x 128 grid, 246 784 rotate-and-project points."""
    import json

    from tsadar_amd import ThomsonParams
    from tsadar_amd.diagnostic import ThomsonScatteringDiagnostic

    z = np.load("tests/golden/oracle_arts.npz")
    cfg = json.load(open(f"tests/golden/{tag}_deck.json"))
    sa = _angular_sa(cfg)   # (spectype "angular" for the geometry lookup, then "angular_full": test_angular_1v.py:53-58)
    n0, n1 = cfg["other"]["CCDsize"]
    batch = dict(e_data=np.ones((n0, n1)), i_data=np.ones((n0, n1)), noise_e=np.array([0]), noise_i=np.array([0]),
                 e_amps=np.array([1]), i_amps=np.array([1]))   # test_angular_1v.py:62-69
    diag = ThomsonScatteringDiagnostic(cfg, sa)
    tp = ThomsonParams(cfg["parameters"], num_params=1, batch=False, activate=True)
    E, I, lamE, lamI = diag(tp, batch)
    assert E.shape == (860, 1024)
    ref = z[f"ThryE_{tag}"]
    np.testing.assert_allclose(lamE, z[f"lam_{tag}"], rtol=1e-13)
    err = np.max(np.abs(E[z["rows"]] - ref)) / np.max(np.abs(ref))
    assert err < 1e-7, err
    np.testing.assert_allclose(E[z["rows"]], ref, rtol=1e-4, atol=1e-9 * np.max(ref))   # the reference's own tolerance form


def test_loss_grad_packed_layout(torch_mod):
    """tsff_loss_grad_packed: [S_iaw, S_blue, S_red | g[k][b_global]] with this call's B lineouts in the columns
    [b_offset, b_offset + B) of every row (ravel order of the trainable leaves) and ZERO in every other column -- also
    when the buffer held something else before (it is all-reduced in place step after step) -- for the one-sweep and the
    two-sweep kernel; equal to tsff_loss_grad's per-lineout gradient bit for bit."""
    torch = torch_mod
    cfg = decks.deck_fit()
    B, Bg, off = 5, 13, 6
    sa, batch, normed, i_norm, e_norm = _loss_setup(cfg, B, seed=61)
    eng = _engine(cfg, sa)
    X = util.normed_to_matrix(normed, 1)
    gm = eng.slots.active.astype(np.uint8)
    act = [s for s in range(eng.NP) if gm[s]][::-1]   # any order of the rows is the caller's choice
    w = eng.loss_weights(Bg, i_norm, e_norm)
    for plan in (0, 2, 1):
        eng.set_launch_plan(plan)
        terms, grad, _, _ = eng.loss_grad(X, batch, w, gm)
        out = torch.full((3 + len(act) * Bg,), 7.0, dtype=torch.float64, device=eng.device)
        packed, E, I = eng.loss_grad_packed(X, batch, w, gm, act, Bg, off, want_spectra=True, out=out)
        p = packed.cpu().numpy()
        assert np.array_equal(p[:3], terms.cpu().numpy())
        rows = p[3:].reshape(len(act), Bg)
        assert np.array_equal(rows[:, off:off + B], grad.cpu().numpy()[:, act].T), plan
        assert np.all(rows[:, :off] == 0.0) and np.all(rows[:, off + B:] == 0.0)
        assert E is not None and bool(torch.isfinite(E).all())
    eng.set_launch_plan(0)
    with pytest.raises(L.TsffError, match="packed-output"):
        eng.loss_grad_packed(X, batch, w, gm, act, Bg, Bg - 2)   # columns past the end of the global batch


@pytest.mark.parametrize("kind", ["epw", "iaw"])
def test_dispersion_known_answers_through_the_hip_path(torch_mod, kind):
    """The reference's two known-answer tests (tests/test_form_factor/test_epw.py:33-74: EPW peaks vs Bohm-Gross
    omega^2 = wpe^2 + 3 k^2 vTe^2; test_iaw.py:40-71: IAW peaks vs omega = 2 kL sqrt((Z Te + 3 Ti)/Mi); rtol 1e-2 both) on
    the spectrum the HIP path computes (tsff_form_factor, theta = 60 degrees, 4096 wavelengths), and the same spectrum against
    the oracle's."""
    from scipy.signal import find_peaks

    cfg = decks.deck_kat(kind)
    cfg["other"]["points_per_pixel"] = 4
    cfg["data"]["fit_rng"].update(forward_epw_start=400.0, forward_epw_end=700.0, forward_iaw_start=525.0, forward_iaw_end=528.0)
    cfg["other"]["extraoptions"]["load_ion_spec"] = True
    cfg = decks.finish(cfg)
    sa = dict(sa=np.array([60.0]), weights=np.ones((1, 1)))
    eng = _engine(cfg, sa, activate=False)
    P_ = cfg["parameters"]
    phys = {k: np.array([v]) for k, v in dict(Te=P_["electron"]["Te"]["val"], ne=P_["electron"]["ne"]["val"], m=2.0, lam=P_["general"]["lam"]["val"],
                                               amp1=1.0, amp2=1.0, amp3=1.0, ne_gradient=0.0, Te_gradient=0.0, ud=0.0, Va=0.0,
                                               Ti_1=P_["ion-1"]["Ti"]["val"], Z_1=P_["ion-1"]["Z"]["val"], A_1=P_["ion-1"]["A"]["val"], fract_1=1.0).items()}
    X = util.normed_to_matrix(phys, 1)
    feature, rng = (0, [400.0, 700.0]) if kind == "epw" else (1, [525.0, 528.0])
    npts = cfg["other"]["npts"]
    spec = eng.form_factor(feature, X).cpu().numpy()[0, 0, :, 0]
    p = orc.lineout_params(phys, 0, 1)
    Po, lam_cm = orc.form_factor(rng, npts, 0.0, sa["sa"], 1, p, orc.velocity_grid(128), orc.dlm_fe(2.0, 128))
    assert np.max(np.abs(spec - np.squeeze(Po)) / np.abs(np.squeeze(Po))) < 1e-7
    lam_cm = np.squeeze(lam_cm)
    omgpe = orc.C0 * np.sqrt(p["ne"] * 1e20)
    omgL = 2 * np.pi * 1e7 * orc.C / p["lam"]
    kL = np.sqrt(omgL**2 - omgpe**2) / orc.C
    if kind == "epw":
        peaks, props = find_peaks(spec, height=(0.01, 0.5), prominence=0.02)
        hi = peaks[np.argmax(props["peak_heights"])]
        lo = peaks[np.argsort(props["peak_heights"])[0]]
        model = 2 * np.pi * orc.C / lam_cm[[hi, lo]]
        ks = np.sqrt(model**2 - omgpe**2) / orc.C
        k = np.sqrt(ks**2 + kL**2 - 2 * ks * kL * np.cos(60 * np.pi / 180))
        omg = np.sqrt(omgpe**2 + 3 * k**2 * (p["Te"] / orc.ME))
        np.testing.assert_allclose(model, [omgL + omg[0], omgL - omg[1]], rtol=1e-2)
    else:
        peaks, props = find_peaks(spec, height=0.1, prominence=0.2)
        hi = peaks[np.argmax(props["peak_heights"])]
        second = peaks[np.argpartition(props["peak_heights"], -2)[-2]]
        model = 2 * np.pi * orc.C / lam_cm[[hi, second]]
        omg = 2 * kL * np.sqrt((0.5 + 3 * 0.2) / orc.MP)
        np.testing.assert_allclose(sorted([omgL + omg, omgL - omg]), sorted(model), rtol=1e-2)
        cs = np.sqrt((p["Z"][0] * p["Te"] + 3 * p["Ti"][0]) / (p["A"][0] * orc.MP))
        assert abs(abs(model[0] - model[1]) / (2 * kL * cs) - 1) < 0.1
