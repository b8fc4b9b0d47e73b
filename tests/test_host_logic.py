"""Host-side logic of the drop-in layer (no GPU): parameter container, pytree helpers, static tables."""
import copy

import numpy as np
import pytest

import decks
import util
from oracle import tsadar_oracle as orc
from tsadar_amd import ThomsonParams, _lib as L, distribution as D, tree
from tsadar_amd import engine as E
from tsadar_amd.calibration import get_scattering_angles, sa_lookup


def test_thomson_params_matches_oracle_leaves_and_physical_values():
    cfg = decks.deck_1d()
    tp = ThomsonParams(cfg["parameters"], num_params=3, batch=True, activate=True)
    normed = orc.init_normed_params(cfg["parameters"], 3, True)
    np.testing.assert_allclose(tp.to_matrix(), util.normed_to_matrix(normed, 1), rtol=0, atol=0)
    phys = orc.physical_params(cfg["parameters"], normed, True)
    P = tp.physical_matrix()
    for k, v in phys.items():
        np.testing.assert_allclose(P[:, util.slot_of(k)], v, rtol=1e-15)
    un = tp.get_unnormed_params()
    assert abs(un["electron"]["Te"][0] - 0.50175174) < 1e-8 and abs(un["general"]["lam"][0] - 524.02200177) < 1e-8
    fitted, n = tp.get_fitted_params(cfg["parameters"])
    assert n == 6 and set(fitted["electron"]) == {"Te", "ne", "m"} and set(fitted["general"]) == {"amp1", "amp2", "lam"}


def test_ravel_order_is_the_reference_pytree_order():
    """ravel_pytree(diff_params) of the reference (loops.py:40-41): parameter-major, field order
    Te, ne, m | Ti, Z | lam, amp1, amp2, amp3, ne_gradient, Te_gradient, ud, Va."""
    cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "lam", "amp1"))
    B = 4
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    tp.X[:] = np.arange(B * tp.X.shape[1]).reshape(B, -1)
    names = [n for n, _ in tp.slots.active_leaves]
    assert names == [("electron", "Te"), ("electron", "ne"), ("ion-1", "Ti"), ("general", "lam"), ("general", "amp1"), ("general", "Va")]
    diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
    flat, unravel = tree.ravel_pytree(diff)
    assert flat.shape == (6 * B,)
    np.testing.assert_array_equal(flat[:B], tp.X[:, L.P_TE])
    np.testing.assert_array_equal(flat[2 * B:3 * B], tp.X[:, L.P_ION0 + L.ION_TI])
    back = tree.combine(static, unravel(flat * 2))
    np.testing.assert_array_equal(back.X[:, L.P_LAM], 2 * tp.X[:, L.P_LAM])
    np.testing.assert_array_equal(back.X[:, L.P_AMP2], tp.X[:, L.P_AMP2])  # not trainable: untouched
    np.testing.assert_array_equal(tp.ravel_grad(tp.X), flat)


def test_dlm_table_and_maxwellian():
    np.testing.assert_allclose(D.dlm_table(128), orc.dlm_table(128), rtol=1e-14)
    for m in (2.0, 2.5158, 4.99):
        np.testing.assert_allclose(D.dlm(m, 128), orc.dlm_fe(m, 128), rtol=1e-13)
    vx = D.velocity_grid(128)
    mx = np.exp(-vx**2 / 2)
    mx = mx / mx.sum() / (vx[1] - vx[0])
    np.testing.assert_allclose(D.maxwellian(128), mx, rtol=1e-5)  # m = 2 column == Maxwellian (Q7) up to the table's 1e-3 grid


def test_gaussian_taps_reproduce_full_same_convolution():
    """Truncated taps keep the reference's 'same' alignment: identical to
    np.convolve(x, g, 'same') up to the dropped tail mass exp(-12^2/2)."""
    rng = np.random.default_rng(0)
    for npts, rngE, sd in ((1024, [400, 700], 1.3), (5120, [400, 700], 1.3), (1024, [525.75, 527.25], 0.015)):
        lam = E.wavelength_axis_nm(rngE, npts)
        x = rng.random(npts) ** 8
        origin = (lam.max() + lam.min()) / 2
        g = (1.0 / (sd * np.sqrt(2 * np.pi))) * np.exp(-((lam - origin) ** 2) / (2 * sd**2))
        ref = np.convolve(x, g, "same")
        taps, dmin = E.gaussian_taps(lam, sd, 12.0)
        y = np.zeros(npts)
        for t, d in enumerate(range(dmin, dmin + taps.size)):
            lo, hi = max(0, d), min(npts, npts + d)
            y[lo:hi] += taps[t] * x[lo - d:hi - d]
        assert np.max(np.abs(y - ref)) < 1e-25 * ref.max() + 1e-28 * ref.max() or np.max(np.abs(y - ref)) / ref.max() < 1e-14
        full, dmin_full = E.gaussian_taps(lam, sd, 0.0)
        assert full.size >= taps.size and dmin_full <= dmin


def test_static_tables_match_oracle():
    xi1, xi2 = E.xi_grids()
    o1, o2 = orc.xi_grids()
    np.testing.assert_array_equal(xi1, o1)
    np.testing.assert_array_equal(xi2, o2)
    zr, zi = E.zprime_tables(xi2)
    np.testing.assert_array_equal(zr, orc.zprime_tables()[0])
    np.testing.assert_array_equal(zi, orc.zprime_tables()[1])


def test_scattering_angles():
    sa = get_scattering_angles(decks.deck_1d())
    np.testing.assert_allclose(sa["sa"], util.P9["sa"])
    np.testing.assert_allclose(sa["weights"], util.P9["weights"])
    assert abs(sa["weights"].sum() - 1) < 1e-3
    for beam in ("B12", "B15", "B23", "B26", "B35", "B42", "B46", "B58", "B62"):
        s = sa_lookup(beam)
        assert s["sa"].shape == (10,) and abs(s["weights"].sum() - 1) < 2e-2
    import pytest

    with pytest.raises(NotImplementedError):
        sa_lookup("nope")


def test_binned_taps_equal_convolve_then_bin():
    """engine.binned_taps folds the ppp-sample bin average into the IRF taps: identical (to rounding) to
    np.convolve(x, g, 'same').reshape(1024, -1).mean(axis=1)."""
    rng = np.random.default_rng(1)
    for ppp, rngE, sd in ((1, [400, 700], 1.3), (2, [400, 700], 1.3), (5, [525.75, 527.25], 0.015)):
        npts = 1024 * ppp
        lam = E.wavelength_axis_nm(rngE, npts)
        x = rng.random(npts) ** 6
        origin = (lam.max() + lam.min()) / 2
        g = (1.0 / (sd * np.sqrt(2 * np.pi))) * np.exp(-((lam - origin) ** 2) / (2 * sd**2))
        ref = np.convolve(x, g, "same").reshape(1024, -1).mean(axis=1)
        taps, dmin = E.gaussian_taps(lam, sd, 12.0)
        hb, off = E.binned_taps(taps, dmin, ppp)
        assert hb.size == taps.size + ppp - 1
        pad = hb.size + abs(off) + ppp
        xp = np.concatenate([np.zeros(pad), x, np.zeros(pad)])
        y = np.array([np.dot(hb, xp[pad + p * ppp + off: pad + p * ppp + off + hb.size]) for p in range(1024)])
        assert np.max(np.abs(y - ref)) < 1e-13 * ref.max()


def test_angular_geometry_and_arbitrary_2v():
    """ARTS calibration data (calibration.py:456-458, 483-491), Arbitrary2V generator (base.py:375-427) and the
    oracle's instrument chain on a synthetic image: unit row maxima scaled by e_amps * amp1|amp2."""
    from oracle import tsadar_oracle as orc
    from tsadar_amd import ThomsonParams, calibration
    from tsadar_amd import distribution as D

    cfg = decks.deck_angular(2, 32, (128, 256), 10, 110)
    cfg["other"]["extraoptions"]["spectype"] = "angular"
    sa = calibration.get_scattering_angles(cfg)
    cfg["other"]["extraoptions"]["spectype"] = "angular_full"
    ang = calibration.angular_pixel_axis()
    assert sa["sa"].shape == (241,) and sa["weights"].shape == (1024, 241) and ang.shape == (1024,)
    assert np.all(np.diff(ang) > 0) and np.all(sa["weights"] >= 0)
    # one_d geometry lookup is what the reference returns for "angular_full" (calibration.py:483)
    assert calibration.get_scattering_angles(cfg)["sa"].shape == (10,)

    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    fe, vx = tp()["electron"]["fe"], tp()["electron"]["v"]
    assert fe.shape == (32, 32) and abs(np.sum(fe) * (vx[1] - vx[0]) ** 2 - 1) < 1e-13
    # learn_log round trip reproduces the grid-normalised super-Gaussian
    f_lin = D.arbitrary_2v(D.arbitrary_2v_init(2.5, 32, False), False)
    np.testing.assert_allclose(fe, f_lin, rtol=1e-12)
    with pytest.raises(NotImplementedError):
        ThomsonParams(cfg["parameters"], 2, batch=True)

    rng = np.random.default_rng(0)
    lam_nm = np.linspace(400, 700, 1024)
    P = np.exp(-0.5 * ((lam_nm[None, :, None] - 450 - 0.8 * np.arange(241)[None, None, :]) / 6.0) ** 2) + 1e-3
    e_amps = rng.uniform(0.5, 2.0, (100, 1))
    p = dict(lam=526.5, amp1=0.7, amp2=1.3)
    E, lam = orc.ats_spectrum(cfg, sa["weights"], ang, P, lam_nm, 256, e_amps, p)
    assert E.shape == (100, 256) and lam.shape == (256,)
    peak = np.max(E / np.where(lam < 526.5, 0.7, 1.3), axis=1)
    np.testing.assert_allclose(peak, e_amps[:, 0], rtol=1e-12)


def test_spherical_harmonics_generator():
    """SphericalHarmonics 2-D f_e (tests/configs/arts2d_test_inputs.yaml numbers: Mora-Yahi l = 1 harmonics) vs the
    oracle's restatement through scipy's sph_harm_y; normalisation, dipole asymmetry, error paths."""
    from oracle import tsadar_oracle as orc

    dc = {"nvx": 128, "dim": 2, "type": "sphericalharmonic", "active": True,
          "params": {"flm_type": "mora-yahi", "init_m": 2.2, "LTx": 225000.0, "LTy": 400000.0, "Nl": 1, "nvr": 64}}
    sh = D.SphericalHarmonics(dc)
    f = sh()
    fo = orc.spherical_harmonics_fe(dc)
    np.testing.assert_allclose(f, fo, rtol=1e-11, atol=1e-40)
    dv = sh.vx[1] - sh.vx[0]
    assert abs(np.sum(f) * dv * dv - 1.0) < 1e-13
    assert abs(sh.get_unnormed_m() - 2.2239479449367296) < 1e-12          # Q6 round trip of init_m = 2.2
    assert np.abs(f - f[:, ::-1]).max() > 1e-4 * f.max()                 # heat-flux dipole breaks the symmetry
    up = sh.get_unnormed_params()["flm"]
    assert set(up[1].keys()) == {0, 1} and up[0][0].shape == (64,)
    dc2 = copy.deepcopy(dc)
    dc2["params"]["flm_type"] = "arbitrary"
    f2 = D.SphericalHarmonics(dc2)()
    np.testing.assert_allclose(f2, f2[::-1, ::-1], rtol=1e-12)           # zero-initialised harmonics: isotropic
    # vector-Jacobian product of the generator (what LossFunction contracts the GPU's table adjoint with): the analytic chain
    # of the free radial functions against central differences of the generator, at non-trivial parameter values
    dc3 = copy.deepcopy(dc2)
    dc3["nvx"], dc3["params"]["nvr"], dc3["params"]["init_m"] = 64, 32, 2.6
    sh3 = D.SphericalHarmonics(dc3)
    rng = np.random.default_rng(0)
    th = sh3.get_params()
    th[:-1] = rng.normal(0.0, 0.6, th.size - 1)
    sh3.set_params(th)
    fbar = rng.standard_normal((64, 64))
    ga, gfd = sh3.vjp(fbar), sh3._vjp_fd(fbar)
    assert ga.shape == (2 * 2 * 32 + 1,) and np.max(np.abs(ga - gfd)) < 1e-7 * np.max(np.abs(gfd))
    np.testing.assert_array_equal(sh3.get_params(), th)                  # the generator is left untouched
    assert np.max(np.abs(sh.vjp(fbar.repeat(2, 0).repeat(2, 1)) - sh._vjp_fd(fbar.repeat(2, 0).repeat(2, 1)))) == 0.0  # Mora-Yahi: FD
    dc2["params"]["flm_type"] = "no-such-model"
    with pytest.raises(NotImplementedError):
        D.SphericalHarmonics(dc2)
    cfg = decks.deck_angular(2, 128)
    cfg["parameters"]["electron"]["fe"] = dc
    tp = ThomsonParams(cfg["parameters"], 1, batch=False, activate=True)
    np.testing.assert_allclose(tp()["electron"]["fe"], f, rtol=0, atol=0)


def test_arbitrary_1v_generator_and_ravel_order():
    """Arbitrary1V host mirror: Butterworth smoothing as a matrix == the scan, normalisation, VJP vs finite
    differences, and the position of the fval leaves in the flat vector."""
    nvx = 48
    fv = D.arbitrary_1v_init(2.5, nvx) * (1 + 0.05 * np.sin(np.arange(nvx)))
    S = D.butterworth_matrix(nvx)
    fwd = D._butterworth_pass(fv, 100.0, 6.0)
    ref = D._butterworth_pass(fwd[::-1], 100.0, 6.0)[::-1]
    np.testing.assert_allclose(S @ fv, ref, rtol=1e-12, atol=1e-14)
    fe = D.arbitrary_1v(fv)
    assert abs(np.sum(fe) * 12.0 / nvx - 1) < 1e-13 and np.all(fe > 0)
    g = np.random.default_rng(1).normal(size=nvx)
    v = D.arbitrary_1v_vjp(fv, g)
    for i in (0, 7, 24, 47):
        a, b = fv.copy(), fv.copy()
        a[i] += 1e-6
        b[i] -= 1e-6
        fd = (np.dot(g, D.arbitrary_1v(a)) - np.dot(g, D.arbitrary_1v(b))) / 2e-6
        assert abs(fd - v[i]) < 1e-6 * max(1.0, abs(v[i]))
    cfg = decks.deck_fit(nvx=nvx, active=("Te", "Ti", "lam"))
    cfg["parameters"]["electron"]["fe"] = {"active": True, "type": "arbitrary", "dim": 1, "nvx": nvx, "params": {"init_m": 3.0}}
    tp = ThomsonParams(cfg["parameters"], 3, batch=True, activate=True)
    assert tp.fval.shape == (3, nvx) and tp()["electron"]["fe"].shape == (3, nvx)
    spec = tree.get_filter_spec(cfg["parameters"], tp)
    assert [n for n, _ in spec] == [("electron", "Te"), ("electron", "fval"), ("ion-1", "Ti"), ("general", "lam")]
    diff, static = tree.partition(tp, spec)
    flat, unravel = tree.ravel_pytree(diff)
    assert flat.size == 3 * 3 + 3 * nvx
    np.testing.assert_array_equal(flat[3 : 3 + nvx], tp.fval[0])       # lineout-major fval leaves
    back = tree.combine(static, unravel(flat * 1.0))
    np.testing.assert_array_equal(back.fval, tp.fval)
    fitted, n = tp.get_fitted_params(cfg["parameters"])
    assert "f" in fitted["electron"] and n == 4


def test_spherical_harmonics_nn_radial_functions():
    """flm_type "nn" (FLM_NN, spherical_harmonics.py:14-50): two MLPs per harmonic over the radial axis with caller-supplied layer
    weights.  Forward against a torch twin of the reference's expression, the hand-written backward pass (MLP.backward chained
    through 10^-a, the radial interpolation, the floor and the normalisation) against torch autograd of that twin."""
    import torch

    nvx, nvr, width, depth = 32, 24, 8, 3
    rng = np.random.default_rng(3)
    sizes = [1] + [width] * depth + [1]
    mk = lambda scale: {"weights": [rng.normal(0, scale, (sizes[j + 1], sizes[j])) for j in range(depth + 1)],
                        "biases": [rng.normal(0, 0.3, sizes[j + 1]) for j in range(depth + 1)]}
    nn_w = {f"1,{m}": {"flm_mag": mk(0.7), "flm_sign": mk(0.9)} for m in (0, 1)}
    dc = {"nvx": nvx, "dim": 2, "type": "sphericalharmonic", "active": True,
          "params": {"flm_type": "nn", "init_m": 2.4, "Nl": 1, "nvr": nvr, "nn_weights": nn_w}}
    sh = D.SphericalHarmonics(dc)
    theta = sh.get_params()
    nW = sum(w.size for w in nn_w["1,0"]["flm_mag"]["weights"])
    assert theta.size == 2 * 2 * nW + 1 and nW == width + 2 * width * width + width
    np.testing.assert_array_equal(theta[:width], nn_w["1,0"]["flm_mag"]["weights"][0].ravel())   # flm_mag first, layer by layer
    f = sh()
    dv = sh.vx[1] - sh.vx[0]
    assert abs(np.sum(f) * dv * dv - 1.0) < 1e-13 and np.abs(f - f[:, ::-1]).max() > 1e-6 * f.max()

    # torch twin of SphericalHarmonics.__call__ with FLM_NN radial functions (spherical_harmonics.py:42-50, 300-318)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    vr, f00 = T(sh.vr), T(sh.get_f00())
    leaves = []

    def mlp(spec, final):
        h = vr[:, None]
        for j, (W, b) in enumerate(zip(spec["weights"], spec["biases"])):
            Wt = T(W).requires_grad_(True)
            leaves.append(Wt)
            z = h @ Wt.T + T(b)
            h = torch.relu(z) if (j < depth or final == "relu") else torch.tanh(z)
        return h[:, 0]

    def interp(x, xp, fp, right):   # np.interp with a constant to the right of the last node
        i = np.clip(np.searchsorted(xp, x, side="right") - 1, 0, xp.size - 2)
        t = T(np.clip((x - xp[i]) / (xp[i + 1] - xp[i]), 0.0, 1.0))
        v = fp[i] * (1 - t) + fp[i + 1] * t
        return torch.where(T(x) <= xp[-1], v, torch.full_like(v, right))

    q = sh.vr_vxvy.ravel()
    ft = interp(q, sh.vr, f00, 1e-16)
    for m in (0, 1):
        spec = nn_w[f"1,{m}"]
        flm = f00 * 10.0 ** (-mlp(spec["flm_mag"], "relu")) * mlp(spec["flm_sign"], "tanh")
        ft = ft + interp(q, sh.vr, flm, 1e-32) * T(D.real_sph_harm(1, m, sh.phi, sh.th).ravel())
    ft = torch.clamp(ft, min=1e-32)
    ft = (ft / (ft.sum() * dv * dv)).reshape(nvx, nvx)
    np.testing.assert_allclose(f, ft.detach().numpy(), rtol=1e-12, atol=1e-300)
    fbar = rng.standard_normal((nvx, nvx))
    (ft * T(fbar)).sum().backward()
    ref = np.concatenate([w.grad.numpy().ravel() for w in leaves])
    ga = sh.vjp(fbar)
    assert np.max(np.abs(ga[:-1] - ref)) < 1e-10 * np.max(np.abs(ref))
    gfd = sh._vjp_fd(fbar)
    assert abs(ga[-1] - gfd[-1]) < 1e-6 * max(abs(gfd[-1]), 1e-12) and np.max(np.abs(ga[:-1] - gfd[:-1])) < 1e-6 * np.max(np.abs(ref))
    np.testing.assert_array_equal(sh.get_params(), theta)
    # without supplied weights: a documented NumPy initialisation (not the reference's PRNGKey numbers), the reference's layer sizes
    dc["params"].pop("nn_weights")
    sh2 = D.SphericalHarmonics(dc)
    assert sh2.get_params().size == 2 * 2 * (32 + 2 * 32 * 32 + 32) + 1 and np.isfinite(sh2()).all()
