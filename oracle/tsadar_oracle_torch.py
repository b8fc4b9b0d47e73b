"""Gradient oracle: the same restatement as ``tsadar_oracle.py`` written with torch float64 CPU
tensors so that ``torch.autograd`` plays the role JAX autodiff plays in the reference
(``eqx.filter_value_and_grad(__loss__)``, tsadar/inverse/loss_function.py:108).

TEST INFRASTRUCTURE ONLY (see the header of tsadar_oracle.py).  The forward values of this twin are
checked against the NumPy oracle (tests/test_oracle_torch.py), which is itself pinned to the
reference's golden vector; its gradients are checked against finite differences of the NumPy
oracle.  Sub-gradient conventions are those of JAX: ``max`` -> one-hot at the arg-max (torch.amax
splits ties evenly; ties have measure zero), linear interpolation -> slope of the active segment,
``where`` masks -> pass-through.
"""
from __future__ import annotations

import numpy as np
import torch

from . import tsadar_oracle as orc

DT = torch.float64


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=DT)


def interp_linear(x, xp, fp, left=None, right=None):
    """jnp.interp (see tsadar_oracle.interp_linear)."""
    xp, fp = _t(xp), _t(fp)
    i = torch.clamp(torch.searchsorted(xp, x.detach().contiguous(), right=True), 1, xp.numel() - 1)
    dx = xp[i] - xp[i - 1]
    df = fp[i] - fp[i - 1]
    f = fp[i - 1] + ((x - xp[i - 1]) / dx) * df
    lo = fp[0] if left is None else left
    hi = fp[-1] if right is None else right
    f = torch.where(x < xp[0], lo, f)
    f = torch.where(x > xp[-1], hi, f)
    return f


def hermite_slopes(x, f):
    d = (f[1:] - f[:-1]) / (x[1:] - x[:-1])
    return torch.cat([d[:1], 0.5 * (d[:-1] + d[1:]), d[-1:]])


def interp_hermite(xq, x, f, lo, hi):
    x = _t(x)
    fx = hermite_slopes(x, f)
    i = torch.clamp(torch.searchsorted(x, xq.detach().contiguous(), right=True), 1, x.numel() - 1)
    dx = x[i] - x[i - 1]
    t = (xq - x[i - 1]) / dx
    f0, f1 = f[i - 1], f[i]
    m0, m1 = fx[i - 1] * dx, fx[i] * dx
    c2 = -3 * f0 + 3 * f1 - 2 * m0 - m1
    c3 = 2 * f0 - 2 * f1 + m0 + m1
    fq = f0 + t * (m0 + t * (c2 + t * c3))
    fq = torch.where(xq < x[0], torch.full_like(fq, lo), fq)
    fq = torch.where(xq > x[-1], torch.full_like(fq, hi), fq)
    return fq


def gradient_uniform(f, h):
    return torch.cat([((f[1] - f[0]) / h)[None], (f[2:] - f[:-2]) / (2 * h), ((f[-1] - f[-2]) / h)[None]])


def chi_table(vx, fe):
    """form_factor.py:263-268 + ratintn.py (differentiable in fe)."""
    xi1, xi2 = (_t(a) for a in orc.xi_grids())
    ratmod = torch.exp(interp_hermite(xi1, vx, torch.log(fe), -50.0, -50.0))
    ratdf = gradient_uniform(ratmod, xi1[1] - xi1[0])
    g = xi1[None, :] - xi2[:, None]
    f = ratdf
    fdif = f[1:-1] - f[0:-2]
    gdif = g[:, 1:-1] - g[:, 0:-2]
    fav = 0.5 * (f[1:-1] + f[0:-2])
    gav = 0.5 * (g[:, 1:-1] + g[:, 0:-2])
    tmp = fav * gdif - gav * fdif
    rfn = fdif / gdif + tmp * torch.log(torch.abs((gav + 0.5 * gdif) / (gav - 0.5 * gdif))) / gdif**2
    zdif = xi1[1:-1] - xi1[0:-2]
    return torch.sum(rfn * zdif, dim=1)


def dlm_fe(m, nvx):
    """DLM1V.__call__ (base.py:277-294), differentiable in m."""
    vx = orc.velocity_grid(nvx)
    tab = _t(orc.dlm_table(nvx))
    max_ = _t(orc.DLM_M_AXIS)
    k = int(np.clip(np.searchsorted(orc.DLM_M_AXIS, float(m.detach()), side="right"), 1, 30))
    t = (m - max_[k - 1]) / (max_[k] - max_[k - 1])
    f = tab[:, k - 1] + t * (tab[:, k] - tab[:, k - 1])
    if float(m.detach()) < 2.0:
        f = tab[:, 0] + 0.0 * m
    if float(m.detach()) > 5.0:
        f = tab[:, -1] + 0.0 * m
    return f / torch.sum(f) / (vx[1] - vx[0])


def physical_params(cfg_params, normed, activate=True):
    """ts_params.py:583-603 for tensors ``normed[name]`` of shape [B]."""
    def act(active, x):
        return torch.sigmoid(x) if (activate and active) else x

    el = cfg_params["electron"]
    phys = {}
    for k in ["Te", "ne"]:
        phys[k] = act(el[k]["active"], normed[k]) * (el[k]["ub"] - el[k]["lb"]) + el[k]["lb"]
    if "m" in normed:
        phys["m"] = act(el["fe"].get("active", False), normed["m"]) * 3.0 + 2.0
    species = orc.ion_species(cfg_params)
    fsum = 0.0
    for s, sp in enumerate(species):
        ic = cfg_params[sp]
        for k in ["Ti", "Z"]:
            phys[f"{k}_{s+1}"] = act(ic[k]["active"], normed[f"{k}_{s+1}"]) * (ic[k]["ub"] - ic[k]["lb"]) + ic[k]["lb"]
        phys[f"A_{s+1}"] = normed[f"A_{s+1}"]
        phys[f"fract_{s+1}"] = act(ic["fract"]["active"], normed[f"fract_{s+1}"])
        if s > 0 and ic["Ti"].get("same", False):
            phys[f"Ti_{s+1}"] = phys["Ti_1"]
        fsum = fsum + phys[f"fract_{s+1}"]
    for s in range(len(species)):
        phys[f"fract_{s+1}"] = phys[f"fract_{s+1}"] / fsum
    g = cfg_params["general"]
    for k in orc.GENERAL_KEYS:
        phys[k] = act(g[k]["active"], normed[k]) * (g[k]["ub"] - g[k]["lb"]) + g[k]["lb"]
    return phys


def form_factor(lam_range, npts, lam_shift, sa_deg, G, p, vx, fe, W):
    """form_factor.py:163-298 for one lineout (tensors; see tsadar_oracle.form_factor)."""
    xi1, xi2 = orc.xi_grids()
    zr_tab, zi_tab = orc.zprime_tables()
    lam_axis = np.linspace(lam_range[0], lam_range[1], npts)
    omgL_num = 2 * np.pi * 1e7 * orc.C
    omgs = _t(2e7 * np.pi * orc.C / lam_axis)[None, :, None]
    lin = lambda v: (1 - v / 200) + (torch.arange(G, dtype=DT) * ((v / 100) / (G - 1)) if G > 1 else torch.zeros(1, dtype=DT))
    ne = 1.0e20 * p["ne"] * lin(p["ne_gradient"])
    Te = p["Te"] * lin(p["Te_gradient"])
    lam = p["lam"] + lam_shift
    A = torch.stack([_t(a) for a in p["A"]])
    Z = torch.stack(list(p["Z"])).reshape(1, 1, 1, -1)
    Ti = torch.stack(list(p["Ti"]))
    fract = torch.stack(list(p["fract"])).reshape(1, 1, 1, -1)
    Va = p["Va"] * 1e6
    ud = p["ud"] * 1e6
    Mi = (A * orc.MP).reshape(1, 1, 1, -1)
    sarad = _t(np.asarray(sa_deg) * np.pi / 180).reshape(1, 1, -1)
    omgL = omgL_num / lam
    omgpe = orc.C0 * torch.sqrt(ne[:, None, None])
    omg = omgs - omgL
    ks = torch.sqrt(omgs**2 - omgpe**2) / orc.C
    kL = torch.sqrt(omgL**2 - omgpe**2) / orc.C
    k = torch.sqrt(ks**2 + kL**2 - 2 * ks * kL * torch.cos(sarad))
    omgdop = omg - k * Va
    vTe = torch.sqrt(Te[:, None, None] / orc.ME)
    klde = (vTe / omgpe) * k
    Zbar = torch.sum(Z * fract)
    ni = fract * ne[:, None, None, None] / Zbar
    omgpi = orc.C0 * Z * torch.sqrt(ni * orc.ME / Mi)
    vTi = torch.sqrt(Ti.reshape(1, 1, 1, -1) / Mi)
    kldi = (vTi / omgpi) * k[..., None]
    xii = (1.0 / (np.sqrt(2.0) * vTi)) * (omgdop / k)[..., None]
    ZpiR = interp_linear(xii, xi2, zr_tab, left=xii**-2, right=xii**-2)
    ZpiI = interp_linear(xii, xi2, zi_tab, left=torch.zeros_like(xii), right=torch.zeros_like(xii))
    chiIr = torch.sum(-0.5 / kldi**2 * ZpiR, dim=3)
    chiIi = torch.sum(-0.5 / kldi**2 * ZpiI, dim=3)
    xie = omgdop / (k * vTe) - ud / vTe
    fe_vphi = torch.exp(interp_hermite(xie, vx, torch.log(fe), -50.0, -50.0))
    df = (fe_vphi[:, 1:, :] - fe_vphi[:, :-1, :]) / (xie[:, 1:, :] - xie[:, :-1, :])
    df = torch.cat([df, torch.zeros((G, 1, df.shape[2]), dtype=DT)], dim=1)
    chiEi = np.pi / klde**2 * df
    chiEr = -1.0 / klde**2 * interp_linear(xie, xi2, W)
    epsr = 1.0 + chiEr + chiIr
    epsi = chiEi + chiIi
    eps2 = epsr**2 + epsi**2
    ion_fact = fract * Z**2 / Zbar / vTi
    ion_comp = ion_fact * ((chiEr**2 + chiEi**2)[..., None] * torch.exp(-(xii**2)) / np.sqrt(2 * np.pi))
    ele_comp = ((1.0 + chiIr) ** 2 + chiIi**2) * fe_vphi / vTe
    S_ion = torch.sum(1.0 / k[..., None] * ion_comp / eps2[..., None], dim=3)
    S_ele = 1.0 / k * ele_comp / eps2
    PsOmg = (S_ion + S_ele) * (1 + 2 * omgdop / omgL) * orc.RE**2 * ne[:, None, None]
    lams = 2 * np.pi * orc.C / omgs
    return PsOmg * 2 * np.pi * orc.C / lams**2, lams[0, :, 0]


def _conv_same(x, g):
    n = x.numel()
    full = torch.nn.functional.conv1d(x.reshape(1, 1, -1), g.flip(0).reshape(1, 1, -1), padding=n - 1).reshape(-1)
    c = (n - 1) // 2
    return full[c : c + n]


def _irf(lam, modl, stddev):
    origin = (lam.max() + lam.min()) / 2.0
    g = (1.0 / (stddev * np.sqrt(2.0 * np.pi))) * torch.exp(-((lam - origin) ** 2.0) / (2.0 * stddev**2.0))
    y = _conv_same(modl, g)
    return (torch.amax(modl) / torch.amax(y)) * y


def ts_diag(cfg, sa, normed, batch, activate=True, fe_batch=None):
    """thomson_diagnostic.py:109-142 with tensors; returns (ThryE, ThryI, lamE, lamI) [B, 1024]."""
    cfgp = cfg["parameters"]
    other = cfg["other"]
    ext = other["extraoptions"]
    phys = physical_params(cfgp, normed, activate)
    n_ion = len(orc.ion_species(cfgp))
    B = normed["Te"].numel()
    G = cfgp["general"]["Te_gradient"]["num_grad_points"]
    nvx = cfgp["electron"]["fe"]["nvx"]
    vx = orc.velocity_grid(nvx)
    w_ang = _t(np.asarray(sa["weights"])[0])
    outE, outI, lamE, lamI = [], [], [], []

    def col(name, b):
        v = _t(batch[name])
        if v.ndim == 0:
            return v
        v = v[b] if v.shape[0] == B else v
        return v.reshape(()) if v.ndim == 1 and v.shape[0] == 1 else v

    for b in range(B):
        p = {k: phys[k][b] for k in ["Te", "ne", "lam", "amp1", "amp2", "amp3", "ne_gradient", "Te_gradient", "ud", "Va"]}
        for k in ["Ti", "Z", "A", "fract"]:
            p[k] = [phys[f"{k}_{s+1}"][b] for s in range(n_ion)]
        if fe_batch is not None:
            fe = _t(fe_batch)
            fe = fe[b] if fe.ndim == 2 else fe
        else:
            fe = dlm_fe(phys["m"][b], nvx)
        W = chi_table(vx, fe) if fe.requires_grad else _t(orc.chi_table_cached(vx, fe.detach().numpy()))
        for feature in ("ion", "ele"):
            if not ext["load_ion_spec" if feature == "ion" else "load_ele_spec"]:
                continue
            rng, shift = (other["lamrangI"], 0.0) if feature == "ion" else (other["lamrangE"], cfg["data"]["ele_lam_shift"])
            P, lam_cm = form_factor(rng, other["npts"], shift, sa["sa"], G, p, vx, fe, W)
            lam = lam_cm * 1e7
            modl = torch.sum(torch.mean(P, dim=0) * w_ang, dim=1)
            if feature == "ele":
                filt = other["iawfilter"]
                if filt[0]:
                    fb, fr = filt[3] - filt[2] / 2, filt[3] + filt[2] / 2
                    if other["lamrangE"][0] < fr and other["lamrangE"][1] > fb:
                        modl = torch.where((fb < lam) & (fr > lam), modl * 10.0 ** (-filt[1]), modl)
                y = _irf(lam, modl, other["PhysParams"]["widIRF"]["spect_stddev_ele"])
                y = y.reshape(1024, -1).mean(dim=1)
                lamb = lam.reshape(1024, -1).mean(dim=1)
                y = col("e_amps", b) * y / torch.amax(y)
                y = torch.where(lamb < p["lam"], p["amp1"] * y, p["amp2"] * y)
                outE.append(y + col("noise_e", b))
                lamE.append(lamb)
            else:
                if other["PhysParams"]["widIRF"]["spect_stddev_ion"]:
                    y = _irf(lam, modl, other["PhysParams"]["widIRF"]["spect_stddev_ion"])
                    y = y.reshape(1024, -1).mean(dim=1)
                    lamb = lam.reshape(1024, -1).mean(dim=1)
                    y = p["amp3"] * col("i_amps", b) * y / torch.amax(y)
                else:   # irf.py:82-86: no ion IRF -> ThryI = modlI (un-normalised, un-binned)
                    y, lamb = modl, lam
                outI.append(y + col("noise_i", b))
                lamI.append(lamb)
    z = torch.zeros((B, 1024), dtype=DT)
    st = lambda l: torch.stack(l) if l else z
    return st(outE), st(outI), st(lamE), st(lamI)


def masked_sums(cfg, sa, normed, batch, activate=True, fe_batch=None):
    """Un-normalised masked sums [S_iaw, S_blue, S_red] of the loss functional over the given lineouts,
    the number of fitted entries of each, and the spectra (loss_function.py:190-267 before the mean)."""
    ThryE, ThryI, lamE, lamI = ts_diag(cfg, sa, normed, batch, activate, fe_batch)
    ext = cfg["other"]["extraoptions"]
    method = cfg["optimizer"]["loss_method"]
    iaw, blue, red = orc.fit_masks(cfg, lamE.detach().numpy(), lamI.detach().numpy())

    def fun(d, t):
        d = _t(d)
        if method == "l1":
            return torch.abs(d - t)
        if method == "l2":
            return torch.square(d - t)
        if method == "log-cosh":
            return torch.log(torch.cosh(d - t))
        return t - d * torch.log(t)

    z = torch.zeros((), dtype=DT)
    S = [z, z, z]
    N = [0, 0, 0]
    if ext["fit_IAW"]:
        S[0], N[0] = fun(batch["i_data"], ThryI)[torch.as_tensor(iaw)].sum(), int(iaw.sum())
    if ext["fit_EPWb"]:
        S[1], N[1] = fun(batch["e_data"], ThryE)[torch.as_tensor(blue)].sum(), int(blue.sum())
    if ext["fit_EPWr"]:
        S[2], N[2] = fun(batch["e_data"], ThryE)[torch.as_tensor(red)].sum(), int(red.sum())
    return torch.stack(S), N, ThryE, ThryI


def loss(cfg, sa, normed, batch, i_norm, e_norm, activate=True, fe_batch=None):
    """loss_function.py:364-373 (nanmean over the masked entries of the whole batch)."""
    S, N, ThryE, ThryI = masked_sums(cfg, sa, normed, batch, activate, fe_batch)
    ext = cfg["other"]["extraoptions"]
    dep = cfg["optimizer"]["loss_method"] in ("l1", "l2")  # the other functionals ignore the denominator
    ui, ue = (i_norm**2, e_norm**2) if dep else (1.0, 1.0)
    i_err = S[0] / (N[0] * ui) if ext["fit_IAW"] else torch.zeros((), dtype=DT)
    e_err = torch.zeros((), dtype=DT)
    if ext["fit_EPWb"]:
        e_err = e_err + S[1] / (N[1] * ue)
    if ext["fit_EPWr"]:
        e_err = e_err + S[2] / (N[2] * ue)
        if ext["fit_EPWb"]:
            e_err = e_err * 0.5
    return cfg["data"]["ion_loss_scale"] * i_err + e_err, ThryE, ThryI


def hessian_loss(cfg, sa, normed, batch, activate=True):
    """``LossFunction._loss_for_hess_fn_`` (loss_function.py:173-188): denominators |data| + 1e-10, sum reduce,
    i_error + e_error (e_error halved when both EPW ranges are fitted, :262-264)."""
    ThryE, ThryI, lamE, lamI = ts_diag(cfg, sa, normed, batch, activate)
    ext = cfg["other"]["extraoptions"]
    iaw, blue, red = orc.fit_masks(cfg, lamE.detach().numpy(), lamI.detach().numpy())
    ed, idt = _t(batch["e_data"]), _t(batch["i_data"])
    fe = torch.square(ed - ThryE) / (torch.abs(ed) + 1e-10)
    fi = torch.square(idt - ThryI) / (torch.abs(idt) + 1e-10)
    i_err = fi[torch.as_tensor(iaw)].sum() if ext["fit_IAW"] else torch.zeros((), dtype=DT)
    e_err = torch.zeros((), dtype=DT)
    if ext["fit_EPWb"]:
        e_err = e_err + fe[torch.as_tensor(blue)].sum()
    if ext["fit_EPWr"]:
        e_err = e_err + fe[torch.as_tensor(red)].sum()
        if ext["fit_EPWb"]:
            e_err = e_err * 0.5
    return i_err + e_err


def hessian(cfg, sa, normed_np, batch, names, activate=True):
    """Dense Hessian [P, P] of hessian_loss w.r.t. the leaves ``names`` of a ONE-lineout batch (double backward)."""
    base = {k: _t(v).clone() for k, v in normed_np.items()}

    def f(vec):
        nm = dict(base)
        for i, k in enumerate(names):
            nm[k] = vec[i:i + 1]
        return hessian_loss(cfg, sa, nm, batch, activate)

    x0 = torch.cat([base[k].reshape(1) for k in names])
    return torch.autograd.functional.hessian(f, x0).numpy()


def value_and_grad(cfg, sa, normed_np, batch, i_norm, e_norm, names, activate=True, fe_batch=None):
    """(loss, {name: dloss/d normed[name]  [B]}, ThryE, ThryI) by reverse-mode autodiff."""
    normed = {k: _t(v).clone() for k, v in normed_np.items()}
    for k in names:
        normed[k].requires_grad_(True)
    val, E, I = loss(cfg, sa, normed, batch, i_norm, e_norm, activate, fe_batch)
    grads = torch.autograd.grad(val, [normed[k] for k in names], allow_unused=True)
    out = {k: (g.numpy() if g is not None else np.zeros_like(normed_np[k])) for k, g in zip(names, grads)}
    return float(val.detach()), out, E.detach().numpy(), I.detach().numpy()


def value_and_grad_fe(cfg, sa, normed_np, batch, i_norm, e_norm, names, fe_batch, activate=True):
    """As value_and_grad plus d loss / d fe_batch [B, nvx]: what equinox.filter_value_and_grad yields for the
    leaves of a free-form distribution function before the generator's own chain rule (base.py:157-204)."""
    normed = {k: _t(v).clone() for k, v in normed_np.items()}
    for k in names:
        normed[k].requires_grad_(True)
    fe = _t(np.asarray(fe_batch)).clone().requires_grad_(True)
    val, E, I = loss(cfg, sa, normed, batch, i_norm, e_norm, activate, fe)
    grads = torch.autograd.grad(val, [normed[k] for k in names] + [fe], allow_unused=True)
    out = {k: (g.numpy() if g is not None else np.zeros_like(normed_np[k])) for k, g in zip(names, grads[:-1])}
    return float(val.detach()), out, grads[-1].numpy(), E.detach().numpy(), I.detach().numpy()
