"""Electron distribution functions on the reference's velocity grid (host side, static data).

Mirrors ``tsadar/core/modules/distribution_functions/base.py``: the velocity grid (:149-151), the
DLM table ``f_vx_m`` (:266-272) and ``DLM1V.__call__`` (:277-294).  The reference loads the table
from ``external/numDistFuncs/DLM_x_-3_-10_10_m_-1_2_5.mat``, a file that is not shipped with the
source tree; its content -- the 1-D projection of the unit-normalised 3-D super-Gaussian with
v_th = sqrt(2) -- has a closed form, used here (validated against the reference's golden vector
``tests/test_forward/ThryE-1d.npy`` to 6e-13, see tests/test_oracle_golden.py).
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np
from scipy.special import gamma, gammaincc

M_AXIS = np.linspace(2, 5, 31)


def velocity_grid(nvx: int) -> np.ndarray:
    vmax = 6.0
    dv = 2 * vmax / nvx
    return np.linspace(-vmax + dv / 2, vmax - dv / 2, nvx)


def _projected_super_gaussian(x, m):
    # f1(x) = 2 pi int_|x|^inf u f3(u) du,  f3(u) = m/(4 pi a^3 Gamma(3/m)) exp(-(u/a)^m)
    a = np.sqrt(3.0 * gamma(3.0 / m) / (2.0 * gamma(5.0 / m))) * np.sqrt(2.0)
    norm = m / (4.0 * np.pi * a**3 * gamma(3.0 / m))
    return 2.0 * np.pi * norm * (a * a / m) * gamma(2.0 / m) * gammaincc(2.0 / m, (np.abs(x) / a) ** m)


@lru_cache(maxsize=8)
def dlm_table(nvx: int) -> np.ndarray:
    """[nvx, 31]: the reference's 20001-point table (x = linspace(-10, 10, 20001)) linearly
    interpolated onto vx -- only the two bracketing table nodes of every vx are evaluated."""
    vx = velocity_grid(nvx)
    grid = np.linspace(-10, 10, 20001)
    hi = np.clip(np.searchsorted(grid, vx, side="right"), 1, grid.size - 1)
    x0, x1 = grid[hi - 1], grid[hi]
    w = (vx - x0) / (x1 - x0)
    cols = []
    for m in M_AXIS:
        f0, f1 = _projected_super_gaussian(x0, m), _projected_super_gaussian(x1, m)
        cols.append(f0 + w * (f1 - f0))
    return np.ascontiguousarray(np.stack(cols, axis=1))


def dlm(m: float, nvx: int) -> np.ndarray:
    """DLM1V.__call__: fe(vx) for super-Gaussian order m (linear in m between table columns)."""
    vx = velocity_grid(nvx)
    tab = dlm_table(nvx)
    k = int(np.clip(np.searchsorted(M_AXIS, m, side="right"), 1, M_AXIS.size - 1))
    t = (m - M_AXIS[k - 1]) / (M_AXIS[k] - M_AXIS[k - 1])
    t = min(max(t, 0.0), 1.0)
    f = tab[:, k - 1] + t * (tab[:, k] - tab[:, k - 1])
    return f / np.sum(f) / (vx[1] - vx[0])


def maxwellian(nvx: int) -> np.ndarray:
    """The reference's Maxwellian is DLM m = 2 (tests/configs/epw_inputs.yaml:13-21)."""
    return dlm(2.0, nvx)


def arbitrary_2v_init(m: float, nvx: int, learn_log: bool) -> np.ndarray:
    """Arbitrary2V.init_dlm (base.py:375-408): the stored parameter array fval [nvx, nvx] -- the square root of the
    (optionally -log10 of the) grid-normalised 2-D super-Gaussian of order m with v_th = sqrt(2)."""
    vx = velocity_grid(nvx)
    vth = np.sqrt(2.0)
    alpha = np.sqrt(3.0 * gamma(3.0 / m) / 2.0 / gamma(5.0 / m))
    cst = m / (4.0 * np.pi * alpha**3.0 * gamma(3.0 / m))
    f = cst / vth**3.0 * np.exp(-((np.sqrt(vx[:, None] ** 2.0 + vx[None, :] ** 2.0) / alpha / vth) ** m))
    f = f / np.sum(f) / (vx[1] - vx[0]) ** 2.0
    if learn_log:
        f = -np.log10(f)
    return np.sqrt(f)


def arbitrary_2v(fval: np.ndarray, learn_log: bool) -> np.ndarray:
    """Arbitrary2V.__call__ (base.py:413-427): fval -> normalised f_e(vx, vy)."""
    nvx = fval.shape[-1]
    vx = velocity_grid(nvx)
    f = np.asarray(fval, dtype=np.float64) ** 2.0
    if learn_log:
        f = np.power(10.0, -f)
    return f / np.sum(f) / (vx[1] - vx[0]) ** 2.0


def arbitrary_2v_vjp(fval: np.ndarray, learn_log: bool, fe_bar: np.ndarray) -> np.ndarray:
    """Transpose of the Jacobian of :func:`arbitrary_2v`: d loss / d f_e[nvx, nvx] -> d loss / d fval[nvx, nvx]."""
    nvx = fval.shape[-1]
    vx = velocity_grid(nvx)
    c = 1.0 / (vx[1] - vx[0]) ** 2.0
    fv = np.asarray(fval, dtype=np.float64)
    f = fv**2.0
    if learn_log:
        f = np.power(10.0, -f)
    tot = np.sum(f)
    f_bar = c * (fe_bar / tot - np.sum(fe_bar * f) / tot**2)  # F = c f / sum f
    if learn_log:
        return f_bar * (-np.log(10.0) * f) * 2.0 * fv
    return f_bar * 2.0 * fv


# ---------------------------------------------------------------------------------------------
# SphericalHarmonics (spherical_harmonics.py:150-318): f(vx, vy) = f00(|v|) + sum_{l<=Nl, m<=l} f_lm(|v|) Re Y_l^m.
# The reference evaluates jax.scipy.special.sph_harm(m, l, azimuth = arccos(vy/|vy|), polar = arctan2(vy, vx)); with
# the azimuth in {0, pi} and the associated Legendre functions built from sqrt(1 - cos^2) (Condon-Shortley phase),
# Re Y_l^m = N_lm P_l^m(cos th) cos(m az), P_l^m evaluated with sin = |sin th|.  Parameters: the super-Gaussian
# order m of f00 (sigmoid-activated like every other leaf) and the radial functions f_lm: "mora-yahi" (one
# log10 gradient length per harmonic) or "arbitrary" (two free arrays per harmonic).  flm_type "nn" needs the
# reference's equinox MLP initialisation (jax PRNG) and is not built.
# ---------------------------------------------------------------------------------------------
def _assoc_legendre(l: int, m: int, x: np.ndarray) -> np.ndarray:
    """P_l^m(x) with the Condon-Shortley phase, sin = sqrt(1 - x^2) >= 0 (scipy.special.lpmv convention)."""
    from scipy.special import lpmv

    return lpmv(m, l, x)


def real_sph_harm(l: int, m: int, azimuth: np.ndarray, polar: np.ndarray) -> np.ndarray:
    from math import factorial

    norm = np.sqrt((2 * l + 1) / (4 * np.pi) * factorial(l - m) / factorial(l + m))
    return norm * _assoc_legendre(l, m, np.cos(polar)) * np.cos(m * azimuth)


class MLP:
    """eqx.nn.MLP(in_size=1, out_size=1, width_size, depth) as the reference's FLM_NN builds it (spherical_harmonics.py:37-38):
    ``depth`` hidden Linear layers of ``width`` units plus the output layer, relu between the layers, ``final`` ("relu" / "tanh")
    after the last, evaluated at every node of a 1-D axis.  Weights are caller-supplied (``weights[j]`` [out, in],
    ``biases[j]`` [out]); the trainable leaves are the WEIGHTS only, as in the reference's filter spec (base.py:503-516).
    ``backward`` is the hand-written reverse pass: d loss / d output[n] -> d loss / d weights[j]."""

    def __init__(self, weights, biases, final: str):
        self.W = [np.array(w, dtype=np.float64) for w in weights]
        self.b = [np.array(b, dtype=np.float64).reshape(-1) for b in biases]
        assert len(self.W) == len(self.b) and self.W[0].shape[1] == 1 and self.W[-1].shape[0] == 1
        for j in range(len(self.W)):
            assert self.W[j].shape[0] == self.b[j].size and (j == 0 or self.W[j].shape[1] == self.W[j - 1].shape[0])
        assert final in ("relu", "tanh")
        self.final = final

    @staticmethod
    def default(width: int, depth: int, final: str, seed: int):
        """Deterministic NumPy initialisation, uniform(-1/sqrt(in), 1/sqrt(in)) like equinox's Linear -- NOT the reference's numbers:
        those come from jax.random.PRNGKey(0) / (42), which cannot be reproduced without JAX.  Pass the weights of a reference
        run through ``params["nn_weights"]`` to continue from them."""
        rng = np.random.default_rng(seed)
        sizes = [1] + [width] * depth + [1]
        W = [rng.uniform(-1, 1, (sizes[j + 1], sizes[j])) / np.sqrt(sizes[j]) for j in range(len(sizes) - 1)]
        b = [rng.uniform(-1, 1, sizes[j + 1]) / np.sqrt(sizes[j]) for j in range(len(sizes) - 1)]
        return MLP(W, b, final)

    def forward(self, x: np.ndarray, keep: bool = False) -> np.ndarray:
        h = np.asarray(x, dtype=np.float64).reshape(-1, 1)   # [n, 1]
        acts, pre = [h], []
        for j, (W, b) in enumerate(zip(self.W, self.b)):
            z = h @ W.T + b
            pre.append(z)
            last = j == len(self.W) - 1
            h = (np.maximum(z, 0.0) if (not last or self.final == "relu") else np.tanh(z))
            acts.append(h)
        if keep:
            self._acts, self._pre = acts, pre
        return h[:, 0]

    def backward(self, out_bar: np.ndarray):
        """out_bar [n] = d loss / d forward(x)[n] (of the LAST forward(keep=True)) -> [d loss / d W_j]."""
        g = np.asarray(out_bar, dtype=np.float64).reshape(-1, 1)
        grads = [None] * len(self.W)
        for j in range(len(self.W) - 1, -1, -1):
            z, a_in = self._pre[j], self._acts[j]
            last = j == len(self.W) - 1
            dz = g * ((z > 0.0) if (not last or self.final == "relu") else (1.0 - np.tanh(z) ** 2))
            grads[j] = dz.T @ a_in
            g = dz @ self.W[j]
        return grads

    def n_weights(self) -> int:
        return int(sum(w.size for w in self.W))


class SphericalHarmonics:
    """Host mirror of the reference class: ``__call__()`` -> f_e[nvx, nvx]; ``vx``; ``get_unnormed_params()``."""

    def __init__(self, dist_cfg: dict):
        p = dist_cfg["params"]
        self.nvx = int(dist_cfg["nvx"])
        self.vx = velocity_grid(self.nvx)
        vmax = 6.0 * 1.05 * np.sqrt(2.0)
        nvr = int(p["nvr"])
        dvr = vmax / nvr
        self.vr = np.linspace(dvr / 2, vmax - dvr / 2, nvr)
        vx, vy = np.meshgrid(self.vx, self.vx)
        self.th = np.arctan2(vy, vx)
        self.phi = np.arccos(vy / np.abs(vy))
        self.vr_vxvy = np.sqrt(vx**2 + vy**2)
        self.Nl = int(p["Nl"])
        self.m_scale, self.m_shift = 3.0, 2.0
        x = (float(p["init_m"]) - self.m_shift) / self.m_scale
        self.normed_m = np.log(1e-2 + x / (1 - x + 1e-2))
        self.flm_type = str(p.get("flm_type", "arbitrary")).casefold()
        self.flm = {}
        for l in range(1, self.Nl + 1):
            for m in range(l + 1):
                if self.flm_type == "mora-yahi":
                    if l != 1:
                        raise NotImplementedError("Mora-Yahi only supports l=1, m=0 and l=1, m=1")
                    self.flm[(l, m)] = {"log_10_LT": np.log10(p["LTx"] if m == 0 else p["LTy"])}
                elif self.flm_type == "arbitrary":
                    self.flm[(l, m)] = {"flm_sign": np.zeros(nvr), "flm_mag": np.zeros(nvr)}
                elif self.flm_type == "nn":
                    # FLM_NN (spherical_harmonics.py:14-50): two MLPs per harmonic over the radial axis, magnitude (relu end,
                    # -> f00 10^-out) and sign (tanh end).  Layer weights and biases come from the deck (params["nn_weights"]
                    # ["l,m"] = {"flm_mag": {"weights": [...], "biases": [...]}, "flm_sign": {...}}) or, absent that, from a
                    # documented NumPy initialisation (MLP.default): the reference's PRNGKey(0) / (42) numbers need JAX.
                    given = (p.get("nn_weights") or {}).get(f"{l},{m}")
                    nets = {}
                    for k, (name, final, seed) in enumerate((("flm_mag", "relu", 0), ("flm_sign", "tanh", 42))):
                        if given is not None:
                            nets[name] = MLP(given[name]["weights"], given[name]["biases"], final)
                        else:
                            nets[name] = MLP.default(int(p.get("nn_width", 32)), int(p.get("nn_depth", 3)), final, seed + 1000 * (l * 10 + m))
                    self.flm[(l, m)] = nets
                else:
                    raise NotImplementedError(f"Unknown flm_type: {p.get('flm_type')}")

    # trainable leaves in the reference's pytree order (get_distribution_filter_spec, base.py:484-523): the radial
    # functions of every harmonic (log_10_LT for Mora-Yahi; flm_sign then flm_mag for the free radial functions), then
    # normed_m
    # (flm_type "nn": the layer weights of flm_mag, then of flm_sign -- the field order of FLM_NN --, biases are static)
    def get_params(self) -> np.ndarray:
        parts = []
        for key in sorted(self.flm):
            prm = self.flm[key]
            if self.flm_type == "nn":
                parts += [w.ravel() for name in ("flm_mag", "flm_sign") for w in prm[name].W]
            else:
                parts += [np.atleast_1d(prm["log_10_LT"])] if self.flm_type == "mora-yahi" else [prm["flm_sign"], prm["flm_mag"]]
        return np.concatenate(parts + [np.atleast_1d(self.normed_m)]).astype(np.float64)

    def set_params(self, vec) -> None:
        vec = np.asarray(vec, dtype=np.float64).ravel()
        o = 0
        for key in sorted(self.flm):
            prm = self.flm[key]
            if self.flm_type == "mora-yahi":
                prm["log_10_LT"] = float(vec[o]); o += 1
            elif self.flm_type == "nn":
                for name in ("flm_mag", "flm_sign"):
                    for j, w in enumerate(prm[name].W):
                        prm[name].W[j] = vec[o : o + w.size].reshape(w.shape).copy(); o += w.size
            else:
                n = self.vr.size
                prm["flm_sign"] = vec[o : o + n].copy(); o += n
                prm["flm_mag"] = vec[o : o + n].copy(); o += n
        self.normed_m = float(vec[o])

    def vjp(self, fe_bar: np.ndarray, step: float = 1e-6) -> np.ndarray:
        """d loss / d f_e[nvx, nvx] -> d loss / d get_params().  Free radial functions ("arbitrary": 2 nvr values per
        harmonic) are chained analytically -- normalisation, floor, radial interpolation (transposed), 10^mag * sign,
        sigmoid / tanh, smoothing (transposed) --; the super-Gaussian order of f00 and the Mora-Yahi gradient lengths (one
        scalar each) by central differences of the generator."""
        if self.flm_type == "nn":
            return self._vjp_nn(fe_bar, step)
        if self.flm_type != "arbitrary":
            return self._vjp_fd(fe_bar, step)
        theta = self.get_params()
        out = np.zeros_like(theta)
        f00 = self.get_f00()
        nvr = self.vr.size
        # forward pieces
        rad, aux = {}, {}
        w = np.hanning(nvr // 4)
        w = w / w.sum()
        M = np.stack([np.convolve(e, w, mode="same") for e in np.eye(nvr)], axis=1)   # sm(a) = M @ a
        f = np.interp(self.vr_vxvy, self.vr, f00, right=1e-16)
        for key in sorted(self.flm):
            prm = self.flm[key]
            v, u = M @ prm["flm_sign"], M @ prm["flm_mag"]
            sg = 1.0 / (1.0 + np.exp(-u))
            p10, th = 10.0 ** (-10.0 * sg), np.tanh(v)
            rad[key], aux[key] = p10 * th, (p10, th, sg)
            f = f + np.interp(self.vr_vxvy, self.vr, rad[key], right=1e-32) * real_sph_harm(key[0], key[1], self.phi, self.th)
        live = f > 1e-32
        fc = np.maximum(f, 1e-32)
        tot, c = np.sum(fc), 1.0 / (self.vx[1] - self.vx[0]) ** 2
        f_bar = c * (fe_bar / tot - np.sum(fe_bar * fc) / tot**2) * live
        # transposed linear interpolation onto the radial nodes (np.interp: left of the first node = its value, right of
        # the last = the constant `right`)
        q = self.vr_vxvy.ravel()
        i = np.clip(np.searchsorted(self.vr, q, side="right") - 1, 0, nvr - 2)
        t = np.clip((q - self.vr[i]) / (self.vr[i + 1] - self.vr[i]), 0.0, 1.0)
        inside = q <= self.vr[-1]
        o = 0
        for key in sorted(self.flm):
            g = (f_bar * real_sph_harm(key[0], key[1], self.phi, self.th)).ravel() * inside
            r_bar = np.bincount(i, weights=g * (1.0 - t), minlength=nvr) + np.bincount(i + 1, weights=g * t, minlength=nvr)
            p10, th, sg = aux[key]
            out[o : o + nvr] = M.T @ (r_bar * p10 * (1.0 - th**2))                                       # flm_sign
            out[o + nvr : o + 2 * nvr] = M.T @ (r_bar * rad[key] * np.log(10.0) * (-10.0) * sg * (1.0 - sg))   # flm_mag
            o += 2 * nvr
        # the order of f00: one scalar, central difference
        vals = []
        for sgn in (+1.0, -1.0):
            self.normed_m = theta[-1] + sgn * step
            vals.append(self())
        self.normed_m = theta[-1]
        out[-1] = np.sum(fe_bar * (vals[0] - vals[1])) / (2.0 * step)
        return out

    def _vjp_nn(self, fe_bar: np.ndarray, step: float = 1e-6) -> np.ndarray:
        """flm_type "nn": normalisation, floor and the transposed radial interpolation as for the free radial functions, then
        flm = f00 10^(-a) s with a = flm_mag(vr) (relu end), s = flm_sign(vr) (tanh end): the hand-written backward pass of the
        two MLPs (MLP.backward) gives d loss / d layer weights; the order of f00 (one scalar) by central differences."""
        theta = self.get_params()
        out = np.zeros_like(theta)
        f00 = self.get_f00()
        nvr = self.vr.size
        f = np.interp(self.vr_vxvy, self.vr, f00, right=1e-16)
        fw = {}
        for key in sorted(self.flm):
            a = self.flm[key]["flm_mag"].forward(self.vr, keep=True)
            sg = self.flm[key]["flm_sign"].forward(self.vr, keep=True)
            mag = f00 * 10.0 ** (-a)
            fw[key] = (mag, sg)
            f = f + np.interp(self.vr_vxvy, self.vr, mag * sg, right=1e-32) * real_sph_harm(key[0], key[1], self.phi, self.th)
        live = f > 1e-32
        fc = np.maximum(f, 1e-32)
        tot, c = np.sum(fc), 1.0 / (self.vx[1] - self.vx[0]) ** 2
        f_bar = c * (fe_bar / tot - np.sum(fe_bar * fc) / tot**2) * live
        q = self.vr_vxvy.ravel()
        i = np.clip(np.searchsorted(self.vr, q, side="right") - 1, 0, nvr - 2)
        t = np.clip((q - self.vr[i]) / (self.vr[i + 1] - self.vr[i]), 0.0, 1.0)
        inside = q <= self.vr[-1]
        o = 0
        for key in sorted(self.flm):
            g = (f_bar * real_sph_harm(key[0], key[1], self.phi, self.th)).ravel() * inside
            r_bar = np.bincount(i, weights=g * (1.0 - t), minlength=nvr) + np.bincount(i + 1, weights=g * t, minlength=nvr)
            mag, sg = fw[key]
            for name, seed in (("flm_mag", r_bar * sg * mag * (-np.log(10.0))), ("flm_sign", r_bar * mag)):
                for gw in self.flm[key][name].backward(seed):
                    out[o : o + gw.size] = gw.ravel(); o += gw.size
        vals = []
        for sgn in (+1.0, -1.0):
            self.normed_m = theta[-1] + sgn * step
            vals.append(self())
        self.normed_m = theta[-1]
        out[-1] = np.sum(fe_bar * (vals[0] - vals[1])) / (2.0 * step)
        return out

    def _vjp_fd(self, fe_bar: np.ndarray, step: float = 1e-6) -> np.ndarray:
        """Central differences of the generator, 2 evaluations per parameter (cross-check of vjp; Mora-Yahi)."""
        theta = self.get_params()
        out = np.zeros_like(theta)
        for i in range(theta.size):
            vals = []
            for sgn in (+1.0, -1.0):
                t = theta.copy()
                t[i] += sgn * step
                self.set_params(t)
                vals.append(self())
            out[i] = np.sum(fe_bar * (vals[0] - vals[1])) / (2.0 * step)
        self.set_params(theta)
        return out

    def get_unnormed_m(self) -> float:
        return 1.0 / (1.0 + np.exp(-self.normed_m)) * self.m_scale + self.m_shift

    def get_f00(self) -> np.ndarray:
        m = self.get_unnormed_m()
        v0 = 1.0 / np.sqrt(gamma(5.0 / m) / 3.0 / gamma(3.0 / m))
        f00 = m / (4 * np.pi * gamma(3.0 / m)) / v0**3.0 * np.exp(-((self.vr / v0) ** m))
        return f00 / (np.sum(f00 * 4 * np.pi * self.vr**2.0) * (self.vr[1] - self.vr[0]))

    def radial(self, l: int, m: int, f00: np.ndarray) -> np.ndarray:
        prm = self.flm[(l, m)]
        if self.flm_type == "mora-yahi":  # FLM_MY.__call__, Mora & Yahi 1982 eq. 3
            mf = self.get_unnormed_m()
            ve = gamma(5.0 / mf) / 3 / gamma(3.0 / mf)
            lam_v = (self.vr / ve) ** 4.0
            coeff = (mf / 2 * self.vr**mf - 5 * mf / 12 * gamma(8 / mf) / gamma(6 / mf) * self.vr ** (mf - 2) - 1.5) * lam_v
            return coeff / 10 ** prm["log_10_LT"] * f00
        if self.flm_type == "nn":   # FLM_NN.__call__ (spherical_harmonics.py:42-50)
            return f00 * 10.0 ** (-prm["flm_mag"].forward(self.vr)) * prm["flm_sign"].forward(self.vr)
        nvr = self.vr.size  # ArbitraryVr.__call__
        w = np.hanning(nvr // 4)
        w = w / w.sum()
        sm = lambda a: np.convolve(a, w, mode="same")
        sign = np.tanh(sm(prm["flm_sign"]))
        mag = -(1.0 / (1.0 + np.exp(-sm(prm["flm_mag"])))) * 10
        return 10**mag * sign

    def get_unnormed_params(self) -> dict:
        f00 = self.get_f00()
        out = {0: {0: f00}, 1: {}}
        for (l, m) in self.flm:
            out.setdefault(l, {})[m] = self.radial(l, m, f00)
        return {"flm": out}

    def __call__(self) -> np.ndarray:
        f00 = self.get_f00()
        f = np.interp(self.vr_vxvy, self.vr, f00, right=1e-16)
        for (l, m) in self.flm:
            flm = np.interp(self.vr_vxvy, self.vr, self.radial(l, m, f00), right=1e-32)
            f = f + flm * real_sph_harm(l, m, self.phi, self.th)
        f = np.maximum(f, 1e-32)
        return f / (np.sum(f) * (self.vx[1] - self.vx[0]) ** 2)


# ---------------------------------------------------------------------------------------------
# Arbitrary1V (base.py:157-204): free-form 1-D distribution function.  Stored leaf: fval [nvx];
# f_e = normalise(10 ** -((7 smooth(fval)) ** 2)), smooth = forward-backward second-order Butterworth filter
# (base.py:41-97; f_sampling = 100, f_cutoff = 6).  The filter is linear and homogeneous in its input (the scan is
# seeded with the first two samples), so it is applied as a constant [nvx, nvx] matrix and its adjoint is the transpose.
# ---------------------------------------------------------------------------------------------
def _butterworth_pass(signal: np.ndarray, f_sampling: float, f_cutoff: float) -> np.ndarray:
    ff = f_cutoff / f_sampling
    ita = 1.0 / np.tan(np.pi * ff)
    q = np.sqrt(2.0)
    b0 = 1.0 / (1.0 + q * ita + ita**2)
    b1, b2 = 2 * b0, b0
    a1 = 2.0 * (ita**2 - 1.0) * b0
    a2 = -(1.0 - q * ita + ita**2) * b0
    x1, x2, y1, y2 = signal[1], signal[0], signal[1], signal[0]
    out = []
    for x in signal[2:]:
        y = b0 * x + b1 * x1 + b2 * x2 + a1 * y1 + a2 * y2
        x1, x2, y1, y2 = x, x1, y, y1
        out.append(y)
    out = np.array(out)
    return np.concatenate((out[0:1], out[0:1], out))


@lru_cache(maxsize=8)
def butterworth_matrix(n: int, f_sampling: float = 100.0, f_cutoff: float = 6.0) -> np.ndarray:
    """S with smooth(x) = S @ x for method "forward_backward" (forward pass, then the same pass on the flipped
    signal, flipped back)."""
    S = np.zeros((n, n))
    eye = np.eye(n)
    for j in range(n):
        fwd = _butterworth_pass(eye[:, j], f_sampling, f_cutoff)
        S[:, j] = _butterworth_pass(fwd[::-1], f_sampling, f_cutoff)[::-1]
    return S


def arbitrary_1v_init(m: float, nvx: int) -> np.ndarray:
    """Arbitrary1V.init_dlm (base.py:188-196): fval of a super-Gaussian of order m (v_th = 1 here)."""
    vx = velocity_grid(nvx)
    alpha = np.sqrt(3.0 * gamma(3.0 / m) / 2.0 / gamma(5.0 / m))
    cst = m / (4.0 * np.pi * alpha**3.0 * gamma(3.0 / m))
    f = cst * np.exp(-(np.abs(vx / alpha) ** m))
    f = f / np.sum(f) / (vx[1] - vx[0])
    return np.sqrt(-np.log10(f)) / 7.0


def arbitrary_1v(fval: np.ndarray) -> np.ndarray:
    """Arbitrary1V.__call__ (base.py:201-204) for fval [..., nvx] -> f_e [..., nvx]."""
    fval = np.asarray(fval, dtype=np.float64)
    nvx = fval.shape[-1]
    dv = 12.0 / nvx
    u = fval @ butterworth_matrix(nvx).T
    f = np.power(10.0, -((7.0 * u) ** 2.0))
    return f / np.sum(f, axis=-1, keepdims=True) / dv


def arbitrary_1v_vjp(fval: np.ndarray, g_fe: np.ndarray) -> np.ndarray:
    """d loss / d fval given g_fe = d loss / d f_e (chain rule of arbitrary_1v)."""
    fval = np.asarray(fval, dtype=np.float64)
    nvx = fval.shape[-1]
    dv = 12.0 / nvx
    S = butterworth_matrix(nvx)
    u = fval @ S.T
    f = np.power(10.0, -((7.0 * u) ** 2.0))
    Z = np.sum(f, axis=-1, keepdims=True)
    fe = f / Z / dv
    g_f = (g_fe / dv - np.sum(g_fe * fe, axis=-1, keepdims=True)) / Z
    g_u = g_f * (-np.log(10.0) * f) * (98.0 * u)
    return g_u @ S
