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
