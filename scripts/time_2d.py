"""Timing of tsff_form_factor_2d at the reference ARTS size (128^2 f_e, 241 angles x 1024 lambda) and at
BASELINE config 4 (256^2 f_e, 512 angles x 1024 lambda)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tsadar_amd import synthetic as S
from tsadar_amd.engine import Engine
from tsadar_amd import ThomsonParams

for nv, na in ((128, 241), (256, 512)):
    cfg = S.baseline_deck()
    sa = dict(sa=np.linspace(19, 139, na), weights=np.ones((1, na)) / na)
    eng = Engine(cfg, sa, activate=False)
    tp = ThomsonParams(cfg["parameters"], 1, batch=True, activate=False)
    vx = np.linspace(-6 + 6.0 / nv, 6 - 6.0 / nv, nv)
    X, Y = np.meshgrid(vx, vx, indexing="ij")
    f = np.exp(-((X / 1.2) ** 2 + (Y / 0.9) ** 2) ** 1.3 / 2)
    f /= f.sum() * (vx[1] - vx[0]) ** 2
    fd = eng.dev(f)
    P = eng.form_factor_2d(0, tp.physical_matrix(), fd, 10.0, 20.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    P = eng.form_factor_2d(0, tp.physical_matrix(), fd, 10.0, 20.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    npt = 1024 * na
    print(f"nv {nv} angles {na}: {dt*1e3:.1f} ms per EPW image ({npt} points, {npt*nv*nv/dt/1e9:.1f} G bicubic evaluations/s), finite {bool(torch.isfinite(P).all())}")
    Pbar = torch.randn_like(P)
    for want_table in (False, True):
        eng.form_factor_2d_grad(0, tp.physical_matrix(), fd, Pbar, 10.0, 20.0, want_table=want_table)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gp, gf = eng.form_factor_2d_grad(0, tp.physical_matrix(), fd, Pbar, 10.0, 20.0, want_table=want_table)
        torch.cuda.synchronize()
        da = time.perf_counter() - t0
        print(f"   adjoint (table adjoint {want_table}): {da*1e3:.1f} ms = {da/dt:.2f} forwards, finite "
              f"{bool(torch.isfinite(gp).all()) and (gf is None or bool(torch.isfinite(gf).all()))}", flush=True)
    # the fit-loop form: the forward keeps the projection records, the adjoint does no sampling
    def _t(fn, n=2):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    ds = _t(lambda: eng.form_factor_2d(0, tp.physical_matrix(), fd, 10.0, 20.0, save=True))
    da = _t(lambda: eng.form_factor_2d_grad(0, tp.physical_matrix(), fd, Pbar, 10.0, 20.0, want_table=False, use_saved=True))
    dt_ = _t(lambda: eng.form_factor_2d_grad(0, tp.physical_matrix(), fd, Pbar, 10.0, 20.0, want_table=True, use_saved=True))
    print(f"   fit-loop form: forward + records {ds:.1f} ms, adjoint from records {da:.1f} ms (parameters), {dt_:.1f} ms (with table adjoint)", flush=True)
