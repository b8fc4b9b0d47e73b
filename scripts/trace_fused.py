#!/usr/bin/env python
"""Workgroup timeline of k_spectrum_fused (measurement build -DTSFF_TRACE=1, build_ab/libtsff_trace.so): every workgroup
stamps s_memrealtime (100 MHz) at its phase boundaries; this script runs the headline step once with the stamps on and prints
where a launch's time goes per CU: fill, steady state, drain; phase durations per feature; wavefront imbalance of the sweep.

usage (anywhere, on a saved trace): python scripts/trace_fused.py --analyse trace.npz
usage (GPU box): TSFF_LIBRARY=$PWD/build_ab/libtsff_trace.so python scripts/trace_fused.py [B] [out.npz] [extra plan bits]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
analyse_only = len(sys.argv) > 2 and sys.argv[1] == "--analyse"
B = int(sys.argv[1]) if len(sys.argv) > 1 and not analyse_only else 4096
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "trace_fused.npz")
plan = int(sys.argv[3]) if len(sys.argv) > 3 else 0

if analyse_only:
    z = np.load(sys.argv[2])
    t, B, kt = z['trace'], int(z['B']), z['kernel_ms']
    nwg = 2 * B
else:
    import torch

    from tsadar_amd import _lib as L
    from tsadar_amd import synthetic as S
    from tsadar_amd.calibration import sa_lookup
    from tsadar_amd.engine import Engine

    lib = L.load()
    cfg = S.baseline_deck(batch_size=B)
    sa = sa_lookup("P9")
    sa = dict(sa=sa["sa"], weights=sa["weights"] * np.ones([B, 10]))
    eng = Engine(cfg, sa, activate=True)
    if plan:
        eng.set_launch_plan(plan)
    rng = np.random.default_rng(S.SEED)
    truth = S.draw_params(cfg, B, rng)
    batch = S.make_batch(eng, truth, rng)
    guess = S.draw_params(cfg, B, rng)
    X = eng.dev(guess.to_matrix())
    gmask = guess.grad_mask()
    act = [s for _, s in guess.slots.active_leaves]
    w = eng.loss_weights(B, float(batch["i_data"].max()), float(batch["e_data"].max()), cfg["data"]["ion_loss_scale"])
    packed = torch.zeros(3 + len(act) * B, dtype=torch.float64, device=eng.device)
    for _ in range(3):
        eng.loss_grad_packed(X, batch, w, gmask, act, B, 0, out=packed)
    torch.cuda.synchronize()
    nwg = 2 * B
    tr = torch.zeros((nwg, 16), dtype=torch.int64, device=eng.device)
    fn = lib.tsff_debug_trace
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p]
    assert fn(tr.data_ptr()) == 0
    eng.enable_timing(4)
    for _ in range(2):
        eng.loss_grad_packed(X, batch, w, gmask, act, B, 0, out=packed)
    torch.cuda.synchronize()
    kt = eng.kernel_times_ms()
    assert fn(None) == 0
    t = tr.cpu().numpy().astype(np.int64)
    np.savez_compressed(out, trace=t, B=B, kernel_ms=kt)


hw = t[:, 0]
loc = ((hw >> 32) & 0xF) * 65536 + (hw & 0xFFFFFFFF & 0xFF00)   # (XCC id, SE/SH/CU bits of HW_ID): one value per CU
cus, cu_idx = np.unique(loc, return_inverse=True)
T = t[:, 1:9].astype(np.float64) * 0.01   # microseconds (100 MHz)
t0 = T[:, 0].min()
T -= t0
end = T[:, 7].max()
print(f"B {B}: kernel by HIP events {np.round(kt, 4)} ms; span of the stamps {end:.1f} us; distinct CU ids {len(cus)}; workgroups {nwg}")
names = ["prologue", "sweep", "conv1+argmax", "loss+sums", "ybar+conv2", "xbar+contract+wavesums", "chain+store"]
for f, nm in ((0, "EPW"), (1, "IAW")):
    sl = slice(f * B, (f + 1) * B)
    d = np.diff(T[sl], axis=1)
    tot = T[sl, 7] - T[sl, 0]
    print(f"  {nm}: workgroup latency mean {tot.mean():.1f} us (p5 {np.percentile(tot, 5):.1f}, p95 {np.percentile(tot, 95):.1f}); phases (mean us): "
          + ", ".join(f"{n} {v:.1f}" for n, v in zip(names, d.mean(axis=0))))
    wv = t[sl, 9:13].astype(np.float64) * 0.01 - t0 - T[sl, 1:2]   # each wavefront's sweep end, from the start of the sweep
    print(f"       sweep end per wavefront (mean us from sweep start): {np.round(wv.mean(axis=0), 1)}; barrier wait = max - mean: {np.mean(wv.max(axis=1) - wv.mean(axis=1)):.1f}")
# per CU: first start, last end, busy intervals
starts, ends = T[:, 0], T[:, 7]
per_cu_first = np.array([starts[cu_idx == c].min() for c in range(len(cus))])
per_cu_last = np.array([ends[cu_idx == c].max() for c in range(len(cus))])
per_cu_n = np.bincount(cu_idx)
print(f"  per CU: workgroups min/mean/max {per_cu_n.min()}/{per_cu_n.mean():.1f}/{per_cu_n.max()}; first start mean {per_cu_first.mean():.1f} max {per_cu_first.max():.1f} us; "
      f"last end min {per_cu_last.min():.1f} mean {per_cu_last.mean():.1f} max {per_cu_last.max():.1f} us")
# concurrency over time: number of resident workgroups, in 5 us bins
edges = np.arange(0.0, end + 5.0, 5.0)
occ = np.zeros(len(edges) - 1)
for i in range(len(edges) - 1):
    lo, hi = edges[i], edges[i + 1]
    occ[i] = np.clip(np.minimum(ends, hi) - np.maximum(starts, lo), 0, None).sum() / 5.0
print("  resident workgroups (5 us bins):", " ".join(f"{int(round(o))}" for o in occ))
# throughput over time: workgroups finished per 20 us
fin = np.histogram(ends, bins=np.arange(0.0, end + 20.0, 20.0))[0]
print("  workgroups finished per 20 us:", " ".join(str(int(x)) for x in fin))
# sweep-phase overlap: per CU, fraction of time with 0, 1, 2 workgroups in their sweep
sw0, sw1 = T[:, 1], T[:, 2]
grid_t = np.arange(0.0, end, 0.5)
tot_by_n = np.zeros(4)
for c in range(len(cus)):
    m = cu_idx == c
    n_sw = ((sw0[m][:, None] <= grid_t[None, :]) & (grid_t[None, :] < sw1[m][:, None])).sum(axis=0)
    for k in range(4):
        tot_by_n[k] += (n_sw == k).sum()
tot_by_n /= tot_by_n.sum()
print("  share of (CU, time) with k workgroups inside their sweep, k = 0..3:", np.round(tot_by_n, 3))
