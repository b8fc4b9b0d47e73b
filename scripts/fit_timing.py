"""End-to-end cost of the fit step and of a whole fit at B = 4096: LossFunction.vg_loss ms per call with a
kernel / PCIe+host split, and one scipy L-BFGS-B fit exactly as the reference drives it (inverse/loops.py:43-51, maxiter =
num_epochs = 120).  Writes profiles/<tag>_fit_timing.json (and gpurun_out/).   usage: python scripts/fit_timing.py <tag> [B] [points_per_pixel]"""
import cProfile, io, json, os, pstats, sys, time
import numpy as np, torch, scipy.optimize as spopt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsadar_amd import synthetic as S, tree
from tsadar_amd.loss_function import LossFunction
from tsadar_amd.calibration import sa_lookup

tag = sys.argv[1] if len(sys.argv) > 1 else "fit"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
PPP = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = S.baseline_deck(points_per_pixel=PPP, batch_size=B)
cfg["optimizer"]["method"] = "l-bfgs-b"
sa = sa_lookup("P9"); sa = dict(sa=sa["sa"], weights=sa["weights"] * np.ones([B, 10]))
rng = np.random.default_rng(S.SEED)
truth = S.draw_params(cfg, B, rng)
from tsadar_amd.engine import Engine
eng0 = Engine(cfg, sa)
batch = S.make_batch(eng0, truth, rng)
hb = {k: (v.cpu().numpy() if v is not None else None) for k, v in batch.items()}
hb["noise_e"] = np.zeros((B, 1024)); hb["noise_i"] = np.zeros((B, 1024))
del eng0
lf = LossFunction(cfg, sa, hb)
tp = S.draw_params(cfg, B, rng)
diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
x0, lf.unravel_weights = tree.ravel_pytree(diff)
eng = lf.ts_diag.engine(tp.activate)
for _ in range(5): lf.vg_loss(x0, static, hb)
n = 100
eng.enable_timing(n)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n): v, g = lf.vg_loss(x0, static, hb)
dt_call = (time.perf_counter() - t) / n * 1e3
kt = eng.kernel_times_ms()
res = {"B": B, "points_per_pixel": PPP, "free_params_per_lineout": len(diff.values), "unknowns": int(x0.size),
       "vg_loss_ms_per_call": dt_call, "kernel_ms_avg": float(np.mean(kt)), "kernel_ms_median": float(np.median(kt)),
       "host_pcie_ms_per_call": dt_call - float(np.mean(kt)),
       "host_pcie_over_kernel": (dt_call - float(np.mean(kt))) / float(np.mean(kt))}
pr = cProfile.Profile(); pr.enable()
for _ in range(50): lf.vg_loss(x0, static, hb)
pr.disable()
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(14)
res["cprofile_top_tottime"] = [l.strip() for l in sio.getvalue().splitlines() if l.strip()][4:22]
# one whole fit, as loops._1d_scipy_loop_ runs it
eng.enable_timing(4096)
calls = [0]
def fun(x, *a):
    calls[0] += 1
    return lf.vg_loss(x, *a)
torch.cuda.synchronize()
t = time.perf_counter()
out = spopt.minimize(fun, x0, args=(static, hb), method="l-bfgs-b", jac=True, bounds=None, options={"disp": False, "maxiter": 120})
torch.cuda.synchronize()
dt_fit = time.perf_counter() - t
kt = eng.kernel_times_ms()
res["lbfgs_fit"] = {"wall_s": dt_fit, "iterations": int(out.nit), "function_evaluations": int(out.nfev), "vg_loss_calls": calls[0],
                    "final_loss": float(out.fun), "initial_loss": float(v),
                    "kernel_s": float(np.sum(kt)) * 1e-3 * (calls[0] / max(len(kt), 1)),
                    "vg_loss_s (calls x ms per call above)": calls[0] * dt_call * 1e-3,
                    "scipy_lbfgsb_host_s": dt_fit - calls[0] * dt_call * 1e-3,
                    "spectra_per_s_incl_optimiser": B * calls[0] / dt_fit}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for d in ("profiles", "gpurun_out"):
    json.dump(res, open(os.path.join(ROOT, d, f"{tag}_fit_timing.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
