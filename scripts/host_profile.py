"""Where does a fit step spend host time?  cProfile of LossFunction.vg_loss (B = 4096) on the GPU box."""
import cProfile, pstats, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsadar_amd import synthetic as S, tree, ThomsonParams
from tsadar_amd.engine import Engine
from tsadar_amd.loss_function import LossFunction
from tsadar_amd.calibration import sa_lookup

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = S.baseline_deck(batch_size=B)
sa = sa_lookup("P9"); sa = dict(sa=sa["sa"], weights=sa["weights"] * np.ones([B, 10]))
eng = Engine(cfg, sa)
rng = np.random.default_rng(1)
truth = S.draw_params(cfg, B, rng); batch = S.make_batch(eng, truth, rng)
hb = {k: (v.cpu().numpy() if v is not None else None) for k, v in batch.items()}
hb["noise_e"] = np.zeros((B, 1024)); hb["noise_i"] = np.zeros((B, 1024))
lf = LossFunction(cfg, sa, hb)
tp = S.draw_params(cfg, B, rng)
diff, static = tree.partition(tp, tree.get_filter_spec(cfg["parameters"], tp))
x0, lf.unravel_weights = tree.ravel_pytree(diff)
for _ in range(3): lf.vg_loss(x0, static, hb)
t = time.perf_counter()
n = 50
for _ in range(n): lf.vg_loss(x0, static, hb)
dt = (time.perf_counter() - t) / n
print("vg_loss ms per call", dt * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(n): lf.vg_loss(x0, static, hb)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
