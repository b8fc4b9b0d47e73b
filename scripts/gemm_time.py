"""Times k_fe_vectors + k_wgemm through tsff_chi_table-like path: the DLM bench, kernel stats only."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from tsadar_amd.engine import Engine
from tsadar_amd import synthetic as S

cfg = S.baseline_deck(1, 128, ("Te", "ne", "m", "amp1", "amp2", "lam"), 4096)
sa = dict(sa=np.linspace(53.637560, 66.1191, 10), weights=np.ones((1, 10)) / 10)
eng = Engine(cfg, sa)
print("fma peak", eng.fp64_fma_peak_tflops(), "mfma peak", eng.fp64_mfma_peak_tflops())
