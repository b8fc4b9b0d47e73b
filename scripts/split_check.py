"""k_spectrum_rows: the split (one workgroup per round) and the unsplit form at batch sizes around the switch, repeated calls -- bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import decks, util
from oracle import tsadar_oracle as orc
from tsadar_amd.engine import Engine
bad = 0
for ppp in (2, 5):
    cfg = decks.deck_fit(points_per_pixel=ppp)
    for B in (1, 7, 63, 64, 65, 130):
        sa = util.sa_fit(B)
        batch = util.synthetic_batch(cfg, sa, B, seed=900 + B)
        normed = util.random_lineouts(cfg, B, seed=950 + B)
        i_norm, e_norm = orc.loss_norms(cfg, batch)
        eng = Engine(cfg, sa)
        w = eng.loss_weights(B, i_norm, e_norm)
        X = util.normed_to_matrix(normed, 1); gm = eng.slots.active.astype(np.uint8)
        ref = None
        for plan in (0, 1, 0, 0, 1, 0):
            eng.set_launch_plan(plan)
            out = [a.cpu().numpy() for a in eng.loss_grad(X, batch, w, gm, want_spectra=True)]
            if ref is None: ref = out
            same = all(np.array_equal(a, b) for a, b in zip(ref, out))
            if not same: bad += 1; print("MISMATCH ppp", ppp, "B", B, "plan", plan)
        print("ppp", ppp, "B", B, "ok" if bad == 0 else "bad so far %d" % bad, flush=True)
print("split_check:", "all identical" if bad == 0 else "%d mismatches" % bad)
sys.exit(1 if bad else 0)
