"""Generates tests/golden/oracle_fit_b4.npz: inputs and expected outputs of the hot path for four
synthetic lineouts of the BASELINE fit deck (EPW + IAW, Maxwellian f_e, 1024 wavelength points, P9
angles), computed by the CPU oracle (NumPy forward, torch-f64 autodiff gradient).  The oracle itself is
pinned to the reference's golden vector (tests/test_oracle_golden.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import decks  # noqa: E402
import util  # noqa: E402
from oracle import tsadar_oracle as orc  # noqa: E402
from oracle import tsadar_oracle_torch as ot  # noqa: E402

NAMES = ["Te", "ne", "Ti_1", "Va", "lam", "amp1"]


def main():
    B = 4
    cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "lam", "amp1"))
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=20251004)
    normed = util.random_lineouts(cfg, B, seed=20251004)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    loss, E, I = orc.loss(cfg, sa, normed, batch, i_norm, e_norm)
    val, g, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, NAMES)
    assert abs(val - loss) < 1e-12 * abs(loss)
    aloss, sq, _, _ = orc.array_loss(cfg, sa, normed, batch)
    out = dict(X=util.normed_to_matrix(normed, 1), loss=loss, i_norm=i_norm, e_norm=e_norm, ThryE=E, ThryI=I,
               grad=np.stack([g[k] for k in NAMES], axis=1), array_loss=aloss, sqdev_ele=sq["ele"], sqdev_ion=sq["ion"])
    out.update({k: np.asarray(v) for k, v in batch.items()})
    np.savez_compressed(os.path.join(HERE, "oracle_fit_b4.npz"), **out)
    print("wrote oracle_fit_b4.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
