"""Fixture of the 2-D angular path from the CPU oracle (no GPU involved): FormFactor.calc_in_2D on a 48 x 48
anisotropic table at a handful of wavelengths and three oblique angles, and the ARTS instrument chain (weight matrix,
2-D IRF, 8 x 4 resolution units) on a 1-D DLM form factor at the reference's calibration geometry.
    python tests/golden/make_golden_2d.py        # writes tests/golden/oracle_2d.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import decks  # noqa: E402
import util  # noqa: E402
from oracle import tsadar_oracle as orc  # noqa: E402


def fe2d(nv):
    vx = orc.velocity_grid(nv)
    X, Y = np.meshgrid(vx, vx, indexing="ij")
    f = np.exp(-((X / 1.3) ** 2 + (Y / 0.8) ** 2) ** 1.4 / 2) + 0.05 * np.exp(-((X - 2.0) ** 2 + (Y + 1.0) ** 2))
    return vx, f / (f.sum() * (vx[1] - vx[0]) ** 2)


def main():
    out = {}
    # ---- 2-D form factor ----
    cfg = decks.deck_fit()
    B, nv = 2, 48
    sa = np.array([35.0, 60.0, 110.0])
    normed = util.random_lineouts(cfg, B, seed=61, ranges=dict(ud=(-1.5, 1.5)))
    phys = orc.physical_params(cfg["parameters"], normed, True)
    phys["ud"] = np.array([0.8, -1.1])
    vx, f2 = fe2d(nv)
    idx = np.array([0, 1, 100, 333, 511, 512, 700, 1023])
    out.update(ff_X=util.normed_to_matrix(phys, 1), ff_fe2d=f2, ff_sa=sa, ff_idx=idx, ff_ud_angle=25.0, ff_va_angle=-40.0)
    for feature, rng in ((0, cfg["other"]["lamrangE"]), (1, cfg["other"]["lamrangI"])):
        P = np.stack([orc.form_factor_2d(rng, 1024, 0.0, sa, 1, orc.lineout_params(phys, b, 1), vx, f2, 25.0, -40.0, lam_index=idx)[0]
                      for b in range(B)])
        out[f"ff_P{feature}"] = P
    # ---- ARTS instrument chain on a 1-D DLM form factor ----
    from tsadar_amd import calibration

    acfg = decks.deck_angular(1, 64, (128, 256), 10, 110)
    acfg["other"]["extraoptions"]["spectype"] = "angular"
    asa = calibration.get_scattering_angles(acfg)
    acfg["other"]["extraoptions"]["spectype"] = "angular_full"
    ang_axis = calibration.angular_pixel_axis()
    an = orc.init_normed_params(acfg["parameters"], 1, True)
    ap = orc.lineout_params(orc.physical_params(acfg["parameters"], an, True), 0, 1)
    Po, lam_cm = orc.form_factor(acfg["other"]["lamrangE"], 1024, 0.0, asa["sa"], 1, ap, orc.velocity_grid(64), orc.dlm_fe(float(ap["m"]), 64))
    e_amps = np.random.default_rng(3).uniform(0.5, 2.0, (100, 1))
    E, lam_o = orc.ats_spectrum(acfg, asa["weights"], ang_axis, Po, np.squeeze(lam_cm) * 1e7, 256, e_amps, ap)
    out.update(ats_E=E, ats_lam=lam_o, ats_e_amps=e_amps)
    np.savez_compressed(os.path.join(HERE, "oracle_2d.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
