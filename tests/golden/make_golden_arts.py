"""Angular (ARTS) regression fixtures from the reference's OWN test decks, run through the oracle:

    tests/configs/arts1v_test_defaults.yaml + arts1v_test_inputs.yaml      (tests/test_forward/test_angular_1v.py)
    tests/configs/arts2v_test_defaults.yaml + arts2d_test_inputs.yaml      (tests/test_forward/test_angular_2v.py; the
        "arts2v_test_inputs.yaml" it opens does not exist in the reference tree, arts2d_test_inputs.yaml is the deck of that name)

merged exactly as the tests merge them (flatten -> update -> unflatten) and with the derived keys the tests add.  The
reference's golden arrays for these two tests (ThryE-arts1v.npy, ThryE-arts2v.npy, shape [860, 1024]) are NOT in the reference
tree (.MISSING_LARGE_BLOBS), so parity of the angular path stays UNPINNED; these fixtures hold what the oracle's restatement
gives for the same decks, so that the day the blobs are available the comparison is one line:

    np.testing.assert_allclose(np.load("ThryE-arts1v.npy")[z["rows"]], z["ThryE"], rtol=1e-4)

Written: tests/golden/arts1v_deck.json, arts2v_deck.json (the merged decks: input data), tests/golden/oracle_arts.npz (every
10th row of the two [860, 1024] images, the wavelength axis).   python tests/golden/make_golden_arts.py   (CPU, ~10 min on 8 cores)
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import tsadar_oracle as orc  # noqa: E402

REF = "/root/reference/tests/configs"
ROWS = np.arange(0, 860, 10)


def flatten(d, pre=()):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten(v, pre + (k,)))
        else:
            out[pre + (k,)] = v
    return out


def unflatten(f):
    out = {}
    for ks, v in f.items():
        d = out
        for k in ks[:-1]:
            d = d.setdefault(k, {})
        d[ks[-1]] = v
    return out


def merged_deck(defaults, inputs):
    """The merge and the derived keys of tests/test_forward/test_angular_1v.py:33-51."""
    import yaml

    d = flatten(yaml.safe_load(open(os.path.join(REF, defaults))))
    d.update(flatten(yaml.safe_load(open(os.path.join(REF, inputs)))))
    cfg = unflatten(d)
    r = cfg["data"]["fit_rng"]
    cfg["other"]["lamrangE"] = [r["forward_epw_start"], r["forward_epw_end"]]
    cfg["other"]["lamrangI"] = [r["forward_iaw_start"], r["forward_iaw_end"]]
    cfg["other"]["npts"] = int(cfg["other"]["CCDsize"][1] * cfg["other"]["points_per_pixel"])
    return cfg


def angular_sa(cfg):
    from tsadar_amd import calibration

    cfg["other"]["extraoptions"]["spectype"] = "angular"
    sa = calibration.get_scattering_angles(cfg)
    cfg["other"]["extraoptions"]["spectype"] = "angular_full"   # test_angular_1v.py:58
    sa["angAxis"] = calibration.angular_pixel_axis()
    return sa


def _ff2d_chunk(args):
    cfg, sa_deg, p, vx, fe2, ud, va, idx = args
    return orc.form_factor_2d(cfg["other"]["lamrangE"], cfg["other"]["npts"], 0.0, sa_deg, 1, p, vx, fe2, ud, va, lam_index=idx)[0]


def oracle_image(cfg, sa):
    npts = cfg["other"]["npts"]
    phys = orc.physical_params(cfg["parameters"], orc.init_normed_params(cfg["parameters"], 1, True), True)
    p = orc.lineout_params(phys, 0, 1)
    fecfg = cfg["parameters"]["electron"]["fe"]
    gen = cfg["parameters"]["general"]
    if int(fecfg.get("dim", 1)) == 1:
        nvx = fecfg["nvx"]
        Po, lam_cm = orc.form_factor(cfg["other"]["lamrangE"], npts, 0.0, sa["sa"], 1, p, orc.velocity_grid(nvx), orc.dlm_fe(float(p["m"]), nvx))
        lam_nm = np.squeeze(lam_cm) * 1e7
    else:
        fe2 = orc.spherical_harmonics_fe(fecfg)
        vx = orc.velocity_grid(fecfg["nvx"])
        chunks = np.array_split(np.arange(npts), 64)
        with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
            parts = pool.map(_ff2d_chunk, [(cfg, sa["sa"], p, vx, fe2, gen["ud"]["angle"], gen["Va"]["angle"], c) for c in chunks])
        Po = np.concatenate(parts, axis=1)
        lam_nm = np.linspace(cfg["other"]["lamrangE"][0], cfg["other"]["lamrangE"][1], npts)
    n_out = cfg["other"]["CCDsize"][1]   # dummy_batch e_data is [CCDsize[0], CCDsize[1]]  (test_angular_1v.py:62-65)
    rows = cfg["data"]["lineouts"]["end"] - cfg["data"]["lineouts"]["start"]
    return orc.ats_spectrum(cfg, sa["weights"], sa["angAxis"], Po, lam_nm, n_out, np.ones((rows, 1)), p)


def main():
    out = {"rows": ROWS}
    for tag, dfl, inp in (("arts1v", "arts1v_test_defaults.yaml", "arts1v_test_inputs.yaml"),
                          ("arts2v", "arts2v_test_defaults.yaml", "arts2d_test_inputs.yaml")):
        cfg = merged_deck(dfl, inp)
        json.dump(cfg, open(os.path.join(HERE, f"{tag}_deck.json"), "w"), indent=1, sort_keys=True)
        sa = angular_sa(cfg)
        E, lam = oracle_image(cfg, sa)
        assert E.shape == (860, 1024) and np.all(np.isfinite(E)), E.shape
        out[f"ThryE_{tag}"] = E[ROWS]
        out[f"lam_{tag}"] = lam
        print(tag, E.shape, float(E.max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "oracle_arts.npz"), **out)


if __name__ == "__main__":
    main()
