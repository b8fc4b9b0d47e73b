#!/usr/bin/env python
"""Benchmark of the hot path: spectra/s for forward + gradient of one EPW(1024 lambda) + IAW(1024 lambda)
spectrum, batch 4096 per GPU, P = 6 free parameters (BASELINE.json metric, configs[2] / configs[4]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one evaluation of loss + gradient (tsff_loss_grad_packed) over this rank's 4096 lineouts, with the
inputs (params, data, amplitudes) already resident in HBM; the gradient kernels write [3 loss sums | gradient in the
optimiser's ravel order over the global batch] straight into the buffer that -- for N > 1 -- the step's single RCCL
all-reduce sums in place (weak scaling: 4096 lineouts per GPU).  Rank 0 prints ONE JSON line.  ``value`` is that
HBM-resident rate; ``value_pcie_inclusive`` is the step SURVEY.md 8(d) describes (params host -> device and loss +
gradient device -> host through pinned staging every step, as LossFunction.vg_loss does).  The oracle is used only for
the ``cpu_baseline`` leg (rank 0, N = 1).

Other lines (same schema, not the headline metric): ``--forward-only --batch 256`` (BASELINE configs[1]), ``--dlm``
(per-lineout super-Gaussian order: W tables by the FP64 MFMA GEMM), ``--free-form``, ``--config4`` (BASELINE configs[3]:
2-D angular form factor, 256 x 256 f_e, 512 angles x 1024 wavelengths).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
FP64_PEAK = 78.6e12    # flop/s, AMD public spec for FP64 vector (not in the local guide)
FLOP_PER_SPECTRUM = 13.0e6  # forward + adjoint, shared W table
FLOP_PER_SPECTRUM_FWD = 4.3e6  # forward only (4 MFLOP of points + 0.3 MFLOP of instrument function)
N_SIMD = 1024               # 256 CUs x 4 SIMDs: a VALU instruction occupies its SIMD's issue port for 4 cycles


def self_launch(args) -> int:
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks as FRESH processes
    (``python -m torch.distributed.run`` on 127.0.0.1, one rank per GPU) before this process has imported torch or touched
    the GPU, pass their stdout's one JSON line through and exit with their code.  (The driver's own
    ``python -m torch.distributed.run ... bench.py --gpus N`` sets WORLD_SIZE and never comes through here.)"""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in p.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)   # (anything a rank wrote to stdout besides the line)
    if p.returncode != 0:
        print(f"bench.py: the {args.gpus}-rank launch failed with exit code {p.returncode}", file=sys.stderr)
        return p.returncode
    if len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0])
    return 0


def algorithmic_bytes(NP: int, with_noise: bool) -> int:
    """SURVEY.md section 8d, per spectrum (EPW + IAW), forward + gradient."""
    rd = 2 * 1024 * 8 + 2 * 8 + NP * 8 + (2 * 1024 * 8 if with_noise else 0)
    wr = NP * 8 + 3 * 8
    return rd + wr


def cpu_baseline(cfg, B, n_sample):
    """The oracle on the host cores: oracle/c/tsadar_oracle.cpp (C++/OpenMP restatement of the reference algorithm,
    loss + gradient by forward-mode dual numbers, pinned to the reference's golden vector in tests/test_oracle_c.py) on
    the first ``n_sample`` lineouts of the same seeded draw.  Runs before this process touches the GPU.  Returns the
    ``cpu_baseline`` object."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import c_oracle as co
    from tsadar_amd import synthetic as S
    from tsadar_amd.calibration import sa_lookup

    n = min(n_sample, B)
    rng = np.random.default_rng(S.SEED)
    truth = S.draw_params(cfg, B, rng)
    rng.integers(1 << 31)  # the draw make_batch() consumes on the GPU leg
    guess = S.draw_params(cfg, B, rng)
    p9 = sa_lookup("P9")
    sa = dict(sa=p9["sa"], weights=p9["weights"] * np.ones([n, 10]))
    cores = max(1, min(os.cpu_count() or 1, co.max_threads(), 16))
    ones = dict(e_amps=np.ones(n), i_amps=np.ones(n))
    # (untimed) synthetic 'measured' data: oracle forward at the truth parameters + 1 % noise, amplitudes = row maxima
    _, _, E, I = co.loss_grad(cfg, sa, truth.X[:n], ones, nthreads=cores)
    nrng = np.random.default_rng(S.SEED + 1)
    E = E * (1 + 0.01 * nrng.standard_normal(E.shape))
    I = I * (1 + 0.01 * nrng.standard_normal(I.shape))
    batch = dict(e_data=E, i_data=I, e_amps=E.max(axis=1), i_amps=I.max(axis=1))
    w = np.array([1.0 / n, 0.5 / n, 0.5 / n])
    gm = guess.grad_mask()
    co.loss_grad(cfg, sa, guess.X[:cores], {k: v[:cores] for k, v in batch.items()}, w=w, gmask=gm, nthreads=cores, want_spectra=False)
    t0 = time.perf_counter()
    co.loss_grad(cfg, sa, guess.X[:n], batch, w=w, gmask=gm, nthreads=cores, want_spectra=False)
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "spectra/s", "cores": cores, "kind": "port",
        "sample": (f"first {n} of the {B} lineouts (same seeded parameter draws), loss+grad of one EPW+IAW spectrum each by the "
                   f"C++/OpenMP oracle (restatement of the reference algorithm, forward-mode dual numbers for the 6 free "
                   f"parameters, W table built once), {cores} threads, {dt:.1f} s wall = {dt * cores:.0f} core-seconds"),
    }


def config4_main(args):
    """BASELINE configs[3]: non-Maxwellian f_e on a 256 x 256 velocity grid, 512 scattering angles x 1024 wavelengths, one
    MI355X.  A step = one 2-D form-factor image (tsff_form_factor_2d: rotate the table by the angle of xi_e, project,
    differentiate, interpolate, ratintn -- per (lambda, theta) point; FormFactor.calc_in_2D, form_factor.py:449-587) with the
    table and parameters resident in HBM.  Parity of this path is UNPINNED against the reference (its ARTS goldens are
    not in the reference tree): it is pinned to the oracle's restatement (tests/test_gpu_parity.py::test_config4_*)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tsadar_amd import synthetic as S
    from tsadar_amd import ThomsonParams

    arts = bool(getattr(args, "arts", False))
    nv, na, npts = (128, 241, 1024) if arts else (256, 512, 1024)
    cfg = S.baseline_deck()
    sa = dict(sa=np.linspace(19.0, 139.0, na), weights=np.ones((1, na)) / na)
    vx = np.linspace(-6 + 6.0 / nv, 6 - 6.0 / nv, nv)
    Xg, Yg = np.meshgrid(vx, vx, indexing="ij")
    f = np.exp(-((Xg / 1.3) ** 2 + (Yg / 0.8) ** 2) ** 1.4 / 2) + 0.05 * np.exp(-((Xg - 2.0) ** 2 + (Yg + 1.0) ** 2))
    f /= f.sum() * (vx[1] - vx[0]) ** 2
    ud_ang, va_ang = 25.0, -40.0
    tp = ThomsonParams(cfg["parameters"], 1, batch=True, activate=False)
    phys = tp.physical_matrix()
    phys[0, 9] = 0.7   # a drift, so that the rotation angle varies over the image

    cpu_res = None
    if args.cpu_sample > 0:   # the oracle (NumPy restatement, one core) on a bounded sample: one wavelength x all 512 angles
        from oracle import tsadar_oracle as orc

        names = {"Te": 0, "ne": 1, "lam": 3, "amp1": 4, "amp2": 5, "amp3": 6, "ne_gradient": 7, "Te_gradient": 8, "ud": 9, "Va": 10}
        p = {k: float(phys[0, s]) for k, s in names.items()}
        p.update(Ti=[float(phys[0, 11])], Z=[float(phys[0, 12])], A=[float(phys[0, 13])], fract=[float(phys[0, 14])])
        t0 = time.perf_counter()
        orc.form_factor_2d(cfg["other"]["lamrangE"], npts, 0.0, sa["sa"], 1, p, vx, f, ud_ang, va_ang, lam_index=np.array([300]))
        dt = time.perf_counter() - t0
        cpu_res = {"value": na / dt / (npts * na), "unit": "images/s", "cores": 1, "kind": "port",
                   "sample": f"{na} of the {npts * na} points of one image (wavelength index 300, all {na} angles) by the NumPy oracle "
                             f"(restatement of calc_in_2D / rotate / calc_chi_vals), 1 core, {dt:.1f} s; scaled to whole images"}

    import torch

    from tsadar_amd.engine import Engine

    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    torch.cuda.set_device(0)
    eng = Engine(cfg, sa, activate=False)
    fd, Xd = eng.dev(f), eng.dev(phys)
    P = torch.empty((1, 1, npts, na), dtype=torch.float64, device=eng.device)

    def step():
        return eng.form_factor_2d(0, Xd, fd, ud_ang, va_ang, out=P)

    steps, warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    eng.enable_timing(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = eng.kernel_times_ms()
    kavg_s = float(np.mean(kt)) * 1e-3
    l1_peak = eng.l1_read_peak_tbps()
    lds_peak = 256.0 * 256 * 2.4e9 / 1e12           # ds_read_b64: 256 B per clock and CU (MI355X_MICROARCH.md, LDS), 256 CUs, 2.4 GHz
    samples = float(npts) * na * nv * nv            # bicubic samples per image: every point rotates the whole table
    stencil_bytes = samples * 16 * 8                # 4 x 4 doubles per sample, read through L1/L2 (the table is 532 KB)
    flop = samples * 60.0                           # two Catmull-Rom weight sets (~24) + 16-term contraction (~36)
    req_bytes = stencil_bytes if arts else stencil_bytes / 2   # what the kernel loads per sample: all 16 entries (LDS) / the entering row + column
    res = {
        "metric": "2-D angular form-factor images/s (forward), %dx%d f_e, %d angles x 1024 lambda" % (nv, nv, na),
        "value": steps / dt, "unit": "images/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * dt / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("the reference's ARTS size: non-Maxwellian f_e on a 128x128 v-grid (LDS-resident), 241 scattering angles x 1024 lambda, "
                                "one plasma condition (parity unpinned against the reference; pinned to the oracle)") if arts else
                               ("configs[3]: arts-2d angular, non-Maxwellian f_e on a 256x256 v-grid, 512 scattering angles x 1024 lambda, "
                                "one plasma condition (parity unpinned against the reference; pinned to the oracle)"),
                   "nv": nv, "n_angles": na, "n_lambda": npts, "points": npts * na, "bicubic_samples": samples},
    }
    valu_per_sample = 75.0 if arts else 93.0   # static count of the sampling loop (scripts/isa_mix.py on the kernel: profiles/r02d / r03 notes in DESIGN.md 4.3)
    issue = samples / 64.0 * valu_per_sample * 4.0 / (1024 * 2.4e9 * kavg_s)
    fp64 = {"bound": "fp64-valu", "achieved": flop / kavg_s / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": flop / kavg_s / FP64_PEAK,
            "flop_per_sample": 60.0, "issue_slot_frac": issue, "valu_instructions_per_sample": valu_per_sample,
            "kernel_avg_ms": kavg_s * 1e3, "kernel_median_ms": float(np.median(kt)), "traffic": None,
            "note": "frac: 60 flop per bicubic sample (two Catmull-Rom weight sets + the 16-term contraction) / kernel time / 78.6 TF; "
                    "issue_slot_frac: the sampling loop's VALU instructions per sample (static count) x samples / 64 lanes x 4 cycles / (1024 SIMDs x "
                    "2.4 GHz x kernel time) -- what the kernel is bound by; HBM traffic is the table + 4 MB of P"}
    if arts:
        res["roofline"] = {"bound": "lds", "achieved": req_bytes / kavg_s / 1e12, "peak": lds_peak, "unit": "TB/s", "frac": req_bytes / kavg_s / 1e12 / lds_peak,
                           "traffic": None, "kernel": "k_form_factor_2d<1,true,4,false> (table in LDS, one ds_read_b64 per stencil entry)",
                           "kernel_avg_ms": kavg_s * 1e3, "kernel_median_ms": float(np.median(kt)), "algorithmic_bytes_per_launch": req_bytes,
                           "note": "achieved = 128 B of stencil read from LDS per bicubic sample x samples / kernel time; peak = 256 B per clock and CU of "
                                   "ds_read_b64 x 256 CUs x 2.4 GHz (conflict-free; the sampler's rotated lines measure 2.0 passes per read); the kernel is "
                                   "VALU-issue bound first (roofline_fp64)"}
        res["roofline_fp64"] = fp64
    else:
        # configs[3] (table through L1/L2): FP64-VALU issue is the stated bound (with no requests at all the loop takes 90.6 ms: the floor);
        # the L1 view counts what the rolling window may REQUEST -- the entering row and column, 64 B per sample, each only in the lanes
        # whose cell moves on that axis -- never the 128 B of stencil a sample consumes
        fp64["kernel"] = ("k_form_factor_2d<1,false,1,false> (table read through L1/L2 from the padded copy of k_pad2d and its transpose; rolling 4x4 "
                          "window, requests one sample ahead)")
        res["roofline"] = fp64
        res["roofline_l1"] = {"bound": "l1", "achieved_upper": req_bytes / kavg_s / 1e12, "peak": l1_peak, "unit": "TB/s",
                              "frac_upper": req_bytes / kavg_s / 1e12 / l1_peak,
                              "note": "upper bound of the requested bytes (64 B per sample if every lane's cell moved on both axes every sample; the "
                                      "requests are masked to the lanes that shift: ~0.9 of the lanes on the major axis, ~0.37 on the minor one) / "
                                      "kernel time against the vector-L1 read rate tsff_l1_read_peak measures on this device"}
    if cpu_res is not None:
        res["cpu_baseline"] = cpu_res
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="lineouts per GPU")
    ap.add_argument("--ppp", type=int, default=1, help="points per pixel (1 -> 1024 wavelength points per feature)")
    ap.add_argument("--nvx", type=int, default=128, help="velocity grid of f_e (BASELINE: 128; the reference's production decks: 320)")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="lineouts of the CPU baseline (0 = skip)")
    ap.add_argument("--forward-only", action="store_true", help="configs[1]: forward-only (not the headline metric)")
    ap.add_argument("--dlm", action="store_true",
                    help="variant: the reference's canonical active set {Te, ne, m, amp1, amp2, lam} with a per-lineout "
                         "super-Gaussian order m ~ U(2, 3.5) (per-lineout W tables; not the headline metric)")
    ap.add_argument("--free-form", action="store_true",
                    help="variant: explicit per-lineout f_e tables with the gradient w.r.t. f_e itself (Arbitrary1V-style "
                         "free-form distribution, nvx more unknowns per lineout; not the headline metric)")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE configs[3]: one 2-D angular form-factor image per step (256 x 256 f_e, 512 angles x 1024 lambda; "
                         "parity-unpinned path, not the headline metric)")
    ap.add_argument("--arts", action="store_true",
                    help="the 2-D angular form factor at the reference's ARTS size (128 x 128 f_e in LDS, 241 angles x 1024 lambda); "
                         "like --config4 a parity-unpinned path, not the headline metric")
    ap.add_argument("--irf-cutoff", type=float, default=12.0,
                    help="IRF taps kept within this many standard deviations (engine default 12: the reference's full-length convolution to the "
                         "last bit of a spectrum's 1e-22 tails; 8: differences below 1e-14 of the spectrum's maximum; not the headline setting)")
    ap.add_argument("--spin-ms", type=float, default=300.0, help="untimed clock warm-up in front of the timed region: the step repeated for this many ms (0: none, the protocol of rounds 1 and 2)")
    ap.add_argument("--dlm-blocks", type=int, default=-1, help="--dlm: column blocks of the pipelined step (TSFF_OPT_DLM_BLOCKS; 1 = one stream)")
    ap.add_argument("--plan", type=int, default=0, help="launch plan bit mask (experiments): 0 automatic, 1 never interleave the features, 2 two-sweep kernel instead of the one-sweep one")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.config4 or args.arts:
        return config4_main(args)
    variant = args.dlm or args.free_form

    from tsadar_amd import synthetic as S

    cpu_res = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_sample > 0 and not args.forward_only and not variant:
        cpu_res = cpu_baseline(S.baseline_deck(points_per_pixel=args.ppp, nvx=args.nvx, batch_size=args.batch), args.batch, args.cpu_sample)

    import torch

    from tsadar_amd import distributed as D
    from tsadar_amd.engine import Engine

    # RCCL prints a five-line banner (version, hostname, library path) on the process's STDOUT when its first communicator
    # is created: keep stdout for the one JSON line by pointing fd 1 at stderr until the communicator exists
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        world, rank, local = D.init_from_env()
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        if world > 1:
            import torch.distributed as dist

            warm = torch.zeros(1, dtype=torch.float64, device=dev)
            dist.all_reduce(warm)   # (creates the communicator)
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)

    B = args.batch
    active = ("Te", "ne", "m", "amp1", "amp2", "lam") if args.dlm else S.ACTIVE
    cfg = S.baseline_deck(points_per_pixel=args.ppp, nvx=args.nvx, batch_size=B, active=active)
    from tsadar_amd.calibration import sa_lookup

    sa = sa_lookup("P9")
    sa = dict(sa=sa["sa"], weights=sa["weights"] * np.ones([B, 10]))  # lineouts.py:103
    from tsadar_amd import _lib as L

    eng = Engine(cfg, sa, activate=True, fe_mode=L.FE_PER_LINEOUT if args.free_form else None, irf_cutoff_sigmas=args.irf_cutoff)

    if args.plan:
        eng.set_launch_plan(args.plan)
    if args.dlm_blocks >= 0:
        eng.set_dlm_blocks(args.dlm_blocks)
    # synthetic inputs: each rank draws its own shard (seed offset by rank), data generated on the GPU
    rng = np.random.default_rng(S.SEED + rank)
    truth = S.draw_params(cfg, B, rng, dlm=args.dlm)
    batch = None if args.free_form else S.make_batch(eng, truth, rng)
    guess = S.draw_params(cfg, B, rng, dlm=args.dlm)
    X = eng.dev(guess.to_matrix())
    fe_dev = None
    if args.free_form:  # super-Gaussians of random order as the current iterate of a free-form fit
        from tsadar_amd import distribution as Dist

        fe_dev = eng.dev(np.stack([Dist.dlm(m, eng.nvx) for m in rng.uniform(2.0, 3.5, B)]))
        batch = S.make_batch(eng, truth, rng, fe=fe_dev)
    gmask = guess.grad_mask()
    act = torch.tensor([s for _, s in guess.slots.active_leaves], device=dev)
    P = int(act.numel())
    e_norm = float(batch["e_data"].max())
    i_norm = float(batch["i_data"].max())
    w = eng.loss_weights(B * world, i_norm, e_norm, cfg["data"]["ion_loss_scale"])
    terms = torch.empty(3, dtype=torch.float64, device=dev)
    grad = torch.empty((B, eng.NP), dtype=torch.float64, device=dev)
    act_slots = [s for _, s in guess.slots.active_leaves]
    # the buffer of the step's one collective and one D2H copy: [3 | P x B_global], written by the gradient kernels
    packed = torch.zeros(3 + P * B * world, dtype=torch.float64, device=dev)
    packed_ff = torch.zeros(3 + (P + eng.nvx) * B * world, dtype=torch.float64, device=dev) if args.free_form else None

    # (N > 1) one event pair per step around the all-reduce, on the stream the kernels run on: the first fires when this rank's
    # kernels are done, the second when the reduced buffer is back -- collective + waiting for the slowest rank
    ar_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if world > 1 else []
    ar_i = [-10**9]   # (set to 0 right before the timed region: the event pairs belong to its steps only)

    def step():
        if args.forward_only:
            return eng.forward(X, batch["e_amps"], batch["i_amps"])
        if args.free_form:  # (extra rows d loss / d f_e from a second kernel chain: packed by tsff_pack_fe_rows)
            gfe = eng.loss_grad(X, batch, w, gmask, fe=fe_dev, out=(terms, grad), want_fe_grad=True)[4]
            eng.pack_fe_rows(terms, grad, gfe, act_slots, B * world, rank * B, out=packed_ff)
            if world > 1:
                dist.all_reduce(packed_ff)
            return packed_ff
        eng.loss_grad_packed(X, batch, w, gmask, act_slots, B * world, rank * B, out=packed)
        if world > 1:
            i = ar_i[0]
            ar_i[0] += 1
            if 0 <= i < len(ar_ev):
                ar_ev[i][0].record()
                dist.all_reduce(packed)
                ar_ev[i][1].record()
            else:
                dist.all_reduce(packed)
        return packed

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Clock warm-up (untimed).  The device clock ramps up under sustained load: in ONE burst of back-to-back launches the kernel time of
    # this very step falls from 1.00-1.08 ms to 0.85 ms over the first ~100 launches and stays there (profiles/r03v_ktrace.txt,
    # r03r_kernel_stats.csv), and it is back at 1.0 ms after a pause of 13 ms.  A fit runs thousands of steps back to back, so the
    # steady clock is the regime to report: the first K steps after the W warm-up steps are timed as before (`value_first_burst`: the
    # protocol of rounds 1 and 2), then the same step is repeated for --spin-ms without timing, and EXACTLY K steps are timed behind
    # it between the same fences (`value`).
    first_burst = None
    if args.spin_ms > 0:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dtc = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dtc], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtc = float(t.item())   # (the same number of spin steps on every rank: they hold a collective)
        first_burst = world * B * args.steps / dtc
        for _ in range(int(np.ceil(args.spin_ms * 1e-3 / (dtc / args.steps)))):
            step()
        fence()
    eng.enable_timing(max(args.steps, 1))
    ar_i[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    ktimes = eng.kernel_times_ms()
    if os.environ.get("TSFF_BENCH_KTRACE"):   # kernel time against the launch index of the timed burst (the device clock ramps under load)
        print("kernel_ms_by_launch " + " ".join("%.4f" % v for v in ktimes), file=sys.stderr)
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ar_ms = [a.elapsed_time(b) for a, b in ar_ev[:max(0, min(ar_i[0], len(ar_ev)))]]
        mine = torch.tensor([float(np.mean(ktimes)) if ktimes.size else float("nan"), float(np.mean(ar_ms)) if ar_ms else float("nan"),
                             float(np.median(ar_ms)) if ar_ms else float("nan")], dtype=torch.float64, device=dev)
        allr = torch.empty(3 * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 3).cpu().numpy()
        per_rank = {"kernel_avg_ms": [float(v) for v in allr[:, 0]], "allreduce_avg_ms": [float(v) for v in allr[:, 1]],
                    "allreduce_median_ms": [float(v) for v in allr[:, 2]],
                    "note": ("per rank: HIP-event average of the gradient kernel; event pair around dist.all_reduce on the kernels' stream "
                             "(collective + waiting for the slowest rank); payload %d B" % (8 * packed.numel()))}
    ms_per_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt

    # PCIe-inclusive step (what a host L-BFGS step pays): params H2D + [loss | gradient] D2H every step
    pcie_value = None
    if world == 1 and not args.forward_only and not args.free_form:
        Xh = guess.to_matrix()
        eng.enable_timing(0)
        for _ in range(2):  # (allocates the pinned staging buffers)
            eng.download(packed)
            eng.upload(Xh)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.loss_grad_packed(eng.upload(Xh), batch, w, gmask, act_slots, B, 0, out=packed)
            host = eng.download(packed)
        torch.cuda.synchronize()
        pcie_value = B * args.steps / (time.perf_counter() - t1)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    fp64_measured = eng.fp64_fma_peak_tflops()  # micro-benchmark on this very device, outside the timed region

    kavg_s = float(np.mean(ktimes)) * 1e-3 if ktimes.size else float("nan")
    kmed_ms = float(np.median(ktimes)) if ktimes.size else float("nan")
    # HBM bytes per launch from the PMC passes of this round (rocprofv3 --pmc cannot run inside bench.py):
    # profiles/r01_traffic.json, produced by scripts/profile_round.sh on the same workload
    traffic, tfile = None, None
    import glob

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))  # the latest round's file last
    if cands and not args.forward_only and not variant:
        tfile = cands[-1]
        tj = json.load(open(tfile))
        if tj.get("B") == B and tj.get("ppp") == args.ppp and args.nvx == 128:
            traffic = tj["hbm_bytes_per_launch"]
    abytes = algorithmic_bytes(eng.NP, with_noise=False) if not args.forward_only else (eng.NP * 8 + 16 + 2 * 1024 * 8)
    achieved = B * abytes / kavg_s / 1e9
    flop_spec = FLOP_PER_SPECTRUM_FWD if args.forward_only else FLOP_PER_SPECTRUM
    # issue-slot utilisation of the VALU from the committed SQ counters of this very kernel and workload (rocprofv3 --pmc cannot run
    # inside bench.py): SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles) -- NOT the algorithmic-flop fraction next to it
    issue = None
    kname = "k_forward_pairs<1" if args.forward_only else "k_spectrum_fused<1, 0"
    if not variant and args.ppp == 1 and args.nvx == 128 and not (args.plan & 2):
        for cf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")), reverse=True):
            cj = json.load(open(cf))
            c = {k: v["avg"] for k, v in cj.get("counters", {}).items()}
            if cj.get("kernel_substring", "").startswith(kname) and cj.get("B", 4096) == B and "SQ_INSTS_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0   # (summed over the 8 XCDs)
                issue = {"issue_slot_frac": c["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * cyc), "valu_instructions_per_launch": c["SQ_INSTS_VALU"],
                         "kernel_cycles": cyc, "counters_file": "profiles/" + os.path.basename(cf)}
                if "SQ_ACTIVE_INST_VALU" in c:
                    issue["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * cyc)
                break
    res = {
        "metric": ("spectra/sec (fwd+grad), 1024-lambda EPW+IAW form factor, batch 4096" if not args.forward_only
                   else "spectra/sec (forward only), 1024-lambda EPW+IAW form factor") + (" [DLM variant: per-lineout f_e]" if args.dlm else "") + (" [free-form f_e variant: + gradient w.r.t. f_e]" if args.free_form else "")
                  + (" [variant: %d points per pixel = %d wavelength samples per feature, batch %d per GPU]" % (args.ppp, 1024 * args.ppp, B) if args.ppp != 1 else ""),
        "value": value,
        "unit": "spectra/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "clock_warmup_ms": args.spin_ms,
        "value_first_burst": first_burst,
        "clock_note": ("value: EXACTLY K steps timed between fences behind an untimed clock warm-up (the W warm-up steps, K steps timed as "
                       "value_first_burst, then the step repeated for clock_warmup_ms): the device clock ramps under sustained load -- the kernel "
                       "of this step takes 1.0 ms in the first launches after a pause and 0.85 ms from the ~100th on (profiles/r03v_ktrace.txt); "
                       "value_first_burst is what rounds 1 and 2 reported as value"),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (("configs[2]: batch 4096 EPW+IAW spectra forward+adjoint (loss+grad as inside the L-BFGS fit loop), "
                          "6 free params {Te, ne, Ti, Va, lam, amp1}, Maxwellian f_e (shared W table)") if not variant else
                         ("configs[2] variant: batch 4096 EPW+IAW spectra forward+adjoint, 6 free params + the 128 values of a free-form "
                          "per-lineout f_e (table adjoints, transposed MFMA GEMM, generator chain rule left to the host)") if args.free_form else
                         ("configs[2] variant: batch 4096 EPW+IAW spectra forward+adjoint, 6 free params {Te, ne, m, amp1, amp2, lam} "
                          "of the reference's 1-D fit test, per-lineout DLM f_e (W and dW/dm tables rebuilt every step)"))
            if not args.forward_only else "configs[1]-like: forward-only EPW+IAW spectra, Maxwellian f_e",
            "lineouts_per_gpu": B,
            "global_batch": B * world,
            "n_lambda": 1024 * args.ppp,
            "n_angles": 10,
            "nvx": args.nvx,
            "free_params": P,
            "parallelism": f"lineout-sharded x{world}, one all-reduce of [3 + B*P] f64 per step" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved * 1e9 / HBM_PEAK,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % (os.path.basename(tfile) if tfile else "-"),
            "algorithmic_bytes_per_launch": B * abytes,
            "kernel": (("k_forward_pairs<1,*,2,1> (512 threads per item: one dispatch round)" if 2 * B <= 512 else
                        "k_fused_prep + k_forward_pairs<1,*,*,2> (256 threads per item, three workgroups per CU when the LDS allows)") if args.forward_only else
                       ("k_spectrum<1,1,2,256,false> (two-sweep kernel with table adjoints)" if args.free_form else
                        ("k_fused_prep + k_spectrum_fused<1,%d,*> (2B 256-thread workgroups, one sweep over the points); k_fused_finish behind it" % (1 if args.dlm else 0)
                         if args.ppp == 1 else
                         "k_spectrum_rows<1,%d> (points_per_pixel %d: one sweep in rounds of 1024 samples, Jacobian rows in a global scratch array)" % (1 if args.dlm else 0, args.ppp)))
                       if not (args.plan & 2) else "k_spectrum<1,1,GM,256,false> (two-sweep kernel)"),
            "kernel_avg_ms": kavg_s * 1e3,
            "kernel_median_ms": kmed_ms,
            "algorithmic_bytes_per_spectrum": abytes,
            "note": "the path is FP64-VALU bound (SURVEY.md 8d); see roofline_fp64",
        },
        "roofline_fp64": {
            "bound": "fp64-valu",
            "achieved": B * flop_spec / kavg_s / 1e12,
            "peak": FP64_PEAK / 1e12,
            "unit": "TFLOP/s",
            "frac": B * flop_spec / kavg_s / FP64_PEAK,
            "algorithmic_flop_per_spectrum": flop_spec,
            "peak_measured_fma": fp64_measured,
            "frac_of_measured": B * flop_spec / kavg_s / 1e12 / fp64_measured,
            "note": ("frac: SURVEY.md 8(d)'s algorithmic flop count (an estimate made before the kernels existed) / kernel time / 78.6 TF; "
                     "issue_slot_frac: what the VALU actually issued, from the committed counters -- the two are different quantities"),
        },
    }
    if issue is not None:
        res["roofline_fp64"].update(issue)
    if per_rank is not None:
        res["per_rank"] = per_rank
    if pcie_value is not None:
        res["value_pcie_inclusive"] = pcie_value
        res["value_note"] = ("value: inputs resident in HBM, [3 | P x B] loss + gradient left in HBM; value_pcie_inclusive: + params "
                             "H2D and [loss | gradient] D2H per step through pinned staging (SURVEY.md 8d protocol)")
    if cpu_res is not None:
        res["cpu_baseline"] = cpu_res
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
