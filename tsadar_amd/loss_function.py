"""Drop-in for ``tsadar.inverse.loss_function.LossFunction`` (reference
inverse/loss_function.py:22-418): ``vg_loss`` / ``loss`` / ``array_loss`` with the reference
signatures, evaluated by the HIP engine (forward + hand-written adjoint), optionally sharded over
the GPUs of one node with a single RCCL all-reduce of [loss sums | gradient] per evaluation.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import _lib as L
from . import tree
from . import distributed as D
from .diagnostic import ThomsonScatteringDiagnostic
from .params import ThomsonParams


class LossFunction:
    def __init__(self, cfg: Dict, scattering_angles, dummy_batch, process_group=None, distributed: bool = False):
        """``distributed=True``: this process holds one contiguous shard of the lineouts (rank r owns
        lineouts [r*B_local, (r+1)*B_local) of the global batch); ``vg_loss`` then takes and returns
        GLOBAL flat vectors and every rank gets the full loss and gradient."""
        self.cfg = cfg
        if cfg["optimizer"]["y_norm"]:  # loss_function.py:88-92
            self.i_norm = float(np.amax(dummy_batch["i_data"]))
            self.e_norm = float(np.amax(dummy_batch["e_data"]))
        else:
            self.i_norm = self.e_norm = 1.0
        if cfg["optimizer"].get("x_norm") and cfg.get("nn", {}).get("use"):
            raise NotImplementedError("nn input normalisation is outside the form-factor path")
        self.multiplex_ang = isinstance(cfg["data"].get("shotnum"), list)
        if self.multiplex_ang:
            raise NotImplementedError("multiplexed angular fits are outside the 1-D form-factor path")
        self.ts_diag = ThomsonScatteringDiagnostic(cfg, scattering_angles=scattering_angles)
        self.distributed = distributed
        self.pg = process_group
        self.unravel_weights = None  # set by the caller exactly as in loops.py:41
        self._dev_batch_key = None
        self._dev_batch = None

    # ---- helpers ------------------------------------------------------------------------------
    def _world(self):
        if not self.distributed:
            return 1, 0
        import torch.distributed as dist

        return dist.get_world_size(self.pg), dist.get_rank(self.pg)

    def _device_batch(self, eng, batch, B):
        """Data stay resident on the GPU for the whole fit: convert once per batch object."""
        key = (id(batch), B)
        if self._dev_batch_key != key:
            d = {}
            for k in ("e_amps", "i_amps"):
                d[k] = eng._vec(batch[k], B)
            for k in ("e_data", "i_data", "noise_e", "noise_i"):
                d[k] = eng._mat(batch.get(k), B)
            self._dev_batch, self._dev_batch_key = d, key
        return self._dev_batch

    def _evaluate(self, ts_params: ThomsonParams, batch, want_spectra=False):
        """Local shard: -> (value, grad[B_local, NP] numpy, ThryE, ThryI).  In distributed mode the
        value is the global loss (after the all-reduce) and grad the LOCAL block."""
        import torch

        eng = self.ts_diag.engine(ts_params.activate)
        X = ts_params.to_matrix()
        B = X.shape[0]
        world, rank = self._world()
        w = eng.loss_weights(B * world, self.i_norm, self.e_norm, self.cfg["data"]["ion_loss_scale"])
        db = self._device_batch(eng, batch, B)
        terms, grad, E, I = eng.loss_grad(X, db, w, ts_params.grad_mask(), want_spectra=want_spectra)
        return eng, w, terms, grad, E, I

    # ---- reference API --------------------------------------------------------------------------
    def vg_loss(self, diff_weights, static_weights, batch: Dict):
        """loss_function.py:128-168.  l-bfgs-b: (float value, flat float64 gradient); otherwise
        ((value, aux), gradient as a DiffParams)."""
        import torch

        lbfgs = self.cfg["optimizer"]["method"] == "l-bfgs-b"
        world, rank = self._world()
        if lbfgs:
            diff_weights = self.unravel_weights(np.asarray(diff_weights, dtype=np.float64))
        diff_global = diff_weights
        if world > 1:
            # global -> local slice of every trainable leaf
            lo, hi = D.shard_bounds(diff_weights.values[0].shape[0], world, rank)
            diff_weights = tree.DiffParams(diff_weights.slots, [v[lo:hi] for v in diff_weights.values])
        ts_params = tree.combine(static_weights, diff_weights)
        eng, w, terms, grad, E, I = self._evaluate(ts_params, batch, want_spectra=not lbfgs)
        act = [s for _, s in ts_params.slots.active_leaves]
        gact = grad[:, act].t().contiguous()  # [P, B_local], ravel order
        terms, gflat = D.allreduce_loss_grad(terms, gact, world, rank, self.pg)  # the one collective per step
        host = torch.cat([terms, gflat]).cpu().numpy()  # single D2H copy: 3 + P*B doubles
        value = float(np.dot(host[:3], w))
        flat = host[3:]
        if lbfgs:
            return value, flat
        aux = [E.cpu().numpy() if E is not None else None, ts_params()]
        return (value, aux), diff_global.like(flat)

    def loss(self, weights, batch: Dict):
        """loss_function.py:344-362."""
        if self.cfg["optimizer"]["method"] == "l-bfgs-b":
            raise NotImplementedError("loss() with flat weights needs unravel_pytree, which the reference never sets "
                                      "(loss_function.py:357); use vg_loss")
        eng, w, terms, grad, E, I = self._evaluate(weights, batch, want_spectra=True)
        value = float(np.dot(terms.cpu().numpy(), w))
        return value, [E.cpu().numpy(), weights()]

    def array_loss(self, weights: ThomsonParams, batch: Dict):
        """loss_function.py:375-384 (``post_loss``): per-lineout nanmean(axis=1) with the theory spectra as
        denominators.  -> (loss[B], sqdev{ele, ion}, ThryE, ThryI, params)."""
        eng = self.ts_diag.engine(weights.activate)
        X = weights.to_matrix()
        B = X.shape[0]
        sums, sqe, sqi, E, I = eng.array_loss(X, self._device_batch(eng, batch, B))
        s = sums.cpu().numpy()
        i_err = s[:, 0] / max(eng.n_iaw, 1) if eng.fit_iaw else np.zeros(B)
        e_err = np.zeros(B)
        if eng.fit_blue:
            e_err = e_err + s[:, 1] / max(eng.n_blue, 1)
        if eng.fit_red:
            e_err = e_err + s[:, 2] / max(eng.n_red, 1)
            if eng.fit_blue:
                e_err = e_err * 0.5
        total = self.cfg["data"]["ion_loss_scale"] * i_err + e_err
        sqdev = {"ele": sqe.cpu().numpy(), "ion": sqi.cpu().numpy()}
        return total, sqdev, E.cpu().numpy(), I.cpu().numpy(), weights()

    def h_loss_wrt_params(self, weights, batch):
        raise NotImplementedError("Hessian of the loss (calc_sigmas) is not implemented yet")
