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
        """``distributed=True``: this process holds one contiguous shard of the lineouts (rank r owns the lineouts
        ``distributed.shard_bounds(B_global, world, r)`` of the global batch; the batch need not divide evenly);
        ``vg_loss`` then takes and returns GLOBAL flat vectors and every rank gets the full loss and gradient.
        ``dummy_batch`` may be the rank's LOCAL shard, as in the reference pattern ``LossFunction(cfg, sa, batch)``: the
        loss normalisers (the maxima of the data, loss_function.py:88-92) are then max-reduced over the ranks, so that every
        rank weighs the loss identically -- they are baked into the gradient kernel's weights."""
        self.cfg = cfg
        self.distributed = distributed
        self.pg = process_group
        if cfg["optimizer"]["y_norm"]:  # loss_function.py:88-92
            # (a rank whose shard is empty -- more ranks than lineouts -- contributes -inf to the max over the ranks)
            amax = lambda a: float(np.amax(a)) if np.size(a) or not distributed else -np.inf
            self.i_norm = amax(dummy_batch["i_data"])
            self.e_norm = amax(dummy_batch["e_data"])
            if distributed:
                self.i_norm, self.e_norm = D.allreduce_max([self.i_norm, self.e_norm], process_group)
        else:
            self.i_norm = self.e_norm = 1.0
        if cfg["optimizer"].get("x_norm") and cfg.get("nn", {}).get("use"):
            raise NotImplementedError("nn input normalisation is outside the form-factor path")
        self.multiplex_ang = isinstance(cfg["data"].get("shotnum"), list)
        if self.multiplex_ang:
            raise NotImplementedError("multiplexed angular fits are outside the 1-D form-factor path")
        self.angular = "angular" in cfg["other"]["extraoptions"]["spectype"]
        if self.angular and distributed and int(cfg["parameters"]["electron"].get("fe", {}).get("dim", 1)) != 2:
            raise NotImplementedError("1-D angular fits evaluate one plasma condition in a millisecond: nothing to shard")
        self.fd_step = 1e-5  # normalised units; central differences of the 1-D angular model (see _vg_angular)
        self.force_fd = False  # True: central differences for 2-D distribution functions too (cross-check of the adjoint)
        self.ts_diag = ThomsonScatteringDiagnostic(cfg, scattering_angles=scattering_angles)
        if self.angular and distributed:
            # 2-D angular decks: every rank holds the whole (single) plasma condition; the (lambda, theta) point list of the
            # form factor and of its adjoint is what gets sharded (one all-gather forward, one all-reduce backward)
            import torch.distributed as dist

            self.ts_diag.dist = (dist.get_world_size(process_group), dist.get_rank(process_group), process_group)
        self.unravel_weights = None  # set by the caller exactly as in loops.py:41
        self._gfe = None
        self._dev_batch_src = None   # the batch arrays the device-resident copies were made from (strong references)
        self._dev_batch = None

    # ---- helpers ------------------------------------------------------------------------------
    def _world(self):
        if not self.distributed:
            return 1, 0
        import torch.distributed as dist

        return dist.get_world_size(self.pg), dist.get_rank(self.pg)

    _BATCH_KEYS = ("e_amps", "i_amps", "e_data", "i_data", "noise_e", "noise_i")

    def _device_batch(self, eng, batch, B):
        """Data stay resident on the GPU for the whole fit: converted once per set of batch ARRAYS.  The cache is keyed by
        the identity of the arrays themselves and holds references to them, so a recycled ``id`` of a freed dict or array
        can never alias another batch (the reference feeds freshly built dicts through one loss function, loops.py:133-146,
        postprocess.py:140-149).  Arrays rewritten IN PLACE are not noticed: call ``invalidate_batch()`` after doing that."""
        src = tuple(batch.get(k) for k in self._BATCH_KEYS)
        old = self._dev_batch_src
        if old is None or old[0] != B or len(old[1]) != len(src) or any(a is not b for a, b in zip(old[1], src)):
            d = {}
            for k in ("e_amps", "i_amps"):
                d[k] = eng._vec(batch[k], B)
            for k in ("e_data", "i_data", "noise_e", "noise_i"):
                d[k] = eng._mat(batch.get(k), B)
            self._dev_batch, self._dev_batch_src = d, (B, src)
        return self._dev_batch

    def invalidate_batch(self):
        """Forget the device-resident copy of the batch (after modifying batch arrays in place)."""
        self._dev_batch_src = None
        self._dev_batch = None
        self._ang_dev_src = None

    def _evaluate(self, ts_params: ThomsonParams, batch, want_spectra=False, B_global=None):
        """Local shard: -> (engine, weights, loss sums [3], grad[B_local, NP], ThryE, ThryI), all device tensors.  The 1/N of
        the nanmean uses the TRUE global count ``B_global`` (shards may be uneven or empty, distributed.shard_bounds); without
        it the shard is the whole batch."""
        eng = self.ts_diag.engine(ts_params.activate)
        X = ts_params.to_matrix()
        B = X.shape[0]
        w = eng.loss_weights(B if B_global is None else B_global, self.i_norm, self.e_norm, self.cfg["data"]["ion_loss_scale"])
        if B == 0:   # (more ranks than lineouts: this rank only takes part in the collective)
            z = lambda *shape: eng.torch.zeros(shape, dtype=eng.torch.float64, device=eng.device)
            self._gfe = z(0, eng.nvx) if ts_params.fval is not None else None
            return eng, w, z(3), z(0, X.shape[1]), None, None
        db = self._device_batch(eng, batch, B)
        X = eng.upload(X)  # pinned staging, asynchronous H2D
        if ts_params.fval is not None:  # free-form f_e: explicit tables in, d loss / d fe out
            from . import distribution as Dist

            fe = Dist.arbitrary_1v(ts_params.fval)
            terms, grad, E, I, gfe = eng.loss_grad(X, db, w, ts_params.grad_mask(), fe=fe, want_spectra=want_spectra,
                                                   want_fe_grad=True)
            self._gfe = gfe
            return eng, w, terms, grad, E, I
        self._gfe = None
        terms, grad, E, I = eng.loss_grad(X, db, w, ts_params.grad_mask(), want_spectra=want_spectra)
        return eng, w, terms, grad, E, I

    def _evaluate_packed(self, ts_params: ThomsonParams, batch, act, B_global, b_offset, want_spectra=False):
        """Local shard -> (engine, weights, packed device tensor [3 + P * B_global], ThryE, ThryI): the loss sums and the
        gradient in ravel order over the global batch, this rank's columns filled and the others zero, written by the
        gradient kernels themselves (tsff_loss_grad_packed) -- the buffer of the one all-reduce and of the one
        device-to-host copy of a step."""
        eng = self.ts_diag.engine(ts_params.activate)
        X = ts_params.to_matrix()
        B = X.shape[0]
        w = eng.loss_weights(B_global, self.i_norm, self.e_norm, self.cfg["data"]["ion_loss_scale"])
        if B == 0:   # (more ranks than lineouts: this rank only takes part in the collective)
            return eng, w, eng.torch.zeros(3 + len(act) * B_global, dtype=eng.torch.float64, device=eng.device), None, None
        db = self._device_batch(eng, batch, B)
        packed, E, I = eng.loss_grad_packed(eng.upload(X), db, w, ts_params.grad_mask(), act, B_global, b_offset, want_spectra=want_spectra)
        return eng, w, packed, E, I

    # ---- angular (ARTS) decks ------------------------------------------------------------------
    def _angular_value(self, ts_params: ThomsonParams, batch, want_bar=False):
        """calc_loss / calc_ei_error for one ARTS image (loss_function.py:190-267, 269-341, 364-373): the electron
        feature only (ARTS measures no ion feature), masks on the resolution-unit wavelength axis, nanmean over the
        whole image, (blue + red) / 2.  want_bar: also d value / d ThryE (the seed of the adjoint)."""
        E, _, lamE, _ = self.ts_diag(ts_params, batch)
        d = np.asarray(batch["e_data"], dtype=np.float64)
        method = self.cfg["optimizer"]["loss_method"]
        un = self.e_norm**2
        if method == "l1":
            err = np.abs(d - E) / un
        elif method == "l2":
            err = np.square(d - E) / un
        elif method == "log-cosh":
            err = np.log(np.cosh(d - E))
        else:
            err = E - d * np.log(E)
        ext, r = self.cfg["other"]["extraoptions"], self.cfg["data"]["fit_rng"]
        e_error = 0.0
        wcol = np.zeros(E.shape[1])  # d e_error / d err[:, j] (the same for every row)
        if ext["fit_EPWb"]:
            blue = (lamE > r["blue_min"]) & (lamE < r["blue_max"])
            e_error += float(np.mean(err[:, blue]))
            wcol[blue] += 1.0 / (E.shape[0] * max(int(blue.sum()), 1))
        if ext["fit_EPWr"]:
            red = (lamE > r["red_min"]) & (lamE < r["red_max"])
            e_error += float(np.mean(err[:, red]))
            wcol[red] += 1.0 / (E.shape[0] * max(int(red.sum()), 1))
            if ext["fit_EPWb"]:
                e_error *= 0.5
                wcol *= 0.5
        if not want_bar:
            return e_error, E
        if method == "l1":
            derr = -np.sign(d - E) / un
        elif method == "l2":
            derr = -2.0 * (d - E) / un
        elif method == "log-cosh":
            derr = -np.tanh(d - E)
        else:
            derr = 1.0 - d / E
        return e_error, E, derr * wcol[None, :]

    def _angular_value_device(self, eng, ts_params: ThomsonParams, batch, want_E: bool):
        """_angular_value(want_bar=True) with the image, the data, the masks and the loss seed on the device (a handful of
        element-wise torch operations on [rows, n_lam]; one scalar comes back): -> (value, ThryE or None, Ebar device)."""
        torch = eng.torch
        self.ts_diag._angular(eng, ts_params, batch, to_host=False)
        ctx = self.ts_diag._angular_ctx
        src = (batch["e_data"], batch["noise_e"])   # (strong references: see _device_batch)
        old_src = getattr(self, "_ang_dev_src", None)
        if old_src is None or old_src[0] is not src[0] or old_src[1] is not src[1]:
            lamE = ctx["lamE"]
            ext, r = self.cfg["other"]["extraoptions"], self.cfg["data"]["fit_rng"]
            rows = ctx["E_dev"].shape[0]
            wcol = np.zeros(lamE.size)
            nterm = 0
            for on, lo, hi in ((ext["fit_EPWb"], r["blue_min"], r["blue_max"]), (ext["fit_EPWr"], r["red_min"], r["red_max"])):
                if on:
                    m = (lamE > lo) & (lamE < hi)
                    wcol[m] += 1.0 / (rows * max(int(m.sum()), 1))
                    nterm += 1
            if nterm == 2:
                wcol *= 0.5
            self._ang_dev = dict(d=eng.dev(np.ascontiguousarray(batch["e_data"], dtype=np.float64)), wcol=eng.dev(wcol)[None, :],
                                 noise=eng.dev(np.array(np.atleast_1d(np.asarray(batch["noise_e"], dtype=np.float64)))))
            self._ang_dev_src = src
        dv = self._ang_dev
        Et = ctx["E_dev"] + dv["noise"]
        d, un, method = dv["d"], self.e_norm**2, self.cfg["optimizer"]["loss_method"]
        if method == "l1":
            err, derr = (d - Et).abs() / un, -torch.sign(d - Et) / un
        elif method == "l2":
            err, derr = (d - Et) ** 2 / un, -2.0 * (d - Et) / un
        elif method == "log-cosh":
            err, derr = torch.log(torch.cosh(d - Et)), -torch.tanh(d - Et)
        else:
            err, derr = Et - d * torch.log(Et), 1.0 - d / Et
        value = float((err * dv["wcol"]).sum())
        return value, (Et.cpu().numpy() if want_E else None), derr * dv["wcol"]

    def _vg_angular(self, diff_weights, static_weights, batch):
        """Value and gradient of the angular model by the hand-written adjoint -- loss seed -> tsff_ats_adjoint ->
        tsff_form_factor_2d_grad (2-D distribution functions: one rotated projection of the table per (wavelength, angle)
        point) or tsff_form_factor_grad (1-D) -> chain rule of the parameter transform and of the distribution-function
        generator -- at the cost of a few forward evaluations whatever the number of parameters, which is what
        reverse-mode JAX gives the reference (loops.py:167-275).  ``force_fd`` switches to central differences over
        the trainable scalar leaves (step ``fd_step`` in normalised units), kept as a cross-check."""
        ts_params = tree.combine(static_weights, diff_weights)
        if not getattr(self, "force_fd", False):
            return self._vg_angular_adjoint(ts_params, diff_weights, batch)
        value, E = self._angular_value(ts_params, batch)
        grads = []
        for k, ((name, s), v) in enumerate(zip(diff_weights.slots, diff_weights.values)):
            if s in (tree.FVAL_SLOT, tree.FVAL2D_SLOT):
                raise NotImplementedError("free-form distribution functions are not fitted through finite differences")
            g = np.zeros_like(v)
            for idx in np.ndindex(v.shape):
                vals = []
                for sgn in (+1.0, -1.0):
                    pert = [u.copy() for u in diff_weights.values]
                    pert[k][idx] += sgn * self.fd_step
                    vals.append(self._angular_value(tree.combine(static_weights, tree.DiffParams(diff_weights.slots, pert)), batch)[0])
                g[idx] = (vals[0] - vals[1]) / (2 * self.fd_step)
            grads.append(g)
        return value, E, ts_params, tree.DiffParams(diff_weights.slots, grads)

    def _vg_angular_adjoint(self, ts_params: ThomsonParams, diff_weights, batch):
        from . import distribution as Dist

        eng = self.ts_diag.engine(ts_params.activate)
        want_E = self.cfg["optimizer"]["method"] != "l-bfgs-b"   # (the optax branch returns the image as aux)
        value, E, Ebar = self._angular_value_device(eng, ts_params, batch, want_E)
        ctx = self.ts_diag._angular_ctx
        phys, P = ctx["phys"], ctx["P"]
        p = phys[0]
        gen = self.cfg["parameters"]["general"]
        Pbar, (a1b, a2b) = eng.ats_adjoint(P[0], ctx["e_amps"], p[L.P_LAM], p[L.P_AMP1], p[L.P_AMP2], Ebar)
        sm = ts_params.slots
        gfe_h = None
        if ts_params.fe_dim == 2:
            want_table = any(s in (tree.GEN2D_SLOT, tree.FVAL2D_SLOT) for _, s in diff_weights.slots)
            if self.ts_diag.dist is not None and self.ts_diag.dist[0] > 1:
                gp, gfe = D.form_factor_2d_grad_sharded(eng, 0, phys, ctx["fe2"], Pbar.reshape(P.shape), gen["ud"]["angle"],
                                                        gen["Va"]["angle"], *self.ts_diag.dist, want_table=want_table, use_saved=True)
            else:
                gp, gfe = eng.form_factor_2d_grad(0, phys, ctx["fe2"], Pbar.reshape(P.shape), gen["ud"]["angle"],
                                                  gen["Va"]["angle"], want_table=want_table, use_saved=True)
        else:
            want_fe = any(s in (tree.FVAL_SLOT, L.P_M) for _, s in diff_weights.slots)
            gp, gfe = eng.form_factor_grad(0, phys, ctx["fe1"], Pbar.reshape(P.shape), want_fe=want_fe)
        gphys = gp.cpu().numpy()[0]
        gfe_h = gfe.cpu().numpy() if gfe is not None else None
        if ts_params.fe_dim == 1 and sm.has_m and gfe_h is not None:
            # DLM order: f_e = dlm(m) is a cheap host generator; its derivative by central differences, contracted
            # with d loss / d f_e from the GPU
            nvx, mval, hm = ctx["fe1"].shape[1], float(phys[0, L.P_M]), 1e-6
            gphys[L.P_M] = float(np.dot(gfe_h[0], (Dist.dlm(mval + hm, nvx) - Dist.dlm(mval - hm, nvx)) / (2 * hm)))
        gphys[L.P_AMP1] += a1b
        gphys[L.P_AMP2] += a2b
        for i in range(1, sm.n_ion):  # tied ion temperatures (ts_params.py: Ti "same")
            if sm.ti_same[i]:
                gphys[L.P_ION0 + L.ION_TI] += gphys[L.P_ION0 + 4 * i + L.ION_TI]
                gphys[L.P_ION0 + 4 * i + L.ION_TI] = 0.0
        # physical -> normalised leaves: phys = act(x) * scale + shift
        x = ts_params.X[0]
        sg = 1.0 / (1.0 + np.exp(-x))
        gnorm = gphys * sm.scale * np.where(sm.sigmoid.astype(bool), sg * (1.0 - sg), 1.0)
        grads = []
        for (name, s), v in zip(diff_weights.slots, diff_weights.values):
            if s == tree.FVAL_SLOT:
                grads.append(Dist.arbitrary_1v_vjp(ts_params.fval, gfe_h).reshape(v.shape))
            elif s == tree.GEN2D_SLOT:
                grads.append(ts_params.sph.vjp(gfe_h).reshape(v.shape))
            elif s == tree.FVAL2D_SLOT:
                grads.append(Dist.arbitrary_2v_vjp(ts_params.fval2d, ts_params.learn_log, gfe_h).reshape(v.shape))
            else:
                grads.append(np.full_like(v, gnorm[s]))
        return value, E, ts_params, tree.DiffParams(diff_weights.slots, grads)

    # ---- reference API --------------------------------------------------------------------------
    def vg_loss(self, diff_weights, static_weights, batch: Dict):
        """loss_function.py:128-168.  l-bfgs-b: (float value, flat float64 gradient); otherwise
        ((value, aux), gradient as a DiffParams)."""
        import torch

        lbfgs = self.cfg["optimizer"]["method"] == "l-bfgs-b"
        if getattr(self, "angular", False):
            if lbfgs:
                diff_weights = self.unravel_weights(np.asarray(diff_weights, dtype=np.float64))
            value, E, ts_params, grad = self._vg_angular(diff_weights, static_weights, batch)
            if lbfgs:
                return value, grad.ravel()
            return (value, [E, ts_params()]), grad
        world, rank = self._world()
        if lbfgs:
            diff_weights = self.unravel_weights(np.asarray(diff_weights, dtype=np.float64))
        diff_global = diff_weights
        Bg = int(diff_weights.values[0].shape[0])
        lo, hi = D.shard_bounds(Bg, world, rank) if world > 1 else (0, Bg)
        if world > 1:
            # global -> local slice of every trainable leaf
            diff_weights = tree.DiffParams(diff_weights.slots, [v[lo:hi] for v in diff_weights.values])
        ts_params = tree.combine(static_weights, diff_weights)
        act = [s for _, s in ts_params.slots.active_leaves]
        if ts_params.fval is not None and ts_params.slots.fval_active:
            # free-form f_e: nvx more rows (d loss / d fe, chained on the host); packed by torch, one all-reduce all the same
            eng, w, terms, grad, E, I = self._evaluate(ts_params, batch, want_spectra=not lbfgs, B_global=Bg)
            if hasattr(eng, "pack_fe_rows") and grad.shape[0] > 0:
                # [3 | (P + nvx) x B_global] in ravel order, this rank's columns filled, by one transposing kernel (tsff_pack_fe_rows)
                out = eng.pack_fe_rows(terms, grad, self._gfe, act, Bg, lo)
                if world > 1:
                    import torch.distributed as dist

                    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.pg)
            else:   # (an empty shard, or the CPU rehearsal of the distributed tests: the torch form of the same buffer)
                gact = torch.cat([grad[:, act].t(), self._gfe.t()]).contiguous()  # [P + nvx, B_local], ravel order
                out = D.allreduce_loss_grad(terms, gact, world, rank, self.pg, B_global=Bg, b_offset=lo)
            host = eng.download(out)
            flat = self._chain_fval(host[3:], len(act), diff_global)
        else:
            eng, w, out, E, I = self._evaluate_packed(ts_params, batch, act, Bg, lo, want_spectra=not lbfgs)
            if world > 1:  # the one collective per step, in place on the buffer the kernels wrote
                import torch.distributed as dist

                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.pg)
            host = eng.download(out)  # single D2H copy: 3 + P * B_global doubles (pinned staging on the engine)
            flat = host[3:]
        value = float(np.dot(host[:3], w))
        if lbfgs:
            return value, flat
        aux = [E.cpu().numpy() if E is not None else None, ts_params()]
        return (value, aux), diff_global.like(flat)

    def _chain_fval(self, flat, P, diff_global):
        """[P + nvx, B] rows (scalar leaves, then d loss / d fe) -> the reference's ravel order with the
        Arbitrary1V chain rule applied (d loss / d fval = J^T d loss / d fe, base.py:201-204)."""
        from . import distribution as Dist

        fv = next(v for (_, s), v in zip(diff_global.slots, diff_global.values) if s == tree.FVAL_SLOT)
        Bg, nvx = fv.shape
        rows = flat.reshape(P + nvx, Bg)
        gfval = Dist.arbitrary_1v_vjp(fv, rows[P:].T)
        out, k = [], 0
        for _, s in diff_global.slots:
            if s == tree.FVAL_SLOT:
                out.append(gfval.ravel())
            else:
                out.append(rows[k])
                k += 1
        return np.concatenate(out)

    def loss(self, weights, batch: Dict):
        """loss_function.py:344-362."""
        if getattr(self, "angular", False) and self.cfg["optimizer"]["method"] != "l-bfgs-b":
            value, E = self._angular_value(weights, batch)
            return value, [E, weights()]
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

    def h_loss_wrt_params(self, weights: ThomsonParams, batch: Dict, step: float = 1e-7):
        """Hessian of the reference's ``_loss_for_hess_fn_`` (loss_function.py:173-188: denominators |data| + 1e-10,
        sum reduce, i_error + e_error) w.r.t. the trainable normalised leaves, in the nested layout
        ``get_sigmas`` reads (postprocess.py:188-251): ``hess[species][key][species2][key2]`` is a [B, B] matrix whose
        diagonal holds the per-lineout second derivatives (lineouts do not couple).

        The reference differentiates twice with JAX; here the analytic gradient of the HIP adjoint is
        central-differenced: 2 * P gradient evaluations of the whole batch (every lineout is perturbed at once).
        The step is small on purpose: JAX's second derivative of the piecewise-linear table lookups (Z', W) is
        zero inside a table cell, and a step of 1e-7 (normalised units) keeps nearly every sample inside its
        cell, so the difference quotient reproduces that convention (2e-6 relative against double-backward
        autodiff of the oracle; with 1e-4 the curvature of the tables leaks in and entries change sign)."""
        eng = self.ts_diag.engine(weights.activate)
        X = weights.to_matrix()
        B = X.shape[0]
        db = self._device_batch(eng, batch, B)
        c = 0.5 if (eng.fit_blue and eng.fit_red) else 1.0
        w = np.array([1.0 if eng.fit_iaw else 0.0, c if eng.fit_blue else 0.0, c if eng.fit_red else 0.0])
        leaves = weights.slots.active_leaves
        act = [s for _, s in leaves]
        gm = weights.grad_mask()
        H = np.zeros((B, len(act), len(act)))
        eng.set_denominator_mode(2)
        try:
            for k, s in enumerate(act):
                Xp, Xm = X.copy(), X.copy()
                Xp[:, s] += step
                Xm[:, s] -= step
                gp = eng.loss_grad(Xp, db, w, gm)[1][:, act].cpu().numpy()
                gn = eng.loss_grad(Xm, db, w, gm)[1][:, act].cpu().numpy()
                H[:, :, k] = (gp - gn) / (2 * step)
        finally:
            eng.set_denominator_mode(0)
        H = 0.5 * (H + np.transpose(H, (0, 2, 1)))
        hess = {}
        for a, ((sp1, k1), _) in enumerate(leaves):
            for b_, ((sp2, k2), _) in enumerate(leaves):
                hess.setdefault(sp1, {}).setdefault(k1, {}).setdefault(sp2, {})[k2] = np.diag(H[:, a, b_])
        return hess
