"""Multi-GPU layout of a fit: one process per GPU, lineouts sharded in contiguous blocks, ONE
all-reduce (RCCL over xGMI; gloo in the CPU tests) of ``[S_iaw, S_blue, S_red | gradient]`` per loss
evaluation.  Lineouts are independent through the whole forward model (the reference ``vmap``s them,
core/thomson_diagnostic.py:35-36); the only coupling is the nanmean over the whole batch
(inverse/loss_function.py:237,249,261), i.e. three scalar sums and the 1/N factor.

Each rank fills its own block of the gradient and zeros elsewhere, so after the sum every rank
holds the full loss and the full gradient and can run an identical host L-BFGS step (no broadcast).
The functions here only touch tensors -- they run on CPU tensors with gloo as well.
"""
from __future__ import annotations

import os

import numpy as np


def init_from_env(backend: str | None = None):
    """torch.distributed initialisation from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a one-GPU box (several ranks sharing device 0 over gloo); never set in production
    backend = backend or os.environ.get("TSFF_DIST_BACKEND")
    if "TSFF_FORCE_DEVICE" in os.environ:
        local = int(os.environ["TSFF_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return world, rank, local


def shard_bounds(B_global: int, world: int, rank: int):
    """Contiguous block [lo, hi) of rank ``rank``: chunks of ceil(B_global / world) lineouts, the last one(s) shorter or
    empty when the batch does not divide evenly (reference batches are arbitrary, loops.py:133-146).  The 1/N of the loss
    always uses the true global count (``Engine.loss_weights(B_global, ...)``)."""
    chunk = -(-B_global // world)
    lo = min(rank * chunk, B_global)
    return lo, min(lo + chunk, B_global)


def allreduce_max(values, group=None):
    """Max over the ranks of a few host scalars (the loss normalisers of a lineout-sharded fit) -> list of floats."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [float(v) for v in values]
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(v) for v in t.cpu()]


def pack_local(terms, grad_local, B_global: int, b_offset: int):
    """[3] loss sums and [P, B_local] gradient rows of this rank -> the packed buffer ``[3 | P x B_global]`` of the step's
    all-reduce with this rank's columns [b_offset, b_offset + B_local) filled and zeros elsewhere.  (The plain fit path gets
    this buffer from the gradient kernels themselves, tsff_loss_grad_packed; this torch form serves the free-form f_e
    variant, whose extra rows come from a second kernel chain.)"""
    import torch

    P, Bl = grad_local.shape
    buf = torch.zeros(3 + P * B_global, dtype=grad_local.dtype, device=grad_local.device)
    buf[:3] = terms
    buf[3:].view(P, B_global)[:, b_offset:b_offset + Bl] = grad_local
    return buf


def allreduce_loss_grad(terms, grad_local, world: int, rank: int, group=None, B_global=None, b_offset=None):
    """terms: [3] un-weighted masked sums of this rank; grad_local: [P, B_local] (parameter-major).
    Returns the packed ``[S_iaw, S_blue, S_red | grad (P x B_global)]`` -- identical on every rank.

    The one collective of a fit step: a single all-reduce(sum) in which every rank fills its own block of the gradient
    and zeros elsewhere (24 B + 8 P B_global bytes: 192 KiB at one rank of 4096 lineouts, 1.5 MiB at eight).  Shards
    follow ``shard_bounds`` (uneven batches allowed) unless ``B_global`` / ``b_offset`` say otherwise."""
    import torch

    P, Bl = grad_local.shape
    if world == 1:
        return torch.cat([terms.reshape(3), grad_local.reshape(-1)])
    import torch.distributed as dist

    if B_global is None:
        B_global, b_offset = Bl * world, rank * Bl
    buf = pack_local(terms, grad_local, B_global, b_offset)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def allgather_loss_grad(terms, grad_local, world: int, rank: int, group=None):
    """Same contract as allreduce_loss_grad with an all-gather of each rank's ``[3 + P B_local]`` block instead: half
    the bytes on the wire and no floating-point reduction in the collective (the three loss terms are summed in rank
    order afterwards).  Equal shards only.  Kept as an alternative; the fit loop uses the all-reduce."""
    import torch

    P, Bl = grad_local.shape
    if world == 1:
        return torch.cat([terms.reshape(3), grad_local.reshape(-1)])
    import torch.distributed as dist

    mine = torch.cat([terms.reshape(3), grad_local.reshape(-1)])
    allb = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(allb, mine, group=group)
    allb = allb.view(world, 3 + P * Bl)
    tot = allb[0, :3].clone()
    for r in range(1, world):
        tot = tot + allb[r, :3]
    grad = allb[:, 3:].reshape(world, P, Bl).permute(1, 0, 2).reshape(-1)  # parameter-major over the global batch
    return torch.cat([tot, grad])


def point_range(n_points: int, world: int, rank: int):
    """Contiguous slice of the flat (lineout, gradient point, wavelength, angle) list of the 2-D angular path for
    rank ``rank``: equal chunks of ceil(n / world) points, the last one shorter (form_factor.py:431-447 shards the
    same list with a NamedSharding)."""
    chunk = -(-n_points // world)
    lo = min(rank * chunk, n_points)
    return lo, min(lo + chunk, n_points), chunk


def form_factor_2d_sharded(engine, feature, phys, fe2d, ud_angle, va_angle, world: int, rank: int, group=None, save=False):
    """FormFactor.calc_in_2D over the ranks of a node: f_e and parameters replicated, each rank evaluates its slice of
    the point list (tsff_form_factor_2d_range), one all-gather of equal (padded) chunks assembles P on every rank.
    No reduction is involved; the result is bit-identical to the single-rank one."""
    import torch

    if world == 1:
        return engine.form_factor_2d(feature, phys, fe2d, ud_angle, va_angle, save=save)
    import torch.distributed as dist

    B = np.asarray(phys).reshape(-1, engine.NP).shape[0] if not hasattr(phys, "shape") else int(phys.reshape(-1, engine.NP).shape[0])
    shape = (B, int(engine._cfg_struct.num_grad_points), engine.npts, int(engine._cfg_struct.n_angles))
    n = int(np.prod(shape))
    lo, hi, chunk = point_range(n, world, rank)
    full = torch.zeros(chunk * world, dtype=torch.float64, device=engine.device)
    view = full[:n].view(shape)
    engine.form_factor_2d(feature, phys, fe2d, ud_angle, va_angle, point_range=(lo, hi), out=view, save=save)
    mine = full[rank * chunk : (rank + 1) * chunk].clone()
    dist.all_gather_into_tensor(full, mine, group=group)
    return full[:n].view(shape)


def form_factor_2d_grad_sharded(engine, feature, phys, fe2d, Pbar, ud_angle, va_angle, world: int, rank: int, group=None,
                                want_table=True, use_saved=False):
    """Adjoint of the 2-D path over the ranks of a node: every rank holds Pbar, the table and the parameters, reverses its
    slice of the point list (tsff_form_factor_2d_grad with a point range) and ONE all-reduce sums the packed
    [grad_phys | grad_fe2d] -- both are sums over points.  -> (grad_phys [B, NP], grad_fe2d [nv, nv] or None)."""
    import torch

    if world == 1:
        return engine.form_factor_2d_grad(feature, phys, fe2d, Pbar, ud_angle, va_angle, want_table=want_table, use_saved=use_saved)
    import torch.distributed as dist

    n = int(Pbar.numel()) if hasattr(Pbar, "numel") else int(np.asarray(Pbar).size)
    lo, hi, _ = point_range(n, world, rank)
    gp, gf = engine.form_factor_2d_grad(feature, phys, fe2d, Pbar, ud_angle, va_angle, want_table=want_table, point_range=(lo, hi),
                                        use_saved=use_saved)
    packed = torch.cat([gp.reshape(-1), gf.reshape(-1)]) if gf is not None else gp.reshape(-1).clone()
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    gp_out = packed[: gp.numel()].view_as(gp)
    return gp_out, (packed[gp.numel():].view_as(gf) if gf is not None else None)
