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
    """Contiguous block [lo, hi) of rank ``rank``; the batch must divide evenly (the 1/N factor of
    the loss assumes every rank holds B_global / world lineouts)."""
    if B_global % world:
        raise ValueError(f"global batch {B_global} does not divide over {world} ranks")
    Bl = B_global // world
    return rank * Bl, (rank + 1) * Bl


def allreduce_loss_grad(terms, grad_local, world: int, rank: int, group=None):
    """terms: [3] un-weighted masked sums of this rank; grad_local: [P, B_local] (parameter-major).
    Returns (terms_global [3], grad_global_flat [P * B_global]) -- identical on every rank."""
    import torch

    P, Bl = grad_local.shape
    if world == 1:
        return terms, grad_local.reshape(-1)
    import torch.distributed as dist

    buf = torch.zeros(3 + P * Bl * world, dtype=grad_local.dtype, device=grad_local.device)
    buf[:3] = terms
    buf[3:].view(P, world, Bl)[:, rank, :] = grad_local
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:3], buf[3:]
