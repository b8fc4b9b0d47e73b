"""The N > 1 path on CPU: two gloo ranks, each holding one contiguous shard of the lineouts, one
all-reduce of [3 loss sums | gradient] per evaluation.  The per-shard evaluation is done by the oracle
(the HIP engine needs a GPU); everything around it -- global->local slicing of the flat scipy vector,
the 1/N_global weights, the packed all-reduce, the global flat gradient -- is the product code of
tsadar_amd.loss_function / tsadar_amd.distributed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(B):
    import decks
    import util
    from oracle import tsadar_oracle as orc

    cfg = decks.deck_fit(active=("Te", "ne", "Ti", "Va", "lam", "amp1"))
    sa = util.sa_fit(B)
    batch = util.synthetic_batch(cfg, sa, B, seed=31)
    normed = util.random_lineouts(cfg, B, seed=77)
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    return cfg, sa, batch, normed, i_norm, e_norm


class _OracleEngine:
    """Stands in for tsadar_amd.engine.Engine in this CPU test: same attributes loss_weights() reads."""

    def __init__(self, cfg):
        import util
        from oracle import tsadar_oracle as orc
        from tsadar_amd import engine as E

        ext = cfg["other"]["extraoptions"]
        self.cfg = cfg
        lamE = E.wavelength_axis_nm(cfg["other"]["lamrangE"], cfg["other"]["npts"]).reshape(1024, -1).mean(axis=1)
        lamI = E.wavelength_axis_nm(cfg["other"]["lamrangI"], cfg["other"]["npts"]).reshape(1024, -1).mean(axis=1)
        iaw, blue, red = orc.fit_masks(cfg, lamE, lamI)
        self.fit_iaw, self.fit_blue, self.fit_red = ext["fit_IAW"], ext["fit_EPWb"], ext["fit_EPWr"]
        self.n_iaw, self.n_blue, self.n_red = int(iaw.sum()), int(blue.sum()), int(red.sum())


def _oracle_evaluate(self, ts_params, batch, want_spectra=False):
    """Replacement for LossFunction._evaluate: oracle masked sums and the gradient of the weighted total
    w.r.t. this shard's normalised parameters, as [B_local, NP] like the engine returns it."""
    import util
    from oracle import tsadar_oracle_torch as ot
    from tsadar_amd.engine import Engine

    X = ts_params.to_matrix()
    B = X.shape[0]
    world, rank = self._world()
    eng = _OracleEngine(self.cfg)
    w = Engine.loss_weights(eng, B * world, self.i_norm, self.e_norm, self.cfg["data"]["ion_loss_scale"])
    names = ["Te", "ne", "m", "Ti_1", "Z_1", "A_1", "fract_1", "lam", "amp1", "amp2", "amp3", "ne_gradient", "Te_gradient", "ud", "Va"]
    normed = {k: torch.tensor(X[:, util.slot_of(k)], dtype=torch.float64, requires_grad=True) for k in names}
    sa = dict(sa=util.P9["sa"], weights=util.P9["weights"] * np.ones([B, 10]))
    S, N, E, I = ot.masked_sums(self.cfg, sa, normed, batch)
    total = (S * torch.as_tensor(w)).sum()
    grads = torch.autograd.grad(total, [normed[k] for k in names], allow_unused=True)
    G = torch.zeros((B, X.shape[1]), dtype=torch.float64)
    for k, g in zip(names, grads):
        if g is not None and ts_params.slots.active[util.slot_of(k)]:
            G[:, util.slot_of(k)] = g
    return eng, w, S.detach(), G, E.detach(), I.detach()


def _rank_main(rank, world, port, B, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tsadar_amd import ThomsonParams, distributed as D, tree
    from tsadar_amd.loss_function import LossFunction
    import util

    w_, r_, _ = D.init_from_env(backend="gloo")
    assert (w_, r_) == (world, rank)
    cfg, sa, batch, normed, i_norm, e_norm = _setup(B)
    lo, hi = D.shard_bounds(B, world, rank)
    local = {k: np.asarray(v)[lo:hi] for k, v in batch.items()}
    LossFunction._evaluate = _oracle_evaluate
    LossFunction.__init__ = lambda self, cfg, sa, dummy, process_group=None, distributed=False: None
    lf = LossFunction(cfg, sa, batch)
    lf.cfg, lf.i_norm, lf.e_norm, lf.distributed, lf.pg = cfg, i_norm, e_norm, True, None
    # the caller's side, exactly as loops.py:36-41 does it (global parameters on every rank)
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    tp.X[:] = util.normed_to_matrix(normed, 1)
    diff, static_g = tree.partition(tp)
    x0, lf.unravel_weights = tree.ravel_pytree(diff)
    tp_local = ThomsonParams(cfg["parameters"], hi - lo, batch=True, activate=True)
    tp_local.X[:] = tp.X[lo:hi]
    static_l = tree.StaticParams(tp_local)
    value, flat = lf.vg_loss(x0, static_l, local)
    np.save(os.path.join(out, f"flat_{rank}.npy"), np.concatenate([[value], flat]))
    # the all-gather alternative returns the same tensors as the all-reduce
    t = torch.arange(3, dtype=torch.float64) + rank
    g = torch.arange(10, dtype=torch.float64).reshape(2, 5) * (rank + 1)
    ta, ga = D.allreduce_loss_grad(t, g, world, rank)
    tg, gg = D.allgather_loss_grad(t, g, world, rank)
    assert torch.equal(ta, tg) and torch.equal(ga, gg)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_loss_and_gradient_equal_single_rank(tmp_path):
    B, world = 4, 2
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "flat_0.npy")
    r1 = np.load(tmp_path / "flat_1.npy")
    np.testing.assert_array_equal(r0, r1)  # every rank holds the full loss and gradient
    # single-process reference on the whole batch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import tsadar_oracle_torch as ot

    cfg, sa, batch, normed, i_norm, e_norm = _setup(B)
    names = ["Te", "ne", "Ti_1", "lam", "amp1", "Va"]  # ravel order of the active leaves
    val, g, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    want = np.concatenate([[val]] + [g[k] for k in names])
    assert r0.shape == want.shape == (1 + 6 * B,)
    np.testing.assert_allclose(r0, want, rtol=1e-12, atol=1e-15)


def test_shard_bounds_and_single_rank_passthrough():
    from tsadar_amd import distributed as D

    assert D.shard_bounds(32768, 8, 3) == (12288, 16384)
    with pytest.raises(ValueError):
        D.shard_bounds(10, 4, 0)
    t = torch.arange(3, dtype=torch.float64)
    g = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    tt, gg = D.allreduce_loss_grad(t, g, 1, 0)
    assert torch.equal(tt, t) and torch.equal(gg, g.reshape(-1))
