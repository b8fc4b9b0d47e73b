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


    def download(self, t):
        return t.numpy().copy()


def _oracle_evaluate_packed(self, ts_params, batch, act, B_global, b_offset, want_spectra=False):
    """Replacement for LossFunction._evaluate_packed (which calls tsff_loss_grad_packed on the GPU): oracle masked sums
    and the autodiff gradient of the weighted total w.r.t. this shard's normalised parameters, packed like the kernel
    packs them -- [3 | P x B_global], this rank's columns filled, zeros elsewhere."""
    import util
    from oracle import tsadar_oracle_torch as ot
    from tsadar_amd import distributed as D
    from tsadar_amd.engine import Engine

    X = ts_params.to_matrix()
    B = X.shape[0]
    eng = _OracleEngine(self.cfg)
    w = Engine.loss_weights(eng, B_global, self.i_norm, self.e_norm, self.cfg["data"]["ion_loss_scale"])
    if B == 0:
        return eng, w, torch.zeros(3 + len(act) * B_global, dtype=torch.float64), None, None
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
    packed = D.pack_local(S.detach(), G[:, act].t().contiguous(), B_global, b_offset)
    return eng, w, packed, E.detach(), I.detach()


def _rank_main(rank, world, port, B, out, real_init=False):
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
    LossFunction._evaluate_packed = _oracle_evaluate_packed
    if real_init:
        # the reference pattern LossFunction(cfg, sa, batch) with the rank's LOCAL shard: the normalisers must come out global
        lf = LossFunction(cfg, util.sa_fit(max(hi - lo, 1)), local, distributed=True)
        assert lf.i_norm == i_norm and lf.e_norm == e_norm, (rank, lf.i_norm, i_norm, lf.e_norm, e_norm)
    else:
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
    # the all-gather alternative returns the same buffer as the all-reduce (equal shards)
    t = torch.arange(3, dtype=torch.float64) + rank
    g = torch.arange(10, dtype=torch.float64).reshape(2, 5) * (rank + 1)
    assert torch.equal(D.allreduce_loss_grad(t, g, world, rank), D.allgather_loss_grad(t, g, world, rank))
    dist.destroy_process_group()


def _reference(B):
    """single-process oracle value and gradient of the whole batch, in the ravel order of the active leaves"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import tsadar_oracle_torch as ot

    cfg, sa, batch, normed, i_norm, e_norm = _setup(B)
    names = ["Te", "ne", "Ti_1", "lam", "amp1", "Va"]
    val, g, _, _ = ot.value_and_grad(cfg, sa, normed, batch, i_norm, e_norm, names)
    return np.concatenate([[val]] + [g[k] for k in names])


@pytest.mark.timeout(600)
def test_two_rank_sharded_loss_and_gradient_equal_single_rank(tmp_path):
    B, world = 4, 2
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "flat_0.npy")
    r1 = np.load(tmp_path / "flat_1.npy")
    np.testing.assert_array_equal(r0, r1)  # every rank holds the full loss and gradient
    want = _reference(B)
    assert r0.shape == want.shape == (1 + 6 * B,)
    np.testing.assert_allclose(r0, want, rtol=1e-12, atol=1e-15)


@pytest.mark.timeout(900)
def test_three_ranks_uneven_batch_and_local_dummy_batch(tmp_path):
    """B = 10 over 3 ranks (shards of 4, 4 and 2 lineouts: the batch does not divide) with LossFunction constructed from each
    rank's LOCAL shard, as the reference pattern does: the loss normalisers are max-reduced over the ranks, the 1/N uses
    the true global count, and every rank ends with the single-rank loss and gradient."""
    B, world = 10, 3
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, B, str(tmp_path), True), nprocs=world, join=True)
    r = [np.load(tmp_path / f"flat_{k}.npy") for k in range(world)]
    np.testing.assert_array_equal(r[0], r[1])
    np.testing.assert_array_equal(r[0], r[2])
    want = _reference(B)
    assert r[0].shape == want.shape == (1 + 6 * B,)
    np.testing.assert_allclose(r[0], want, rtol=1e-12, atol=1e-15)


def test_shard_bounds_and_single_rank_passthrough():
    from tsadar_amd import distributed as D

    assert D.shard_bounds(32768, 8, 3) == (12288, 16384)
    assert [D.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [D.shard_bounds(2, 3, r) for r in range(3)] == [(0, 1), (1, 2), (2, 2)]   # more ranks than lineouts: an empty shard
    t = torch.arange(3, dtype=torch.float64)
    g = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    buf = D.allreduce_loss_grad(t, g, 1, 0)
    assert torch.equal(buf[:3], t) and torch.equal(buf[3:], g.reshape(-1))
    p = D.pack_local(t, g, 10, 5)
    assert p.numel() == 33 and torch.equal(p[3:].view(3, 10)[:, 5:9], g) and float(p[3:].view(3, 10)[:, :5].abs().sum()) == 0.0


def test_batch_cache_is_keyed_by_the_arrays_not_by_a_recyclable_id():
    """One LossFunction, many freshly built batch dicts (loops.py:133-146): the device-resident copy must follow the batch
    even when CPython hands a freed dict's id to the next one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tsadar_amd.loss_function import LossFunction

    class FakeEngine:
        def __init__(self):
            self.made = 0

        def _vec(self, a, B):
            self.made += 1
            return np.asarray(a)

        def _mat(self, a, B):
            return None if a is None else np.asarray(a)

    lf = LossFunction.__new__(LossFunction)
    lf._dev_batch_src = lf._dev_batch = None
    eng = FakeEngine()

    def make(v):
        return dict(e_amps=np.full(2, v), i_amps=np.full(2, v), e_data=np.full((2, 1024), v), i_data=np.full((2, 1024), v))

    seen = []
    for v in (1.0, 2.0, 3.0):   # temporaries: each dict is garbage before the next is built
        seen.append(float(lf._device_batch(eng, make(v), 2)["e_data"][0, 0]))
    assert seen == [1.0, 2.0, 3.0] and eng.made == 6
    b = make(4.0)
    lf._device_batch(eng, b, 2)
    lf._device_batch(eng, b, 2)                 # same arrays: converted once
    assert eng.made == 8
    lf._device_batch(eng, dict(b), 2)           # another dict around the same arrays: still the same data
    assert eng.made == 8
    b["e_data"][:] = 5.0                        # in-place edits need an explicit invalidation
    lf.invalidate_batch()
    assert float(lf._device_batch(eng, b, 2)["e_data"][0, 0]) == 5.0 and eng.made == 10
