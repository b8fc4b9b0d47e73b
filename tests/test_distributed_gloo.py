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


@pytest.mark.timeout(1200)
def test_eight_ranks_uneven_batch_with_an_empty_shard(tmp_path):
    """The world size of BASELINE configs[4] (8 ranks) on CPU: B = 20 over 8 gloo ranks -> shards of 3, 3, 3, 3, 3, 3, 2 and 0
    lineouts (one rank only takes part in the collectives), LossFunction built from each rank's LOCAL shard.  Every rank ends
    with the single-rank loss and gradient to 1e-12."""
    B, world = 20, 8
    port = _free_port()
    mp.spawn(_rank_main, args=(world, port, B, str(tmp_path), True), nprocs=world, join=True)
    r = [np.load(tmp_path / f"flat_{k}.npy") for k in range(world)]
    for k in range(1, world):
        np.testing.assert_array_equal(r[0], r[k])
    want = _reference(B)
    assert r[0].shape == want.shape == (1 + 6 * B,)
    np.testing.assert_allclose(r[0], want, rtol=1e-12, atol=1e-15)


def _oracle_evaluate_free_form(self, ts_params, batch, want_spectra=False, B_global=None):
    """Replacement for LossFunction._evaluate on the free-form f_e branch (tsff_loss_grad_fe on the GPU): oracle masked sums,
    autodiff gradient of the weighted total w.r.t. this shard's normalised parameters and w.r.t. its f_e tables."""
    import util
    from oracle import tsadar_oracle_torch as ot
    from tsadar_amd import distribution as Dist
    from tsadar_amd.engine import Engine

    X = ts_params.to_matrix()
    B = X.shape[0]
    eng = _OracleEngine(self.cfg)
    eng.nvx = ts_params.fval.shape[1]
    w = Engine.loss_weights(eng, B if B_global is None else B_global, self.i_norm, self.e_norm, self.cfg["data"]["ion_loss_scale"])
    if B == 0:
        self._gfe = torch.zeros((0, eng.nvx), dtype=torch.float64)
        return eng, w, torch.zeros(3, dtype=torch.float64), torch.zeros((0, X.shape[1]), dtype=torch.float64), None, None
    names = ["Te", "ne", "m", "Ti_1", "Z_1", "A_1", "fract_1", "lam", "amp1", "amp2", "amp3", "ne_gradient", "Te_gradient", "ud", "Va"]
    normed = {k: torch.tensor(X[:, util.slot_of(k)], dtype=torch.float64, requires_grad=True) for k in names}
    fe = torch.tensor(Dist.arbitrary_1v(ts_params.fval), dtype=torch.float64, requires_grad=True)
    sa = dict(sa=util.P9["sa"], weights=util.P9["weights"] * np.ones([B, 10]))
    S, N, E, I = ot.masked_sums(self.cfg_oracle, sa, normed, batch, fe_batch=fe)
    total = (S * torch.as_tensor(w)).sum()
    grads = torch.autograd.grad(total, [normed[k] for k in names] + [fe], allow_unused=True)
    G = torch.zeros((B, X.shape[1]), dtype=torch.float64)
    for k, g in zip(names, grads[:-1]):
        if g is not None and ts_params.slots.active[util.slot_of(k)]:
            G[:, util.slot_of(k)] = g
    self._gfe = grads[-1]
    return eng, w, S.detach(), G, E.detach(), I.detach()


def _free_form_decks(nvx):
    import decks

    cfg = decks.deck_fit(nvx=nvx, active=("Te", "ne", "amp1", "lam"))
    cfg["parameters"]["electron"]["fe"] = {"active": True, "type": "arbitrary", "dim": 1, "nvx": nvx, "params": {"init_m": 2.4}}
    return cfg, decks.deck_fit(nvx=nvx, active=("Te", "ne", "amp1", "lam")), decks.deck_fit(nvx=nvx)


def _free_form_inputs(B, nvx):
    import util
    from tsadar_amd import ThomsonParams

    cfg, cfg_o, cfg_data = _free_form_decks(nvx)
    batch = util.synthetic_batch(cfg_data, util.sa_fit(B), B, seed=81)
    tp = ThomsonParams(cfg["parameters"], B, batch=True, activate=True)
    rng = np.random.default_rng(4)
    tp.fval = tp.fval * (1 + 0.02 * rng.normal(size=tp.fval.shape))
    tp.X[:, 0] += rng.normal(size=B) * 0.1
    return cfg, cfg_o, batch, tp


def _rank_main_free_form(rank, world, port, B, nvx, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import util
    from tsadar_amd import ThomsonParams, distributed as D, tree
    from tsadar_amd.loss_function import LossFunction

    D.init_from_env(backend="gloo")
    cfg, cfg_o, batch, tp = _free_form_inputs(B, nvx)
    lo, hi = D.shard_bounds(B, world, rank)
    local = {k: np.asarray(v)[lo:hi] for k, v in batch.items()}
    LossFunction._evaluate = _oracle_evaluate_free_form
    lf = LossFunction(cfg, util.sa_fit(max(hi - lo, 1)), local, distributed=True)   # the LOCAL shard (empty on the last rank)
    lf.cfg_oracle = cfg_o
    spec = tree.get_filter_spec(cfg["parameters"], tp)
    diff, _ = tree.partition(tp, spec)
    x0, lf.unravel_weights = tree.ravel_pytree(diff)
    tp_local = ThomsonParams(cfg["parameters"], hi - lo, batch=True, activate=True)
    tp_local.X[:] = tp.X[lo:hi]
    tp_local.fval = tp.fval[lo:hi].copy()
    _, static_l = tree.partition(tp_local, tree.get_filter_spec(cfg["parameters"], tp_local))
    value, flat = lf.vg_loss(x0, static_l, local)
    np.save(os.path.join(out, f"ff_{rank}.npy"), np.concatenate([[value], flat]))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_uneven_shards_on_the_free_form_branch(tmp_path):
    """vg_loss with a free-form f_e (the branch that packs [P + nvx] rows with torch) over 4 ranks and B = 5: shards of 2, 2, 1
    and 0 lineouts.  The 1/N of the loss must be the global count on every rank (it used to be B_local * world)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import tsadar_oracle as orc
    from oracle import tsadar_oracle_torch as ot
    from tsadar_amd import distribution as Dist
    import util

    B, world, nvx = 5, 4, 64
    port = _free_port()
    mp.spawn(_rank_main_free_form, args=(world, port, B, nvx, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"ff_{k}.npy") for k in range(world)]
    for k in range(1, world):
        np.testing.assert_array_equal(r[0], r[k])
    cfg, cfg_o, batch, tp = _free_form_inputs(B, nvx)
    names = ["Te", "ne", "lam", "amp1"]
    normed = {k: tp.X[:, util.slot_of(k)].copy() for k in orc.init_normed_params(cfg_o["parameters"], B, True)}
    i_norm, e_norm = orc.loss_norms(cfg, batch)
    vo, ref, ref_fe, _, _ = ot.value_and_grad_fe(cfg_o, util.sa_fit(B), normed, batch, i_norm, e_norm, names, Dist.arbitrary_1v(tp.fval))
    want = np.concatenate([[vo], ref["Te"], ref["ne"], Dist.arbitrary_1v_vjp(tp.fval, ref_fe).ravel(), ref["lam"], ref["amp1"]])
    assert r[0].shape == want.shape
    np.testing.assert_allclose(r[0], want, rtol=1e-10, atol=1e-13 * np.max(np.abs(want)))


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` without a launcher around it starts two fresh rank processes itself (before the parent has
    touched torch or the GPU).  On a box without a GPU both ranks must refuse to run -- "needs a HIP device", no CPU
    fallback -- and the parent must pass the failure on as a non-zero exit code."""
    import subprocess

    if torch.cuda.is_available():
        pytest.skip("GPU box: the launch itself is exercised by the driver's scaling run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs a HIP device") >= 2, p.stderr[-3000:]
    assert "but WORLD_SIZE=1" not in p.stderr
    assert '"metric"' not in p.stdout


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
