"""CPU suite: the N > 1 path.  Two gloo ranks run ShardedPlanner.plan with a deterministic stand-in for
the per-rank HIP loop (candidates are independent, so the stand-in is any per-candidate function of the
injected noise) and the oracle's selection rule; the result must equal the single-process run on the
whole batch: same global values/costs, same winning index, same winning trajectory on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _fake_local(shape, cond, noise=None, sample_fn=None, cost_threshold=None, **kw):
    """Per-candidate 'sampler': x = mean over steps of the noise + observation; value/cost are fixed
    nonlinear functionals of x.  Row b depends on noise[:, b] only -- exactly the property of the real loop."""
    B, H, D = shape
    x = noise.mean(0) + 0.1
    x[:, 0, -cond[0].shape[-1]:] = cond[0]
    v = torch.tanh(x).sum((1, 2)) / (H * D)
    c = (x ** 2).mean((1, 2))
    return x, v, c


def _rule(v, c, thr):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
    import trebi_oracle as O
    return torch.from_numpy(O.feasibility_rule(v.numpy(), c.numpy(), thr.numpy()))


def _argsort(v):
    return torch.from_numpy(np.argsort(-v.numpy(), kind='stable'))


class _Diff:
    transition_dim = 6


def _plan(n, thr):
    from trebi_amd.parallel import ShardedPlanner
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(5, n, 8, 6, generator=g)
    obs = torch.randn(3, generator=g)
    planner = ShardedPlanner(_Diff(), local_fn=_fake_local, rule_fn=_rule, argsort_fn=_argsort)
    return planner.plan(obs, n, 8, None, cost_threshold=thr, noise=noise)


def _worker(rank, world, port, n, thr, q):
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), 'safe-offline-rl-with-diffusion-model_amd')]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = _plan(n, thr)
    q.put((rank, p.best_index, p.best_trajectory.numpy(), p.values.numpy(), p.costs.numpy(), p.order.numpy(),
           p.local_slice))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n,thr,world', [(13, 1.1, 2), (64, 0.9, 2), (64, -5.0, 2), (4096, 0.9, 3), (7, 1.1, 3)])
def test_ranks_equal_one(n, thr, world):
    """2 ranks, and the ragged 3-rank split of configs[2]'s 4096 candidates (1366 / 1365 / 1365)."""
    ref = _plan(n, thr)                      # world size 1, whole batch
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, thr, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0][6][0] == 0 and outs[-1][6][1] == n
    assert all(a[6][1] == b[6][0] for a, b in zip(outs[:-1], outs[1:]))
    for rank, best, traj, v, c, order, _ in outs:
        assert best == ref.best_index
        np.testing.assert_allclose(traj, ref.best_trajectory.numpy(), rtol=1e-6, atol=1e-7)
        if n <= 64 and world == 2:
            np.testing.assert_array_equal(v, ref.values.numpy())
            np.testing.assert_array_equal(c, ref.costs.numpy())
        else:       # the torch CPU stand-in itself rounds differently at other shard sizes (1 ulp): not the exchange
            np.testing.assert_allclose(v, ref.values.numpy(), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(c, ref.costs.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(order, ref.order.numpy())


# ---------------------------------------------------------------------------------------------------------
# data-parallel TRAINING step: every rank accumulates its shard's gradient, ONE all-reduce of
# the flat gradient, then the same optimiser step everywhere.  The per-rank gradient computation is replaced by the
# numpy training oracle (test infrastructure) -- what is under test is trebi_amd.training.Trainer's scaling
# (B_local / B_global per micro-batch, 1 / accumulation steps), the collective and the step bookkeeping.
# ---------------------------------------------------------------------------------------------------------
H_, OBS_, ACT_, T_, DIM_, MULTS_ = 8, 3, 1, 20, 32, (1, 2)


class _OracleEngine:
    bucket_split = 0                     # one collective; _TwoBucketEngine: the flat gradient leaves as two slices

    def __init__(self, diffusion, max_batch):
        sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
        import trebi_oracle as O
        import trebi_train_oracle as TO
        from trebi_amd import spec, synth
        self.O, self.TO = O, TO
        self.sd = synth.synth_state_dict(spec.unet_param_shapes(H_, OBS_ + ACT_, DIM_, MULTS_), 77)
        self.names = list(self.sd.keys())
        self.grads = torch.zeros(sum(v.size for v in self.sd.values()))
        self.lw = TO.loss_weights(H_, OBS_ + ACT_, ACT_, 10.0)
        self.sched = O.make_schedule(T_)
        self.steps = []

    def accumulate(self, x, cond, t, noise, grad_scale=1.0):
        unet = self.TO.TrainableUnet(self.sd, H_, OBS_ + ACT_, DIM_, MULTS_)
        loss, a0, g = self.TO.p_losses(unet, x.numpy(), {0: cond[0].numpy()}, t.numpy(), noise.numpy(), self.lw, ACT_,
                                       self.sched)
        flat = np.concatenate([g[k].reshape(-1) for k in self.names])
        self.grads += grad_scale * torch.from_numpy(flat)
        return torch.tensor([loss, a0])

    def step(self, adam_step, lr, betas, eps, ema_mode, ema_decay):
        self.steps.append((adam_step, ema_mode, self.grads.clone()))
        self.grads.zero_()


class _TwoBucketEngine(_OracleEngine):
    bucket_split = 12345                 # any interior offset: Trainer must reduce [split:] and [:split], nothing twice


def _train_batches(n):
    from trebi_amd import synth
    out = []
    for i in range(2):
        x = synth.synth_normal(3, f'dp.x{i}', (n, H_, OBS_ + ACT_))
        c = synth.synth_normal(3, f'dp.c{i}', (n, OBS_))
        x[:, 0, ACT_:] = c
        t = np.random.default_rng(i).integers(0, T_, size=n)
        nz = synth.synth_normal(3, f'dp.n{i}', (n, H_, OBS_ + ACT_))
        out.append(tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in (x, c, t, nz)))
    return out


class _Model:
    n_timesteps = T_


def _train(n, world, rank, buckets=1):
    from trebi_amd.parallel import shard_bounds
    from trebi_amd.training import Trainer
    tr = Trainer(_Model(), train_batch_size=n, train_lr=2e-4, gradient_accumulate_every=2, step_start_ema=1,
                 update_ema_every=1, engine_factory=_TwoBucketEngine if buckets == 2 else _OracleEngine,
                 distributed=world > 1)
    lo, hi = shard_bounds(n, world, rank)
    batches = [(x[lo:hi], {0: c[lo:hi]}, t[lo:hi], nz[lo:hi]) for x, c, t, nz in _train_batches(n)]
    losses = tr.train_step(batches, global_batch_sizes=[n, n])
    return tr.engine.steps[0], [l.numpy() for l in losses], tr.step


def _train_worker(rank, world, port, n, q, buckets):
    sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), 'safe-offline-rl-with-diffusion-model_amd')]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    (adam_step, ema_mode, g), losses, step = _train(n, world, rank, buckets)
    q.put((rank, adam_step, ema_mode, g.numpy(), step))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n,buckets', [(6, 1), (7, 1), (7, 2)])
def test_data_parallel_gradients_equal_single_rank(n, buckets):
    """All-reduced gradients of 2 ranks (even and ragged shards) == the single-rank gradient of the whole batch,
    with the flat gradient reduced in one piece and as the two buckets the overlapped path uses."""
    (adam_step, ema_mode, ref), _, _ = _train(n, 1, 0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() + n + 7 * buckets) % 2000
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, n, q, buckets)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scale = np.abs(ref.numpy()).max()
    for rank, a, e, g, step in outs:
        assert (a, e, step) == (adam_step, ema_mode, 1)
        assert np.abs(g - ref.numpy()).max() <= 2e-6 * scale, rank      # float32 summation order only
    np.testing.assert_array_equal(outs[0][3], outs[1][3])               # every rank steps with identical gradients
