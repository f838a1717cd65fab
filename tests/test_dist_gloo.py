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


@pytest.mark.parametrize('n,thr', [(13, 1.1), (64, 0.9), (64, -5.0)])
def test_two_ranks_equal_one(n, thr):
    ref = _plan(n, thr)                      # world size 1, whole batch
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, thr, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0][6][1] == outs[1][6][0] and outs[0][6][0] == 0 and outs[1][6][1] == n
    for rank, best, traj, v, c, order, _ in outs:
        assert best == ref.best_index
        np.testing.assert_array_equal(traj, ref.best_trajectory.numpy())
        np.testing.assert_array_equal(v, ref.values.numpy())
        np.testing.assert_array_equal(c, ref.costs.numpy())
        np.testing.assert_array_equal(order, ref.order.numpy())
