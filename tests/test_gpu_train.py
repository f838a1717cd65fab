"""GPU parity of the TRAINING step (BASELINE configs[4]) through the C ABI
(trebi_trainer_*): loss, EVERY parameter gradient, the Adam-updated weights and the EMA copy against
 * tests/golden/train_*.npz -- produced by the reference modules under torch.autograd / torch.optim.Adam
   (tests/golden/make_train_golden.py), and
 * the numpy oracle (oracle/trebi_train_oracle.py, itself pinned to those goldens) for the full tensors.

Bars (stated): losses 1e-4 relative; a parameter's gradient within 1e-3 of that gradient's largest element
(north_star's 1e-3 for fp32), and its golden summary (sum, L2 norm, 256 sampled elements) as in
test_train_oracle_golden.py; parameters after Adam as there (the UPDATE within 2 % of lr on >= 99 % of the sampled
elements: Adam's first steps move every element by about +-lr whatever the gradient's size)."""
import os
import sys

import numpy as np
import pytest
import torch

from _common import GOLDEN
from trebi_amd import spec, synth

pytestmark = pytest.mark.gpu


def summary(v):
    v = np.asarray(v, np.float32).reshape(-1)
    idx = np.linspace(0, v.size - 1, min(v.size, 256)).astype(np.int64)
    return np.concatenate([[v.sum(dtype=np.float64), np.sqrt((v.astype(np.float64) ** 2).sum())], v[idx]]).astype(np.float64)


def _model(fx, seed=None):
    from diffuser.models import TemporalUnet, GaussianDiffusion
    obs, act, H, T = int(fx['obs']), int(fx['act']), int(fx['H']), int(fx['T'])
    mults = tuple(int(m) for m in fx['mults'])
    D = obs + act
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, D, int(fx['dim']), mults), int(fx['seed_w']) if seed is None else seed)
    unet = TemporalUnet(H, D, 0, dim=int(fx['dim']), dim_mults=mults)
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    diff = GaussianDiffusion(unet, H, obs, act, n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False,
                             action_weight=float(fx['action_weight']), loss_discount=1, loss_weights=None).cuda()
    return diff, sd


def _batch(fx, step, i):
    tag = f's{step}m{i}'
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return cu(fx[f'{tag}_x']), {0: cu(fx[f'{tag}_c'])}, cu(fx[f'{tag}_t']), cu(fx[f'{tag}_n'])


def _grads(eng):
    g = eng.grads.detach().cpu().numpy()
    out, off = {}, 0
    for k, shape in eng.unet._shapes.items():
        n = int(np.prod(shape))
        out[k] = g[off:off + n].reshape(shape)
        off += n
    return out


@pytest.fixture(scope='module', params=['train_halfcheetah', 'train_pendulum_148'])
def fx(request):
    return np.load(os.path.join(GOLDEN, request.param + '.npz'))


def test_loss_and_every_parameter_gradient(fx):
    import trebi_oracle as O
    import trebi_train_oracle as TO
    diff, sd = _model(fx)
    eng = diff.train_engine(max_batch=8)
    x, cond, t, noise = _batch(fx, 0, 0)
    loss, info = diff.p_losses(x, cond, t, noise=noise)                       # grad_scale 1: d loss / d params
    torch.cuda.synchronize()
    want_loss, want_a0 = float(fx['s0m0_loss']), float(fx['s0m0_a0'])
    assert abs(float(loss) - want_loss) <= 1e-4 * abs(want_loss)
    assert abs(float(info['a0_loss']) - want_a0) <= 1e-4 * max(abs(want_a0), 1e-3)
    got = _grads(eng)
    # (a) the golden summaries of torch.autograd's gradients
    names = [str(n) for n in fx['names']]
    assert sorted(names) == sorted(got.keys())
    for k in names:
        want = fx[f'grad0.{k}']
        s = summary(got[k])
        scale = want[1] / np.sqrt(got[k].size) + 1e-12
        assert abs(s[1] - want[1]) <= 1e-3 * want[1], k
        assert np.abs(s[2:] - want[2:]).max() <= 1e-3 * 30 * scale, k
    # (b) the full tensors from the oracle
    obs, act, H, T = int(fx['obs']), int(fx['act']), int(fx['H']), int(fx['T'])
    mults = tuple(int(m) for m in fx['mults'])
    unet = TO.TrainableUnet(sd, H, obs + act, int(fx['dim']), mults)
    lw = TO.loss_weights(H, obs + act, act, float(fx['action_weight']))
    _, _, og = TO.p_losses(unet, fx['s0m0_x'], {0: fx['s0m0_c']}, fx['s0m0_t'], fx['s0m0_n'], lw, act, O.make_schedule(T))
    worst = 0.0
    for k in names:
        ref = og[k].reshape(got[k].shape)
        err = np.abs(got[k] - ref).max() / max(np.abs(ref).max(), 1e-12)
        worst = max(worst, err)
        assert err < 1e-3, (k, err)
    print('worst relative gradient error vs oracle:', worst)


def test_accumulation_adam_and_ema_follow_the_reference(fx):
    from trebi_amd.training import Trainer
    diff, sd = _model(fx)
    lr = float(fx['lr'])
    tr = Trainer(diff, train_batch_size=8, train_lr=lr, gradient_accumulate_every=int(fx['accum']),
                 ema_decay=float(fx['ema_decay']), step_start_ema=1, update_ema_every=1)
    names = [str(n) for n in fx['names']]
    init = {k: v.copy() for k, v in sd.items()}
    for step in range(int(fx['steps'])):
        losses = tr.train_step([_batch(fx, step, i) for i in range(int(fx['accum']))])
        torch.cuda.synchronize()
        for i, l in enumerate(losses):
            want = float(fx[f's{step}m{i}_loss'])
            assert abs(float(l[0]) - want) <= 1e-4 * abs(want), (step, i)
        assert float(tr.engine.grads.abs().max()) == 0.0                      # optimizer.zero_grad()
        for which, module in (('p', tr.model.model), ('ema', tr.ema_model.model)):
            cur = {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}
            bad = tot = 0
            for k in names:
                base = summary(init[k])[2:]
                want = fx[f'{which}{step}.{k}'][2:] - base
                got = summary(cur[k])[2:] - base
                d = np.abs(got - want)
                bad += int((d > 0.02 * lr).sum())
                tot += d.size
                assert d.max() <= 2.05 * lr * (step + 1), (which, step, k)
            assert bad <= 0.01 * tot, (which, step, bad, tot)
    # the weights the kernels run on were re-packed: planning-path forward of the trained module == a fresh module
    from diffuser.models import TemporalUnet
    H, D = int(fx['H']), int(fx['obs']) + int(fx['act'])
    mults = tuple(int(m) for m in fx['mults'])
    fresh = TemporalUnet(H, D, 0, dim=int(fx['dim']), dim_mults=mults)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in tr.model.model.state_dict().items()})
    fresh = fresh.cuda()
    fresh.set_max_timesteps(int(fx['T']))
    x, _, t, _ = _batch(fx, 0, 0)
    assert torch.equal(tr.model.model(x, None, t), fresh(x, None, t))


def test_optimizer_and_ema_pin():
    """Adam and the EMA rule of the HIP step (adam_ema_kernel) over 4 optimiser steps against the reference, with a
    decay (0.9) and a learning rate (1e-3) large enough that a wrong or absent EMA update, a misplaced eps or a wrong bias
    correction is far outside the bars (tests/golden/train_ema_pin.npz; bars in _common.check_optimizer_pin)."""
    from _common import check_optimizer_pin
    from trebi_amd.training import Trainer
    fx = np.load(os.path.join(GOLDEN, 'train_ema_pin.npz'))
    diff, sd = _model(fx)
    tr = Trainer(diff, train_batch_size=8, train_lr=float(fx['lr']), gradient_accumulate_every=int(fx['accum']),
                 ema_decay=float(fx['ema_decay']), step_start_ema=1, update_ema_every=1)
    ps, es = [], []
    for step in range(int(fx['steps'])):
        losses = tr.train_step([_batch(fx, step, i) for i in range(int(fx['accum']))])
        torch.cuda.synchronize()
        for i, l in enumerate(losses):
            want = float(fx[f's{step}m{i}_loss'])
            assert abs(float(l[0]) - want) <= 1e-4 * abs(want), (step, i)
        ps.append({k: v.detach().cpu().numpy().copy() for k, v in tr.model.model.state_dict().items()})
        es.append({k: v.detach().cpu().numpy().copy() for k, v in tr.ema_model.model.state_dict().items()})
    print('worst Adam deviation / lr %.2e, worst EMA ratio deviation %.2e (%d / %d elements)' % check_optimizer_pin(fx, ps, es))


def test_trainer_refuses_time_tables_that_do_not_fit_the_lds():
    """The backward keeps per-timestep tables in LDS (gnb_kernel, time_mlp_bwd_kernel): GaussianDiffusion's default
    n_timesteps = 1000 must be refused at trebi_trainer_create with a message, not fail at the first launch."""
    from diffuser.models import TemporalUnet, GaussianDiffusion
    unet = TemporalUnet(32, 23, 0, dim=32, dim_mults=(1, 2, 4, 8))
    diff = GaussianDiffusion(unet, 32, 17, 6, n_timesteps=1000, loss_type='l2', clip_denoised=False, predict_epsilon=False).cuda()
    with pytest.raises(RuntimeError, match='per-timestep tables'):
        diff.train_engine(max_batch=8)


def test_training_step_at_horizon_4():
    """The reference's in-repo halfcheetah shape (H = 4, dim_mults (1,4,8), config/locomotion.py:295-314) through the training
    kernels: gradients of a small batch against the numpy oracle, and the 4096-sample batch through linearity (a batch of
    copies has the same loss and gradient) -- the size at which the chooser turns to position-major / W5 tiles."""
    import trebi_oracle as O
    import trebi_train_oracle as TO
    from diffuser.models import TemporalUnet, GaussianDiffusion
    H, obs, act, T, dim, mults, B = 4, 17, 6, 20, 32, (1, 4, 8), 8
    D = obs + act
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, D, dim, mults), 77)
    unet = TemporalUnet(H, D, 0, dim=dim, dim_mults=mults)
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    diff = GaussianDiffusion(unet, H, obs, act, n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False,
                             action_weight=10.0).cuda()
    eng = diff.train_engine(max_batch=4096)
    x = synth.synth_normal(6, 'h4.x', (B, H, D))
    c = synth.synth_normal(6, 'h4.c', (B, obs))
    x[:, 0, act:] = c
    n = synth.synth_normal(6, 'h4.n', (B, H, D))
    t = np.random.default_rng(4).integers(0, T, size=B)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    loss, _ = diff.p_losses(cu(x), {0: cu(c)}, cu(t), noise=cu(n))
    torch.cuda.synchronize()
    ol, _, og = TO.p_losses(TO.TrainableUnet(sd, H, D, dim, mults), x, {0: c}, t, n, TO.loss_weights(H, D, act, 10.0), act,
                            O.make_schedule(T))
    assert abs(float(loss) - ol) <= 1e-4 * abs(ol)
    got = _grads(eng)
    for k, ref in og.items():
        err = np.abs(got[k] - ref.reshape(got[k].shape)).max() / max(np.abs(ref).max(), 1e-12)
        assert err < 1e-3, (k, err)
    g1 = eng.grads.clone()
    eng.grads.zero_()
    idx = torch.arange(4096, device='cuda') % B
    lossN, _ = diff.p_losses(cu(x)[idx], {0: cu(c)[idx]}, cu(t)[idx], noise=cu(n)[idx])
    torch.cuda.synchronize()
    assert abs(float(lossN) - float(loss)) <= 1e-5 * abs(float(loss))
    off = 0
    for k, shape in eng.unet._shapes.items():
        m = int(np.prod(shape))
        a, b = g1[off:off + m], eng.grads[off:off + m]
        assert float((a - b).abs().max() / a.abs().max().clamp_min(1e-12)) < 1e-3, k
        off += m


@pytest.mark.parametrize('B', [1, 37])
def test_ragged_batches_vs_oracle(B):
    """Batch sizes that fill no tile and no K-split evenly, per-sample timesteps; full tensors vs the oracle."""
    import trebi_oracle as O
    import trebi_train_oracle as TO
    fx = np.load(os.path.join(GOLDEN, 'train_halfcheetah.npz'))
    diff, sd = _model(fx, seed=321)
    eng = diff.train_engine(max_batch=64)
    obs, act, H, T = int(fx['obs']), int(fx['act']), int(fx['H']), int(fx['T'])
    D = obs + act
    x = synth.synth_normal(5, f'rag{B}.x', (B, H, D))
    c = synth.synth_normal(5, f'rag{B}.c', (B, obs))
    x[:, 0, act:] = c
    n = synth.synth_normal(5, f'rag{B}.n', (B, H, D))
    t = np.random.default_rng(B).integers(0, T, size=B)
    loss, _ = diff.p_losses(torch.from_numpy(x).cuda(), {0: torch.from_numpy(c).cuda()}, torch.from_numpy(t).cuda(),
                            noise=torch.from_numpy(n).cuda())
    torch.cuda.synchronize()
    unet = TO.TrainableUnet(sd, H, D, int(fx['dim']), tuple(int(m) for m in fx['mults']))
    lw = TO.loss_weights(H, D, act, float(fx['action_weight']))
    ol, _, og = TO.p_losses(unet, x, {0: c}, t, n, lw, act, O.make_schedule(T))
    assert abs(float(loss) - ol) <= 1e-4 * abs(ol)
    got = _grads(eng)
    for k, ref in og.items():
        err = np.abs(got[k] - ref.reshape(got[k].shape)).max() / max(np.abs(ref).max(), 1e-12)
        assert err < 1e-3, (k, err)


@pytest.mark.parametrize('name,total', [('train_halfcheetah', 4096), ('train_pendulum_148', 1027)])
def test_full_size_batch_through_linearity(name, total):
    """BASELINE configs[4]'s batch (4096) through a size-independent property: the loss is a MEAN over the batch, so a
    batch of copies of the golden micro-batch has the golden loss and the golden gradient, whatever tiles and K-splits
    the kernels pick at that size (and a ragged last copy changes both in a known way: weights n_i / total)."""
    fx = np.load(os.path.join(GOLDEN, name + '.npz'))
    diff, _ = _model(fx)
    eng = diff.train_engine(max_batch=total)
    x, cond, t, noise = _batch(fx, 0, 0)
    Bg = x.shape[0]
    loss1, _ = diff.p_losses(x, cond, t, noise=noise)
    g1 = eng.grads.clone()
    eng.grads.zero_()
    idx = torch.arange(total, device='cuda') % Bg
    if total % Bg:                                     # keep the copies balanced: use a multiple, plus a separate ragged call
        total_even = total - total % Bg
        idx = idx[:total_even]
    lossN, _ = diff.p_losses(x[idx], {0: cond[0][idx]}, t[idx], noise=noise[idx])
    torch.cuda.synchronize()
    assert abs(float(lossN) - float(loss1)) <= 1e-5 * abs(float(loss1))
    gN = eng.grads
    scale = g1.abs().max()
    # per parameter tensor: within 1e-3 of that tensor's largest gradient element
    off = 0
    for k, shape in eng.unet._shapes.items():
        n = int(np.prod(shape))
        a, b = g1[off:off + n], gN[off:off + n]
        err = float((a - b).abs().max() / a.abs().max().clamp_min(1e-12))
        assert err < 1e-3, (k, err)
        off += n
    assert float(scale) > 0


def test_gradients_are_bit_reproducible():
    """No float atomics anywhere in the backward (wgrad slabs reduced in a fixed order, GroupNorm-backward column sums and
    per-timestep rows combined in team order): the same micro-batch gives the same gradient bits, at a batch where every
    workgroup makes several sweeps and timesteps collide."""
    fx = np.load(os.path.join(GOLDEN, 'train_halfcheetah.npz'))
    diff, _ = _model(fx)
    eng = diff.train_engine(max_batch=1500)
    x, cond, t, noise = _batch(fx, 0, 0)
    idx = torch.arange(1500, device='cuda') % x.shape[0]
    tt = (t[idx] + torch.arange(1500, device='cuda')) % int(fx['T'])
    outs, losses = [], []
    for _ in range(3):
        eng.grads.zero_()
        loss, info = diff.p_losses(x[idx], {0: cond[0][idx]}, tt, noise=noise[idx])
        torch.cuda.synchronize()
        outs.append(eng.grads.clone())
        losses.append((float(loss), float(info['a0_loss'])))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert losses[0] == losses[1] == losses[2]           # the reported loss too: per-workgroup partials summed in a fixed order
    assert float(outs[0].abs().max()) > 0


def test_tail_bucket_is_final_at_its_event():
    """trebi_trainer_set_bucket_event: what a side stream reads of grads[bucket_split:] after waiting for the event equals
    the finished gradient bit for bit, the event fires well before the call's last kernel (1500 samples: the down path's
    backward takes ~1 ms), and setting the event changes no gradient."""
    fx = np.load(os.path.join(GOLDEN, 'train_halfcheetah.npz'))
    diff, _ = _model(fx)
    eng = diff.train_engine(max_batch=1500)
    split = eng.bucket_split
    names = eng.names
    assert 0 < split < eng.grads.numel()
    first_tail = [n for n, o in zip(names, np.cumsum([0] + [diff.model.state_dict()[k].numel() for k in names])[:-1])
                  if o >= split][0]
    assert first_tail.startswith('ups.0.'), first_tail
    x, cond, t, noise = _batch(fx, 0, 0)
    idx = torch.arange(1500, device='cuda') % x.shape[0]
    tt = (t[idx] + torch.arange(1500, device='cuda')) % int(fx['T'])
    eng.grads.zero_()
    diff.p_losses(x[idx], {0: cond[0][idx]}, tt, noise=noise[idx])
    torch.cuda.synchronize()
    ref = eng.grads.clone()
    ev, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev.record()
    side = torch.cuda.Stream()
    eng.grads.zero_()
    torch.cuda.synchronize()
    eng.set_bucket_event(ev)
    diff.p_losses(x[idx], {0: cond[0][idx]}, tt, noise=noise[idx])
    end.record()
    eng.set_bucket_event(None)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        early = eng.grads[split:].clone()
    torch.cuda.synchronize()
    assert torch.equal(eng.grads, ref)
    assert torch.equal(early, ref[split:])
    # the event is not just the end of the call: the down path's backward (about a third of the backward's work) runs behind it
    assert ev.elapsed_time(end) > 0.1, ev.elapsed_time(end)


def test_two_bucket_all_reduce_on_the_device(tmp_path):
    """Trainer.train_step's overlapped path (side stream, bucket event, async all-reduce of the tail slice through RCCL)
    in a one-rank group made to look like two ranks: the sums are identities, so the step must equal the plain one."""
    import torch.distributed as dist
    from trebi_amd.training import Trainer
    fx = np.load(os.path.join(GOLDEN, 'train_halfcheetah.npz'))
    x, cond, t, noise = _batch(fx, 0, 0)
    batches = [(x, cond, t, noise), (x.flip(0), {0: cond[0].flip(0)}, t.flip(0), noise.flip(0))]
    B = x.shape[0]

    def run(distributed):
        diff, _ = _model(fx)
        tr = Trainer(diff, train_batch_size=B, train_lr=1e-3, gradient_accumulate_every=2, step_start_ema=1,
                     update_ema_every=1, distributed=distributed, results_folder=str(tmp_path))
        if distributed:
            tr._world = lambda: 2
        for _ in range(2):
            tr.train_step(batches, global_batch_sizes=[B, B])
        torch.cuda.synchronize()
        return tr.engine.params.clone(), tr.engine.ema.clone(), tr

    p0, e0, _ = run(False)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29700 + os.getpid() % 200))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        p1, e1, tr = run(True)
        assert tr._side is not None and tr._bucket_event is not None      # the overlapped path ran
    finally:
        dist.destroy_process_group()
    assert torch.equal(p0, p1) and torch.equal(e0, e1)


class _TinyDataset(torch.utils.data.Dataset):
    """What SequenceDataset hands the loop (datasets/sequence.py: Batch(trajectories, conditions)), as plain tuples."""

    def __init__(self, fx, n):
        obs, act, H = int(fx['obs']), int(fx['act']), int(fx['H'])
        self.x = synth.synth_normal(11, 'tiny.x', (n, H, obs + act))
        self.act = act

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], {0: self.x[i, 0, self.act:]}


def test_trainer_loop_with_a_dataloader_and_checkpoint_round_trip(tmp_path):
    """Trainer.train over a DataLoader (training.py:104-139: ragged last batch, CPU batches, save at step 0) and
    save -> load into a second Trainer (training.py:141-166): same weights, and the kernels run on the loaded ones."""
    from trebi_amd.training import Trainer
    fx = np.load(os.path.join(GOLDEN, 'train_halfcheetah.npz'))
    diff, sd = _model(fx)
    ds = _TinyDataset(fx, 20)                                                  # batches of 8, 8, 4
    tr = Trainer(diff, ds, train_batch_size=8, train_lr=1e-3, gradient_accumulate_every=2, step_start_ema=1,
                 update_ema_every=1, save_freq=1000, log_freq=1, results_folder=str(tmp_path))
    tr.train(3)
    torch.cuda.synchronize()
    assert tr.step == 3 and tr.adam_steps == 3
    assert os.path.exists(tmp_path / 'state_0.pt')                              # step 0 % save_freq == 0
    moved = max(float((v.detach().cpu() - torch.from_numpy(sd[k])).abs().max()) for k, v in tr.model.model.state_dict().items())
    assert 1e-4 < moved < 4e-3                                                  # three Adam steps of about lr each
    path = tr.save(7)
    saved = torch.load(path, map_location='cpu', weights_only=True)
    assert set(saved) == {'step', 'model', 'ema'} and saved['step'] == 3
    assert set(saved['model']) == set(tr.model.state_dict())                    # the reference's key set

    diff2, _ = _model(fx, seed=99)                                              # other weights, then load
    tr2 = Trainer(diff2, train_batch_size=8, train_lr=1e-3, gradient_accumulate_every=2, results_folder=str(tmp_path))
    tr2.load(7)
    assert tr2.step == 3
    for a, b in ((tr.model, tr2.model), (tr.ema_model, tr2.ema_model)):
        sa, sb = a.state_dict(), b.state_dict()
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
    x, cond, t, noise = _batch(fx, 0, 0)
    la, _ = tr.model.p_losses(x, cond, t, noise=noise)
    lb, _ = tr2.model.p_losses(x, cond, t, noise=noise)
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la))                  # training kernels see the loaded weights
    ga, gb = tr.engine.grads, tr2.engine.grads            # every sum of the step is formed in a fixed order: bit for bit
    assert torch.equal(ga, gb)
    assert torch.equal(tr.ema_model.model(x, None, t), tr2.ema_model.model(x, None, t))   # and so does the planning path
