"""CPU suite: the numpy restatement of the TRAINING step (oracle/trebi_train_oracle.py) against
vectors the reference produced under torch.autograd / torch.optim.Adam (tests/golden/make_train_golden.py).

Tolerances: losses 1e-5 relative; every parameter gradient 2e-4 of its own L2 norm scale (sum, norm and up to 256
sampled elements); parameters after Adam: the UPDATE (new - initial) within 2 % of the learning rate on >= 99 % of
the sampled elements (Adam's first steps move every element by about +-lr whatever the gradient's size, so an
element whose gradient is rounding noise may legitimately differ by up to 2 lr)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd')]
import trebi_oracle as O  # noqa: E402
import trebi_train_oracle as TO  # noqa: E402
from trebi_amd import spec, synth  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def summary(v):
    v = np.asarray(v, np.float32).reshape(-1)
    idx = np.linspace(0, v.size - 1, min(v.size, 256)).astype(np.int64)
    return np.concatenate([[v.sum(dtype=np.float64), np.sqrt((v.astype(np.float64) ** 2).sum())], v[idx]]).astype(np.float64)


@pytest.fixture(scope='module', params=['train_halfcheetah', 'train_pendulum_148'])
def replay(request):
    fx = np.load(os.path.join(GOLDEN, request.param + '.npz'))
    obs, act, H, T = int(fx['obs']), int(fx['act']), int(fx['H']), int(fx['T'])
    mults = tuple(int(m) for m in fx['mults'])
    D = obs + act
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, D, int(fx['dim']), mults), int(fx['seed_w']))
    init = {k: v.copy() for k, v in sd.items()}
    ema = {k: v.copy() for k, v in sd.items()}
    opt = TO.Adam(lr=float(fx['lr']))
    rec = dict(fx=fx, init=init, losses=[], params=[], emas=[], grad0=None)
    for step in range(int(fx['steps'])):
        batches = []
        for i in range(int(fx['accum'])):
            tag = f's{step}m{i}'
            batches.append((fx[f'{tag}_x'], {0: fx[f'{tag}_c']}, fx[f'{tag}_t'], fx[f'{tag}_n']))
        if step == 0:                                            # gradient of micro-batch 0 alone
            unet = TO.TrainableUnet(sd, H, D, int(fx['dim']), mults)
            lw = TO.loss_weights(H, D, act, float(fx['action_weight']))
            rec['lw'] = lw
            _, _, rec['grad0'] = TO.p_losses(unet, *batches[0][:1], batches[0][1], batches[0][2], batches[0][3], lw, act,
                                             O.make_schedule(T))
        losses, _ = TO.train_step(sd, ema, opt, batches, H, obs, act, T, step, action_weight=float(fx['action_weight']),
                                  ema_decay=float(fx['ema_decay']), step_start_ema=1, update_ema_every=1,
                                  dim=int(fx['dim']), dim_mults=mults)
        rec['losses'].append(losses)
        rec['params'].append({k: v.copy() for k, v in sd.items()})
        rec['emas'].append({k: v.copy() for k, v in ema.items()})
    return rec


def test_loss_weights_and_losses(replay):
    fx = replay['fx']
    np.testing.assert_allclose(replay['lw'], fx['loss_weights'], rtol=1e-6)
    for step, losses in enumerate(replay['losses']):
        for i, (loss, a0) in enumerate(losses):
            assert abs(loss - float(fx[f's{step}m{i}_loss'])) <= 1e-5 * abs(loss), (step, i)
            assert abs(a0 - float(fx[f's{step}m{i}_a0'])) <= 1e-5 * max(abs(a0), 1e-3), (step, i)


def test_every_parameter_gradient(replay):
    fx = replay['fx']
    names = [str(n) for n in fx['names']]
    assert sorted(names) == sorted(replay['grad0'].keys())        # one gradient per state_dict parameter
    worst = 0.0
    for k in names:
        want = fx[f'grad0.{k}']
        got = summary(replay['grad0'][k])
        n = replay['grad0'][k].size
        scale = want[1] / np.sqrt(n) + 1e-12                       # RMS element
        assert abs(got[1] - want[1]) <= 2e-4 * want[1], k
        assert abs(got[0] - want[0]) <= 2e-4 * want[1] * np.sqrt(n), k
        err = np.abs(got[2:] - want[2:]).max() / (scale * 30)
        worst = max(worst, err)
        assert np.abs(got[2:] - want[2:]).max() <= 2e-4 * 30 * scale, (k, err)
    print('worst sampled gradient error / (30 x rms):', worst)


@pytest.mark.parametrize('which', ['p', 'ema'])
def test_adam_and_ema_updates(replay, which):
    fx = replay['fx']
    lr = float(fx['lr'])
    names = [str(n) for n in fx['names']]
    for step in range(int(fx['steps'])):
        bad = tot = 0
        for k in names:
            base = summary(replay['init'][k])[2:]
            want = fx[f'{which}{step}.{k}'][2:] - base
            got = summary((replay['params'] if which == 'p' else replay['emas'])[step][k])[2:] - base
            d = np.abs(got - want)
            bad += int((d > 0.02 * lr).sum())
            tot += d.size
            assert d.max() <= 2.05 * lr * (step + 1), k            # never worse than a flipped sign
        assert bad <= 0.01 * tot, (which, step, bad, tot)


def test_optimizer_and_ema_pin():
    """The oracle's Adam and EMA against the reference over 4 steps with a decay and a learning rate large enough to see
    (tests/golden/train_ema_pin.npz; bars in _common.check_optimizer_pin)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _common import check_optimizer_pin
    fx = np.load(os.path.join(GOLDEN, 'train_ema_pin.npz'))
    obs, act, H, T = int(fx['obs']), int(fx['act']), int(fx['H']), int(fx['T'])
    mults = tuple(int(m) for m in fx['mults'])
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, obs + act, int(fx['dim']), mults), int(fx['seed_w']))
    ema = {k: v.copy() for k, v in sd.items()}
    opt = TO.Adam(lr=float(fx['lr']))
    ps, es = [], []
    for step in range(int(fx['steps'])):
        batches = [(fx[f's{step}m{i}_x'], {0: fx[f's{step}m{i}_c']}, fx[f's{step}m{i}_t'], fx[f's{step}m{i}_n'])
                   for i in range(int(fx['accum']))]
        losses, _ = TO.train_step(sd, ema, opt, batches, H, obs, act, T, step, action_weight=float(fx['action_weight']),
                                  ema_decay=float(fx['ema_decay']), step_start_ema=1, update_ema_every=1,
                                  dim=int(fx['dim']), dim_mults=mults)
        for i, (loss, _a0) in enumerate(losses):
            assert abs(loss - float(fx[f's{step}m{i}_loss'])) <= 2e-5 * abs(loss), (step, i)
        ps.append({k: v.copy() for k, v in sd.items()})
        es.append({k: v.copy() for k, v in ema.items()})
    print('worst Adam deviation / lr %.2e, worst EMA ratio deviation %.2e (%d / %d elements)' % check_optimizer_pin(fx, ps, es))
