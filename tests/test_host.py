"""CPU suite: host-side logic -- module parameter trees, schedule buffers, sharding arithmetic, and that
the product path refuses to run without a GPU (no CPU fallback)."""
import os

import numpy as np
import pytest
import torch

from _common import GOLDEN
from trebi_amd import spec
from trebi_amd.parallel import shard_bounds


def test_module_state_dict_keys_and_shapes():
    from diffuser.models import TemporalUnet, ValueFunction
    for cls, fn, args in ((TemporalUnet, spec.unet_param_shapes, (32, 23)), (ValueFunction, spec.value_param_shapes, (32, 23)),
                          (TemporalUnet, spec.unet_param_shapes, (32, 23, 32, (1, 4, 8)))):
        shapes = fn(*args)
        net = cls(args[0], args[1], 0, **(dict(dim=args[2], dim_mults=args[3]) if len(args) > 2 else {}))
        sd = net.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), k
    # the documented examples of SURVEY.md section 8b
    sd = TemporalUnet(32, 23, 0).state_dict()
    assert tuple(sd['downs.0.0.blocks.0.block.0.weight'].shape) == (32, 23, 5)
    assert tuple(sd['downs.0.3.conv.weight'].shape) == (32, 32, 3)
    assert tuple(sd['ups.0.3.conv.weight'].shape) == (128, 128, 4)
    assert tuple(sd['final_conv.1.weight'].shape) == (23, 32, 1)
    assert tuple(ValueFunction(32, 23, 0).state_dict()['final_block.0.weight'].shape) == (32, 96)


def test_value_plan_flags_the_reference_shape_bug():
    """(1,4,8) value nets are only self-consistent at horizon 4."""
    assert spec.value_plan(32, 23, 32, (1, 2, 4, 8))['consistent']
    assert spec.value_plan(4, 23, 32, (1, 4, 8))['consistent']
    assert not spec.value_plan(32, 23, 32, (1, 4, 8))['consistent']


def test_diffusion_buffers_match_reference_schedule():
    from diffuser.models import TemporalUnet, GaussianDiffusion
    d = GaussianDiffusion(TemporalUnet(32, 23, 0), 32, 17, 6, n_timesteps=20, loss_type='l2', clip_denoised=False,
                          predict_epsilon=False)
    ref = np.load(os.path.join(GOLDEN, 'schedule_T20.npz'))
    for k in ref.files:
        np.testing.assert_array_equal(getattr(d, k).numpy(), ref[k], err_msg=k)      # bit-exact tables
    keys = [k for k in d.state_dict() if not k.startswith('model.')]
    assert 'loss_fn.weights' in keys and 'posterior_mean_coef2' in keys
    assert d.model.max_timesteps == 20


def test_no_cpu_fallback():
    from diffuser.models import TemporalUnet
    net = TemporalUnet(32, 23, 0)
    with pytest.raises(RuntimeError, match='no CPU path'):
        net(torch.zeros(2, 32, 23), None, torch.zeros(2, dtype=torch.long))


def test_shard_bounds_partition():
    for n in (1, 7, 64, 65, 4096, 65536):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_policy_host_pieces():
    """GuidedPolicy's dataset-free pieces: the normaliser arithmetic (normalization.py:145-149) and the refusal
    of preprocess names (the reference evals them inside its datasets package)."""
    from diffuser.sampling import GuidedPolicy, Trajectories
    from trebi_amd.policy import GaussianNormalizer, _compose
    n = GaussianNormalizer({'observations': ([1.0, 2.0], [2.0, 4.0])})
    x = np.array([[3.0, 10.0]], np.float32)
    np.testing.assert_allclose(n.normalize(x, 'observations'), [[1.0, 2.0]])
    np.testing.assert_allclose(n.unnormalize(n.normalize(x, 'observations'), 'observations'), x)
    assert _compose([lambda v: v + 1, lambda v: v * 2])(1) == 4
    with pytest.raises(RuntimeError, match='callables'):
        _compose(['arctanh_actions'])
    assert Trajectories._fields == ('actions', 'observations', 'rewards', 'terminals', 'values', 'costs')
    assert GuidedPolicy.__call__.__code__.co_varnames[:6] == ('self', 'conditions', 'batch_size', 'verbose',
                                                              'cost_threshold', 'plan_horizon')


def test_reference_checkpoint_format_loads(tmp_path):
    """A file in the Trainer's format (training.py:140-150: {'step','model','ema'} of state_dicts with the
    reference's names) loads with strict=True; a foreign file is refused."""
    from diffuser.models import TemporalUnet, GaussianDiffusion
    from trebi_amd import checkpoint
    mk = lambda: GaussianDiffusion(TemporalUnet(32, 23, 0), 32, 17, 6, n_timesteps=20, loss_type='l2',
                                   clip_denoised=False, predict_epsilon=False)
    src, dst = mk(), mk()
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.5)
    path = os.path.join(tmp_path, 'state_7.pt')
    torch.save({'step': 7, 'model': src.state_dict(), 'ema': src.state_dict()}, path)
    assert checkpoint.load_into(dst, path) == 7
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    bad = os.path.join(tmp_path, 'other.pt')
    torch.save({'weights': 1}, bad)
    with pytest.raises(RuntimeError, match='not a reference trainer checkpoint'):
        checkpoint.load_into(dst, bad)


def test_loss_type_is_honoured_like_the_reference_losses_table(tmp_path):
    """helpers.py:213-218: 'l1'/'l2' carry a `weights` buffer, 'value_l1'/'value_l2' (what the reference's value and
    cost models use, config/locomotion.py:94,137) carry none, anything else is a KeyError.  A reference value
    checkpoint therefore has no `loss_fn.weights` key and must load with strict=True."""
    from diffuser.models import ValueFunction, ValueDiffusion, TemporalUnet, GaussianDiffusion
    from trebi_amd import checkpoint
    kw = dict(n_timesteps=20, clip_denoised=False, predict_epsilon=False)
    v = ValueDiffusion(ValueFunction(32, 23, 0), 32, 17, 6, loss_type='value_l2', **kw)
    assert 'loss_fn.weights' not in v.state_dict()
    d = GaussianDiffusion(TemporalUnet(32, 23, 0), 32, 17, 6, loss_type='l1', **kw)
    assert 'loss_fn.weights' in d.state_dict() and d.loss_fn.kind == 'l1'
    with pytest.raises(KeyError):
        GaussianDiffusion(TemporalUnet(32, 23, 0), 32, 17, 6, loss_type='huber', **kw)
    # the reference ValueDiffusion key set (no loss_fn.weights) loads strictly through the checkpoint reader
    src = ValueDiffusion(ValueFunction(32, 23, 0), 32, 17, 6, loss_type='value_l2', **kw)
    sd = src.state_dict()
    assert not any(k.startswith('loss_fn') for k in sd)
    path = os.path.join(tmp_path, 'state_3.pt')
    torch.save({'step': 3, 'model': sd, 'ema': sd}, path)
    assert checkpoint.load_into(v, path) == 3
    with pytest.raises(RuntimeError):                 # an 'l2' model refuses it: its weights buffer is missing
        checkpoint.load_into(ValueDiffusion(ValueFunction(32, 23, 0), 32, 17, 6, loss_type='l2', **kw), path)
