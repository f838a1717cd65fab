"""Edge cases of the boundary: empty and single-candidate batches, wrong shapes, timesteps outside the table."""
import numpy as np
import pytest
import torch

from _common import rel_err
from trebi_amd import spec, synth

pytestmark = pytest.mark.gpu
H, D = 32, 23


def _net(cls, shapes, seed):
    net = cls(H, D, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()})
    net.set_max_timesteps(20)
    return net.cuda()


@pytest.fixture(scope='module')
def nets():
    from trebi_amd.nets import TemporalUnet, ValueFunction
    return (_net(TemporalUnet, spec.unet_param_shapes(H, D, 32, (1, 2, 4, 8)), 5),
            _net(ValueFunction, spec.value_param_shapes(H, D, 32, (1, 2, 4, 8)), 6))


def test_empty_batch(nets):
    """What the reference's torch modules return for an empty batch: empty tensors of the right shape."""
    from diffuser.models import GaussianDiffusion
    from diffuser.models.diffusion import default_sample_fn
    unet, vf = nets
    x0 = torch.zeros(0, H, D, device='cuda')
    t0 = torch.zeros(0, dtype=torch.long, device='cuda')
    assert unet(x0, None, t0).shape == (0, H, D)
    y, g = vf.value_and_input_grad(x0, t0)
    assert y.shape == (0,) and g.shape == (0, H, D) and vf(x0, None, t0).shape == (0, 1)
    d = GaussianDiffusion(unet, H, 17, 6, n_timesteps=20, loss_type='l2', clip_denoised=False,
                          predict_epsilon=False).cuda()
    s = d.p_sample_loop((0, H, D), {0: torch.zeros(0, 17, device='cuda')}, verbose=False, return_chain=True,
                        sample_fn=default_sample_fn)
    assert s.trajectories.shape == (0, H, D) and s.values.shape == (0,) and s.chains.shape == (0, 21, H, D)


def test_wrong_shapes_are_refused(nets):
    unet, vf = nets
    t = torch.zeros(2, dtype=torch.long, device='cuda')
    for bad in (torch.zeros(2, H + 1, D), torch.zeros(2, H, D - 1), torch.zeros(2, H * D)):
        with pytest.raises(RuntimeError, match='expected x of shape'):
            unet(bad.cuda(), None, t)
        with pytest.raises(RuntimeError, match='expected x of shape'):
            vf.value_and_input_grad(bad.cuda(), t)


def test_single_candidate_and_timestep_range(nets):
    """One candidate equals the first row of a larger batch; timesteps outside [0, max_timesteps) are refused on the
    host (the reference evaluates the embedding for any t; the library's per-timestep tables end at max_timesteps,
    and the C ABI itself -- which cannot look at device memory without a sync -- documents a clamp)."""
    unet, vf = nets
    x = torch.randn(5, H, D, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    t = torch.full((5,), 7, dtype=torch.long, device='cuda')
    full = unet(x, None, t)
    assert rel_err(unet(x[:1], None, t[:1]).cpu().numpy(), full[:1].cpu().numpy()) < 1e-6
    for bad in (99, 20, -4):
        with pytest.raises(RuntimeError, match='timesteps must lie in'):
            unet(x, None, torch.full((5,), bad, dtype=torch.long, device='cuda'))
        with pytest.raises(RuntimeError, match='timesteps must lie in'):
            vf.value_and_input_grad(x, torch.full((5,), bad, dtype=torch.long, device='cuda'))
    assert torch.isfinite(unet(x, None, 19)).all()                  # scalar t broadcasts, upper edge is valid
