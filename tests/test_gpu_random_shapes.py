"""GPU parity on seeded RANDOM shapes: horizon, transition_dim, width (dim), dim_mults and batch drawn per seed,
per-candidate timesteps, TemporalUnet forward and ValueFunction value + input gradient against the numpy oracle
(which is pinned to the reference on the golden configurations).  Shapes the reference itself cannot run -- a
ValueFunction whose head Linear does not fit, temporal.py:191,205 -- must be rejected loudly by the library, never
computed some other way.  Tolerance 1e-4 relative as in test_gpu_nets.py."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import trebi_oracle as O  # noqa: E402
from _common import rel_err  # noqa: E402
from trebi_amd import spec, synth  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4
MULTS = [(1, 2), (1, 2, 4), (1, 4, 8), (1, 2, 4, 8), (1, 1, 2, 4, 8)]


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    mults = MULTS[int(rng.integers(len(MULTS)))]
    n = len(mults)
    H = int(2 ** rng.integers(max(n - 1, 2), 8))                 # the UNet halves the horizon n-1 times: H >= 2^(n-1)
    D = int(rng.integers(3, 31))
    dim = int(rng.choice([32, 64]))
    B = int(rng.integers(1, 41))
    return rng, H, D, dim, mults, B


@pytest.mark.parametrize('seed', range(10))
def test_unet_random_shape(seed):
    from trebi_amd.nets import TemporalUnet
    rng, H, D, dim, mults, B = draw(seed)
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, D, dim, mults), seed)
    net = TemporalUnet(H, D, 0, dim=dim, dim_mults=mults)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.set_max_timesteps(20)
    net = net.cuda()
    x = rng.standard_normal((B, H, D)).astype(np.float32)
    t = rng.integers(0, 20, size=B)
    out = net(torch.from_numpy(x).cuda(), None, torch.from_numpy(t).cuda()).cpu().numpy()
    ref = O.TemporalUnet(sd, H, D, dim, mults).forward(x, None, t)
    assert rel_err(out, ref) < TOL, (H, D, dim, mults, B)


@pytest.mark.parametrize('seed', range(10))
def test_value_random_shape(seed):
    from trebi_amd.nets import ValueFunction
    rng, H, D, dim, mults, B = draw(seed)
    H = min(H, 2 ** (len(mults) + 1))                             # the reference's head fits only then
    sd = synth.synth_state_dict(spec.value_param_shapes(H, D, dim, mults), seed)
    net = ValueFunction(H, D, 0, dim=dim, dim_mults=mults)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.set_max_timesteps(20)
    net = net.cuda()
    x = rng.standard_normal((B, H, D)).astype(np.float32)
    t = rng.integers(0, 20, size=B)
    y, g = net.value_and_input_grad(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    yo, go = O.ValueFunction(sd, H, D, dim, mults).value_and_input_grad(x, None, t)
    assert rel_err(y.cpu().numpy(), yo) < TOL, (H, D, dim, mults, B)
    assert rel_err(g.cpu().numpy(), go) < TOL, (H, D, dim, mults, B)


def test_value_head_that_does_not_fit_is_rejected():
    """horizon 128 with four levels: the reference's forward fails at the head Linear (2x160 @ 288x128); the library
    must refuse the shape instead of computing something."""
    from trebi_amd.nets import ValueFunction
    H, D, mults = 128, 4, (1, 2, 4, 8)
    sd = synth.synth_state_dict(spec.value_param_shapes(H, D, 32, mults), 3)
    net = ValueFunction(H, D, 0, dim=32, dim_mults=mults)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.set_max_timesteps(20)
    net = net.cuda()
    x = torch.zeros(2, H, D, device='cuda')
    with pytest.raises(RuntimeError):
        net.value_and_input_grad(x, torch.zeros(2, device='cuda', dtype=torch.long))
