"""GPU: the one entry point that allocates nothing, trebi_unet_forward_static (include/trebi_hip.h), called
through the C ABI and held bit for bit to trebi_unet_forward (same kernels, caller workspace) -- twice in a
row, the second call with a SMALLER batch, so that whatever the first call left in the per-GPU static scratch
array is under the second one.  Also: a batch whose workspace exceeds the array is refused with a clear error.

Collected last on purpose (test_zz_*): the round-end driver runs pytest -x."""
import ctypes as C

import numpy as np
import pytest
import torch

from _common import fixture_weights, load_fixture

pytestmark = pytest.mark.gpu


def _unet(name):
    from trebi_amd.nets import TemporalUnet
    fx = load_fixture(name)
    sd_u, _, _, um, _ = fixture_weights(fx)
    H, D, dim = int(fx['H']), int(fx['obs']) + int(fx['act']), int(fx['dim'])
    net = TemporalUnet(H, D, 0, dim=dim, dim_mults=um)
    net.load_state_dict({n: torch.from_numpy(v) for n, v in sd_u.items()}, strict=True)
    net.set_max_timesteps(20)
    return fx, net.to('cuda'), H, D


def _both(net, x, t):
    """(caller-workspace result, static-scratch result), both straight through the C ABI."""
    from trebi_amd import _lib
    lib = _lib.lib()
    B, H, D = x.shape
    h = net.net_handle(x.device)
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    nbytes = lib.trebi_unet_workspace_bytes(h, B)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    ref = torch.full((B, H, D), float('nan'), device=x.device)
    out = torch.full((B, H, D), float('nan'), device=x.device)
    _lib.check(lib.trebi_unet_forward(h, _lib.ptr(x), D, _lib.ptr(t), _lib.ptr(ref), D, B, _lib.ptr(ws), nbytes, st))
    _lib.check(lib.trebi_unet_forward_static(h, _lib.ptr(x), D, _lib.ptr(t), _lib.ptr(out), D, B, st))
    torch.cuda.synchronize()
    return ref.cpu().numpy(), out.cpu().numpy()


@pytest.mark.parametrize('name', ['halfcheetah', 'walker148'])
def test_static_scratch_forward_equals_workspace_forward(name):
    fx, net, H, D = _unet(name)
    g = torch.Generator('cuda').manual_seed(11)
    for B in (7, 3):                                   # the second call is smaller and runs over the first one's leftovers
        x = torch.randn(B, H, D, device='cuda', generator=g)
        t = torch.randint(0, 20, (B,), device='cuda', generator=g)
        ref, out = _both(net, x, t)
        assert np.isfinite(ref).all()
        assert np.array_equal(ref, out), f'B={B}: static-scratch forward differs from the workspace forward'
    # and against the golden vector itself (reference output), like test_gpu_nets
    B = int(fx['B'])
    x = torch.from_numpy(fx['net_x']).cuda()
    ref, out = _both(net, x, torch.full((B,), 5, device='cuda', dtype=torch.long))
    assert np.array_equal(ref, out)
    err = np.abs(out - fx['unet_out_t5']).max() / np.abs(fx['unet_out_t5']).max()
    assert err < 1e-4, err


def test_static_scratch_rejects_oversized_batches():
    from trebi_amd import _lib
    lib = _lib.lib()
    _, net, H, D = _unet('halfcheetah')
    h = net.net_handle(torch.device('cuda', 0))
    B = 1
    while lib.trebi_unet_workspace_bytes(h, B) <= (32 << 20):
        B *= 2
    x = torch.zeros(B, H, D, device='cuda')
    t = torch.zeros(B, device='cuda', dtype=torch.long)
    out = torch.empty_like(x)
    rc = lib.trebi_unet_forward_static(h, _lib.ptr(x), D, _lib.ptr(t), _lib.ptr(out), D, B, None)
    assert rc == -4                                    # TREBI_WORKSPACE_TOO_SMALL
    assert b'static' in lib.trebi_last_error()
    torch.cuda.synchronize()
