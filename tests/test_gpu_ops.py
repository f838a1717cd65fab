"""GPU parity, one layer at a time (level 1 of the pyramid): every launch of the TemporalUnet forward,
the ValueFunction forward and the ValueFunction backward-to-input is held to the oracle's unit for that module --
Conv1dBlock (helpers.py:46-63) with the block's time bias / residual (temporal.py:37-46), Downsample1d / Upsample1d
(helpers.py:30-44), the final 1x1 conv, the value head (temporal.py:207-211,237-238) and, backwards, the input-gradient
convolutions behind the GroupNorm + Mish backward -- on the tensors the library itself fed that launch
(trebi_net_layer_info names them in the workspace).  An error in one layer therefore fails under that layer's name only; it
does not travel into the layers behind it.  Bar: 1e-4 of the unit's largest output (exact-fp32 MFMA; summation order only).

What is under test is the standalone launch path (trebi_unet_forward / trebi_value_gradients: conv_gemm_kernel and
head_kernel); the resident chains of the fused loop are held to the golden chains strategy by strategy in test_gpu_paths.py."""
import os
import sys

import numpy as np
import pytest
import torch

from _common import CONFIG_NAMES, fixture_weights, load_fixture, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))

pytestmark = pytest.mark.gpu
TOL = 1e-4
B = 6                                      # does not fill the candidate tiles of the deep layers


def _mk(cls, sd, *a, **k):
    net = cls(*a, **k)
    net.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()}, strict=True)
    net.set_max_timesteps(20)
    return net.to('cuda')


@pytest.fixture(scope='module', params=CONFIG_NAMES)
def setup(request):
    import trebi_oracle as O
    from trebi_amd.nets import TemporalUnet, ValueFunction
    fx = load_fixture(request.param)
    sd_u, sd_r, _, um, vm = fixture_weights(fx)
    H, D, dim = int(fx['H']), int(fx['obs']) + int(fx['act']), int(fx['dim'])
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, H, D)).astype(np.float32)
    t = rng.integers(0, 20, size=B)
    return dict(O=O, x=x, t=t, dim=dim,
                unet=_mk(TemporalUnet, sd_u, H, D, 0, dim=dim, dim_mults=um), ounet=O.TemporalUnet(sd_u, H, D, dim, um),
                vnet=_mk(ValueFunction, sd_r, H, D, 0, dim=dim, dim_mults=vm), ovnet=O.ValueFunction(sd_r, H, D, dim, vm))


def _cf(v):
    """[B, L, C] device view -> the oracle's [B, C, L] numpy layout."""
    return np.ascontiguousarray(v.detach().cpu().numpy().transpose(0, 2, 1))


class _Pass:
    """The tensors of one library call, by tensor reference."""

    def __init__(self, view, x, out=None, grad=None):
        self.view, self.ext = view, {2: x, 3: out, 4: grad}

    def get(self, ref):
        if ref.where == 0:
            return None
        if ref.where == 1:
            return _cf(self.view(ref))
        return _cf(self.ext[ref.where][:, :, :ref.C])


def _forward_unit(O, onet, info, src, src2, res, temb):
    """What the reference's module computes for the layer `info` names, from that layer's own inputs."""
    name = info.name.decode()
    if name.endswith('.blocks.0.block.0'):                       # Conv1dBlock + time bias (temporal.py:43)
        blk = name[:-len('.blocks.0.block.0')]
        tb = O.linear(O.mish(temb), onet.p(blk + '.time_mlp.1.weight'), onet.p(blk + '.time_mlp.1.bias'))
        return onet.conv_block(blk + '.blocks.0', src) + tb[:, :, None]
    if name.endswith('.blocks.1.block.0'):                       # Conv1dBlock + residual (temporal.py:44-45)
        blk = name[:-len('.blocks.1.block.0')]
        out = onet.conv_block(blk + '.blocks.1', src)
        if blk + '.residual_conv.weight' in onet.sd:
            assert src2 is not None and res is None, name
            return out + O.conv1d(src2, onet.p(blk + '.residual_conv.weight'), onet.p(blk + '.residual_conv.bias'))
        assert res is not None and src2 is None, name
        return out + res
    if name == 'final_conv.0.block.0':
        return onet.conv_block('final_conv.0', src)
    if name == 'final_conv.1':
        return O.conv1d(src, onet.p('final_conv.1.weight'), onet.p('final_conv.1.bias'))
    if name.endswith('.conv'):
        return onet.upsample(name[:-5], src) if name.startswith('ups.') else onet.downsample(name[:-5], src)
    raise AssertionError(f'no unit for layer {name}')


def _check_forward(s, net, onet, layers, P, y=None):
    O = s['O']
    temb = onet.time_embed(s['t'], s['dim'])
    seen, worst = set(), []
    for info in layers:
        name = info.name.decode()
        src, src2, res = P.get(info.src), P.get(info.src2), P.get(info.res)
        if info.kind == 1:                                       # value head: Linear -> Mish -> Linear, and dy/dv
            fc = info.Cin
            W0, b0 = onet.p('final_block.0.weight'), onet.p('final_block.0.bias')
            W2, b2 = onet.p('final_block.2.weight'), onet.p('final_block.2.bias')
            h1 = O.linear(np.concatenate([src.reshape(len(src), -1), temb], axis=-1), W0, b0)
            want_y = O.linear(O.mish(h1), W2, b2)[:, 0]
            want_dv = ((np.ones((len(src), 1), np.float32) @ W2) * O.mish_grad(h1)) @ W0[:, :fc]
            errs = {'y': rel_err(y.cpu().numpy(), want_y), 'dy/dv': rel_err(P.get(info.dst)[:, :, 0], want_dv)}
        else:
            cache = {}
            want = _forward_unit(O, onet, info, src, src2, res, temb)
            errs = {'out': rel_err(P.get(info.dst), want)}
            if info.saved_xhat.where:                            # what the backward prologue will read
                pre = name[:-len('.block.0')]
                onet.conv_block(pre, src, cache)
                _, xhat, rstd, z = cache[pre]
                gamma = onet.p(pre + '.block.2.weight')
                errs['saved xhat'] = rel_err(P.get(info.saved_xhat), xhat)
                if info.saved_gain.where:
                    errs["saved mish'(z)*gamma"] = rel_err(P.get(info.saved_gain), O.mish_grad(z) * gamma[None, :, None])
                st = P.view(info.saved_stats).cpu().numpy()         # [B, groups, (mean, rstd)]
                errs['saved rstd'] = rel_err(st[:, :, 1], rstd.reshape(len(src), -1))
        seen.add(name)
        worst += [(e, name, k) for k, e in errs.items() if not e < TOL]
    assert not worst, f'layers off their unit: {sorted(worst, reverse=True)}'
    return seen


def test_every_unet_layer_against_its_module(setup):
    s = setup
    x = torch.from_numpy(s['x']).cuda()
    out = s['unet'](x, None, torch.from_numpy(s['t']).cuda())
    layers, view = s['unet'].layers(B)
    seen = _check_forward(s, s['unet'], s['ounet'], layers, _Pass(view, x, out=out))
    # every weight-carrying module of the state_dict was visited exactly where the forward uses it
    convs = {k[:-len('.weight')] for k, v in s['ounet'].sd.items() if k.endswith('.weight') and v.ndim == 3
             and '.residual_conv' not in k}
    assert convs == seen, sorted(convs ^ seen)


def test_every_value_layer_and_the_head_against_its_module(setup):
    s = setup
    x = torch.from_numpy(s['x']).cuda()
    y, g = s['vnet'].value_and_input_grad(x, torch.from_numpy(s['t']).cuda())
    layers, view = s['vnet'].layers(B)
    seen = _check_forward(s, s['vnet'], s['ovnet'], layers, _Pass(view, x, grad=g), y=y)
    convs = {k[:-len('.weight')] for k, v in s['ovnet'].sd.items() if k.endswith('.weight') and v.ndim == 3
             and '.residual_conv' not in k}
    assert convs | {'final_block'} == seen, sorted((convs | {'final_block'}) ^ seen)


def test_every_value_backward_layer_against_its_module(setup):
    """Input-gradient units: d(conv stride 2), and per Conv1dBlock dX = conv^T(GN'(mish'(z) * dY)) + the residual path,
    with the forward cache RECOMPUTED by the oracle from the forward layer's own input -- so the library's saved
    xhat / gain / statistics are what is being exercised."""
    s, O, onet = setup, setup['O'], setup['ovnet']
    x = torch.from_numpy(s['x']).cuda()
    y, g = s['vnet'].value_and_input_grad(x, torch.from_numpy(s['t']).cuda())
    fwd, view = s['vnet'].layers(B)
    bwd, _ = s['vnet'].layers(B, backward=True)
    P = _Pass(view, x, grad=g)
    fwd_src = {f.name.decode(): P.get(f.src) for f in fwd if f.kind == 0}
    worst, seen = [], set()
    for info in bwd:
        name = info.name.decode()
        assert name.endswith(' (dgrad)'), name
        base = name[:-len(' (dgrad)')]
        dY, src2, res = P.get(info.src), P.get(info.src2), P.get(info.res)
        if base.endswith('.conv'):
            assert not info.prologue
            want = onet.downsample_input_grad(base[:-5], dY, info.dst.L)
        else:
            assert info.prologue and info.pro_xhat.where and info.pro_stats.where, name
            pre = base[:-len('.block.0')]
            cache = {}
            onet.conv_block(pre, fwd_src[base], cache)
            want = onet.conv_block_input_grad(pre, dY, cache)
            if pre.endswith('.blocks.0'):                        # the block's input also feeds the residual path
                blk = pre[:-len('.blocks.0')]
                if blk + '.residual_conv.weight' in onet.sd:
                    want = want + O.conv1d_input_grad(src2, onet.p(blk + '.residual_conv.weight'), src2.shape[2])
                else:
                    want = want + res
        e = rel_err(P.get(info.dst), want)
        seen.add(base)
        if not e < TOL:
            worst.append((e, name))
    assert not worst, f'backward layers off their unit: {sorted(worst, reverse=True)}'
    assert seen == set(fwd_src), sorted(seen ^ set(fwd_src))
