"""Architecture plans and parameter shapes of the two hot-path networks.

This is the single Python-side description of *what tensors exist* in a
TemporalUnet / ValueFunction checkpoint.  The names and shapes are the
reference's ``state_dict()`` keys so that a reference ``state_{epoch}.pt``
``'ema'`` dict loads unchanged:

* TemporalUnet   -- /root/reference/diffuser/models/temporal.py:49-118
* ValueFunction  -- /root/reference/diffuser/models/temporal.py:153-211
* block internals -- temporal.py:18-46, helpers.py:30-63

The HIP library (csrc/trebi.hip, make_arch / param_specs) derives the same plan from the same five
integers; tests/test_host.py checks that the two agree on every name.
"""
from collections import OrderedDict

GN_GROUPS = 8          # helpers.py:51  n_groups=8
KSIZE = 5              # temporal.py:20 kernel_size=5


def _resblock(shapes, prefix, cin, cout, embed_dim):
    # temporal.py:23-35
    for j, (ci, co) in enumerate(((cin, cout), (cout, cout))):
        p = f'{prefix}.blocks.{j}.block'
        shapes[f'{p}.0.weight'] = (co, ci, KSIZE)     # Conv1d
        shapes[f'{p}.0.bias'] = (co,)
        shapes[f'{p}.2.weight'] = (co,)               # GroupNorm gamma
        shapes[f'{p}.2.bias'] = (co,)                 # GroupNorm beta
    shapes[f'{prefix}.time_mlp.1.weight'] = (cout, embed_dim)
    shapes[f'{prefix}.time_mlp.1.bias'] = (cout,)
    if cin != cout:
        shapes[f'{prefix}.residual_conv.weight'] = (cout, cin, 1)
        shapes[f'{prefix}.residual_conv.bias'] = (cout,)


def _time_mlp(shapes, dim):
    # temporal.py:70-75 / 170-175
    shapes['time_mlp.1.weight'] = (dim * 4, dim)
    shapes['time_mlp.1.bias'] = (dim * 4,)
    shapes['time_mlp.3.weight'] = (dim, dim * 4)
    shapes['time_mlp.3.bias'] = (dim,)


def unet_plan(horizon, transition_dim, dim=32, dim_mults=(1, 2, 4, 8), output_dim=None):
    """Resolve channel / length bookkeeping of TemporalUnet (temporal.py:63-118)."""
    if output_dim is None:
        output_dim = transition_dim
    dims = [transition_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    hist = []
    h = horizon
    downs = []
    for ind, (ci, co) in enumerate(in_out):
        hist.append(h)
        last = ind >= n - 1
        downs.append(dict(cin=ci, cout=co, L=h, down=not last))
        if not last:
            h = max(h // 2, 1)
    mid = dict(c=dims[-1], L=h)
    ups = []
    for ind, (ci, co) in enumerate(reversed(in_out[1:])):
        has_up = hist[-(ind + 1)] != hist[-(ind + 2)]
        ups.append(dict(cin=co * 2, cout=ci, L=h, up=has_up))
        h = hist[-(ind + 2)]
    return dict(kind='unet', horizon=horizon, transition_dim=transition_dim, dim=dim,
                dim_mults=tuple(dim_mults), output_dim=output_dim,
                downs=downs, mid=mid, ups=ups, final_L=h)


def unet_param_shapes(horizon, transition_dim, dim=32, dim_mults=(1, 2, 4, 8), output_dim=None):
    plan = unet_plan(horizon, transition_dim, dim, dim_mults, output_dim)
    s = OrderedDict()
    _time_mlp(s, dim)
    for i, d in enumerate(plan['downs']):
        _resblock(s, f'downs.{i}.0', d['cin'], d['cout'], dim)
        _resblock(s, f'downs.{i}.1', d['cout'], d['cout'], dim)
        if d['down']:
            s[f'downs.{i}.3.conv.weight'] = (d['cout'], d['cout'], 3)   # helpers.py:33
            s[f'downs.{i}.3.conv.bias'] = (d['cout'],)
    # registration order in the reference: downs and ups are created first (temporal.py:78-79)
    for i, u in enumerate(plan['ups']):
        _resblock(s, f'ups.{i}.0', u['cin'], u['cout'], dim)
        _resblock(s, f'ups.{i}.1', u['cout'], u['cout'], dim)
        if u['up']:
            s[f'ups.{i}.3.conv.weight'] = (u['cout'], u['cout'], 4)     # ConvTranspose1d [Cin,Cout,K], helpers.py:41
            s[f'ups.{i}.3.conv.bias'] = (u['cout'],)
    c = plan['mid']['c']
    _resblock(s, 'mid_block1', c, c, dim)
    _resblock(s, 'mid_block2', c, c, dim)
    p = 'final_conv.0.block'
    s[f'{p}.0.weight'] = (dim, dim, KSIZE)
    s[f'{p}.0.bias'] = (dim,)
    s[f'{p}.2.weight'] = (dim,)
    s[f'{p}.2.bias'] = (dim,)
    s['final_conv.1.weight'] = (plan['output_dim'], dim, 1)
    s['final_conv.1.bias'] = (plan['output_dim'],)
    return s


def value_plan(horizon, transition_dim, dim=32, dim_mults=(1, 2, 4, 8), out_dim=1):
    """Resolve ValueFunction bookkeeping (temporal.py:166-211).

    ``L`` below is the *actual* sequence length seen by each stage (every stage
    ends in a stride-2 Downsample1d, temporal.py:187), ``fc_dim`` is what the
    constructor computes from its own ``horizon`` counter (temporal.py:190-205).
    The two only agree when the final length is 1 -- see SURVEY.md fact 3.
    """
    dims = [transition_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    down_len = lambda L: (L + 2 - 3) // 2 + 1          # Conv1d(k=3, s=2, p=1)
    h = horizon
    L = horizon
    blocks = []
    for ind, (ci, co) in enumerate(in_out):
        blocks.append(dict(cin=ci, cout=co, L=L))
        L = down_len(L)
        if not ind >= n - 1:
            h = h // 2
    c = dims[-1]
    c2, c3 = c // 2, c // 4
    mid1 = dict(cin=c, cout=c2, L=L)
    L = down_len(L)
    h = h // 2
    mid2 = dict(cin=c2, cout=c3, L=L)
    L = down_len(L)
    h = h // 2
    fc_dim = c3 * max(h, 1)
    return dict(kind='value', horizon=horizon, transition_dim=transition_dim, dim=dim,
                dim_mults=tuple(dim_mults), out_dim=out_dim, blocks=blocks,
                mid1=mid1, mid2=mid2, final_L=L, c3=c3, fc_dim=fc_dim,
                consistent=(fc_dim == c3 * L))


def value_param_shapes(horizon, transition_dim, dim=32, dim_mults=(1, 2, 4, 8), out_dim=1):
    plan = value_plan(horizon, transition_dim, dim, dim_mults, out_dim)
    s = OrderedDict()
    _time_mlp(s, dim)
    for i, b in enumerate(plan['blocks']):
        _resblock(s, f'blocks.{i}.0', b['cin'], b['cout'], dim)
        _resblock(s, f'blocks.{i}.1', b['cout'], b['cout'], dim)
        s[f'blocks.{i}.2.conv.weight'] = (b['cout'], b['cout'], 3)
        s[f'blocks.{i}.2.conv.bias'] = (b['cout'],)
    m1, m2 = plan['mid1'], plan['mid2']
    _resblock(s, 'mid_block1', m1['cin'], m1['cout'], dim)
    s['mid_down1.conv.weight'] = (m1['cout'], m1['cout'], 3)
    s['mid_down1.conv.bias'] = (m1['cout'],)
    _resblock(s, 'mid_block2', m2['cin'], m2['cout'], dim)
    s['mid_down2.conv.weight'] = (m2['cout'], m2['cout'], 3)
    s['mid_down2.conv.bias'] = (m2['cout'],)
    fc = plan['fc_dim']
    s['final_block.0.weight'] = (fc // 2, fc + dim)
    s['final_block.0.bias'] = (fc // 2,)
    s['final_block.2.weight'] = (out_dim, fc // 2)
    s['final_block.2.bias'] = (out_dim,)
    return s
