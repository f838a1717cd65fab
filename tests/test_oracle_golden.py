"""CPU suite: the numpy oracle (oracle/trebi_oracle.py) against the golden vectors the
reference itself produced (tests/golden/make_golden.py).  This is what pins the oracle.

Tolerances: the reference ran through PyTorch's CPU kernels (oneDNN conv, vectorised
GroupNorm), the oracle through numpy/OpenBLAS; both are float32 with different summation
orders, so single-network outputs are compared at 2e-5 relative and 20-step chains (errors
feed back through x) at 2e-4 -- both far inside north_star's 1e-3 bar."""
import numpy as np
import pytest

import trebi_oracle as O
from _common import CONFIG_NAMES, fixture_kwargs, fixture_weights, load_fixture, rel_err, GOLDEN
import os

NET_TOL = 2e-5
CHAIN_TOL = 2e-4


def test_schedule_tables():
    ref = np.load(os.path.join(GOLDEN, 'schedule_T20.npz'))
    mine = O.make_schedule(20)
    for k in ref.files:
        np.testing.assert_allclose(mine[k], ref[k], rtol=2e-6, atol=1e-9, err_msg=k)
    # the float artefacts SURVEY.md 8a row 9 calls out
    assert abs(mine['posterior_mean_coef1'][0] - 0.99999708) < 1e-6
    assert mine['posterior_mean_coef2'][0] == 0
    assert abs(mine['posterior_log_variance_clipped'][0] - np.log(np.float32(1e-20))) < 1e-4


@pytest.fixture(scope='module', params=CONFIG_NAMES)
def setup(request):
    fx = load_fixture(request.param)
    sd_u, sd_r, sd_c, um, vm = fixture_weights(fx)
    H, obs, act = int(fx['H']), int(fx['obs']), int(fx['act'])
    D = obs + act
    unet = O.TemporalUnet(sd_u, H, D, int(fx['dim']), um)
    rnet = O.ValueFunction(sd_r, H, D, int(fx['dim']), vm)
    cnet = O.ValueFunction(sd_c, H, D, int(fx['dim']), vm)
    return fx, unet, rnet, cnet


def test_unet_forward(setup):
    fx, unet, _, _ = setup
    B = int(fx['B'])
    for tt in (0, 5, 19):
        out = unet(fx['net_x'], None, np.full((B,), tt))
        assert rel_err(out, fx[f'unet_out_t{tt}']) < NET_TOL
    taps = {}
    unet.forward(fx['net_x'], None, np.full((B,), 5), taps=taps)
    assert rel_err(taps['down0_block0'], fx['unet_tap_down0_block0_t5']) < NET_TOL
    assert rel_err(taps['mid_out'], fx['unet_tap_mid_out_t5']) < NET_TOL


def test_value_and_gradient(setup):
    fx, _, rnet, cnet = setup
    B = int(fx['B'])
    for tag, net in (('reward', rnet), ('cost', cnet)):
        g = O.ValueGuide(net)
        for tt in (0, 5, 19):
            y, grad = g.gradients(fx['net_x'].copy(), None, np.full((B,), tt))
            assert rel_err(y, fx[f'{tag}_y_t{tt}']) < NET_TOL
            assert rel_err(grad, fx[f'{tag}_grad_t{tt}']) < 5e-5


def _loop(fx, unet, rnet, cnet, which):
    B, H, T = int(fx['B']), int(fx['H']), int(fx['T'])
    obs, act = int(fx['obs']), int(fx['act'])
    diff = O.GaussianDiffusion(unet, H, obs, act, T)
    cond = {0: np.repeat(fx['cond0'][None], B, 0)}
    kw = fixture_kwargs(fx)
    guide, cguide = O.ValueGuide(rnet), O.ValueGuide(cnet)
    thr = np.full((B,), fx['cost_threshold'], np.float32)
    if which == 'constrained':
        fn, skw = O.n_step_guided_and_constrained_p_sample, dict(guide=guide, cost_guide=cguide, cost_threshold=thr, **kw)
    elif which == 'constrained_gt':
        kw['guide_type'] = 1 - kw['guide_type']
        fn, skw = O.n_step_guided_and_constrained_p_sample, dict(guide=guide, cost_guide=cguide, cost_threshold=thr, **kw)
    elif which == 'constrained_infeasible':
        fn, skw = O.n_step_guided_and_constrained_p_sample, dict(
            guide=guide, cost_guide=cguide, cost_threshold=np.full((B,), -1e3, np.float32), **kw)
    elif which == 'guided':
        gkw = {k: v for k, v in kw.items() if k not in ('cost_grad_weight', 'guide_type')}
        fn, skw = O.n_step_guided_p_sample, dict(guide=guide, **gkw)
    else:
        fn, skw = O.default_sample_fn, {}
    return diff.p_sample_loop((B, H, obs + act), cond, fx['noise'], fn, return_chain=True, **skw)


@pytest.mark.parametrize('which', ['constrained', 'constrained_gt', 'constrained_infeasible', 'guided', 'default'])
def test_sample_loop(setup, which):
    fx, unet, rnet, cnet = setup
    x, values, costs, chain = _loop(fx, unet, rnet, cnet, which)
    assert rel_err(chain, fx[f'{which}_chain']) < CHAIN_TOL
    assert rel_err(x, fx[f'{which}_traj']) < CHAIN_TOL
    gv, gc = fx[f'{which}_values'], fx[f'{which}_costs']
    # penalised values are r - 1e8 in float32 (quantum 8): compare feasible ones tightly, the rest by branch
    pen = gv < -1e7
    assert np.array_equal(pen, values < -1e7)
    np.testing.assert_allclose(values[~pen], gv[~pen], rtol=0, atol=CHAIN_TOL * max(1.0, np.abs(gv[~pen]).max(initial=0)))
    np.testing.assert_allclose(costs, gc, rtol=0, atol=CHAIN_TOL)
