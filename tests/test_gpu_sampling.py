"""GPU parity of the sampling loop: the captured HIP loop (trebi_sampler_*) and the per-step
sample functions against the reference's golden chains (same weights, same injected noise).

Tolerance: 20 sequential denoise steps feed rounding differences back through x, so per-step
trajectories are compared at 5e-4 relative (north_star: 1e-3), values/costs at 5e-4 absolute, and
the selected candidate indices must match wherever the golden score gaps exceed that tolerance."""
import numpy as np
import pytest
import torch

from _common import CONFIG_NAMES, fixture_kwargs, fixture_weights, load_fixture, rel_err

pytestmark = pytest.mark.gpu
TOL = 5e-4


def _mk(cls, sd, *a, **k):
    net = cls(*a, **k)
    net.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()}, strict=True)
    return net


@pytest.fixture(scope='module', params=CONFIG_NAMES)
def setup(request):
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide
    fx = load_fixture(request.param)
    sd_u, sd_r, sd_c, um, vm = fixture_weights(fx)
    H, obs, act, dim, T = int(fx['H']), int(fx['obs']), int(fx['act']), int(fx['dim']), int(fx['T'])
    D = obs + act
    kw = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)
    diff = GaussianDiffusion(_mk(TemporalUnet, sd_u, H, D, 0, dim=dim, dim_mults=um), H, obs, act, **kw).cuda()
    guide = ValueGuide(ValueDiffusion(_mk(ValueFunction, sd_r, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **kw)).cuda()
    cguide = ValueGuide(ValueDiffusion(_mk(ValueFunction, sd_c, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **kw)).cuda()
    return fx, diff, guide, cguide


def _args(fx, which, guide, cguide):
    from diffuser.sampling import n_step_guided_p_sample, n_step_guided_and_constrained_p_sample
    from diffuser.models.diffusion import default_sample_fn
    B = int(fx['B'])
    kw = fixture_kwargs(fx)
    thr = torch.full((B,), float(fx['cost_threshold']), device='cuda')
    if which == 'constrained':
        return n_step_guided_and_constrained_p_sample, dict(guide=guide, cost_guide=cguide, cost_threshold=thr, **kw)
    if which == 'constrained_gt':
        kw['guide_type'] = 1 - kw['guide_type']
        return n_step_guided_and_constrained_p_sample, dict(guide=guide, cost_guide=cguide, cost_threshold=thr, **kw)
    if which == 'constrained_infeasible':
        return n_step_guided_and_constrained_p_sample, dict(
            guide=guide, cost_guide=cguide, cost_threshold=torch.full((B,), -1e3, device='cuda'), **kw)
    if which == 'guided':
        gkw = {k: v for k, v in kw.items() if k not in ('cost_grad_weight', 'guide_type')}
        return n_step_guided_p_sample, dict(guide=guide, **gkw)
    return default_sample_fn, {}


def _check(fx, which, s):
    chain = s.chains.cpu().numpy()
    assert rel_err(chain, fx[f'{which}_chain']) < TOL
    gv, gc = fx[f'{which}_values'], fx[f'{which}_costs']
    v, c = s.values.cpu().numpy(), s.costs.cpu().numpy()
    pen = gv < -1e7
    assert np.array_equal(pen, v < -1e7)
    np.testing.assert_allclose(v[~pen], gv[~pen], rtol=0, atol=TOL)
    np.testing.assert_allclose(c, gc, rtol=0, atol=TOL)
    # selected candidates: identical order wherever neighbouring golden scores differ by more than the tolerance
    gaps = np.abs(np.diff(gv))
    if gaps.size == 0 or gaps.min() > 2 * TOL:
        assert rel_err(s.trajectories.cpu().numpy(), fx[f'{which}_traj']) < TOL
    else:
        assert rel_err(s.trajectories[0].cpu().numpy(), fx[f'{which}_traj'][0]) < TOL or gaps[0] <= 2 * TOL


WHICH = ['constrained', 'constrained_gt', 'constrained_infeasible', 'guided', 'default']


@pytest.mark.parametrize('which', WHICH)
def test_fused_loop_golden(setup, which):
    fx, diff, guide, cguide = setup
    B, H, D = int(fx['B']), int(fx['H']), int(fx['obs']) + int(fx['act'])
    fn, kw = _args(fx, which, guide, cguide)
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    noise = torch.from_numpy(fx['noise']).cuda()
    s = diff.p_sample_loop((B, H, D), cond, verbose=False, return_chain=True, sample_fn=fn, noise=noise, **kw)
    _check(fx, which, s)
    # replaying the captured graph gives bit-identical results
    s2 = diff.p_sample_loop((B, H, D), cond, verbose=False, return_chain=True, sample_fn=fn, noise=noise, **kw)
    assert torch.equal(s.chains, s2.chains) and torch.equal(s.values, s2.values)


@pytest.mark.parametrize('which', ['constrained', 'guided', 'default'])
def test_stepwise_functions_golden(setup, which):
    """The reference's own control flow: p_sample_loop calling a per-step sample_fn (here wrapped in a
    lambda so the fused route is not taken)."""
    fx, diff, guide, cguide = setup
    B, H, D = int(fx['B']), int(fx['H']), int(fx['obs']) + int(fx['act'])
    fn, kw = _args(fx, which, guide, cguide)
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    noise = torch.from_numpy(fx['noise']).cuda()
    step = lambda *a, **k: fn(*a, **k)
    s = diff.p_sample_loop((B, H, D), cond, verbose=False, return_chain=True, sample_fn=step, noise=noise, **kw)
    _check(fx, which, s)


def test_eager_graph_and_dead_eval_elimination_agree(setup):
    """Plain stream launches, the captured graph, and the variant that skips guide evaluations whose
    results cannot reach an output (t < t_stopgrad, t != 0) must agree bit for bit."""
    from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
    fx, diff, guide, cguide = setup
    B, H, D = int(fx['B']), int(fx['H']), int(fx['obs']) + int(fx['act'])
    _, kw = _args(fx, 'constrained', guide, cguide)
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    noise = torch.from_numpy(fx['noise']).cuda()
    outs = []
    for opts in (dict(use_graph=True), dict(use_graph=False), dict(use_graph=True, skip_dead_evals=True)):
        x, v, c, _ = diff.sample_raw((B, H, D), cond, sample_fn=fn, noise=noise, **opts, **kw)
        outs.append((x.clone(), v.clone(), c.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))


def test_selection_kernels():
    import trebi_oracle as O
    from trebi_amd.diffusion import feasibility_rule, argsort_descending, max_by_values_in_batch
    rng = np.random.default_rng(0)
    for n in (1, 5, 64, 1000, 5000):
        r = rng.standard_normal(n).astype(np.float32)
        c = rng.standard_normal(n).astype(np.float32)
        r[rng.integers(0, n, n // 4)] = 0.5          # ties
        for thr in (0.0, -10.0, 10.0):
            t = np.full(n, thr, np.float32)
            got = feasibility_rule(torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda(), torch.from_numpy(t).cuda())
            exp = O.feasibility_rule(r, c, t)
            assert np.array_equal(got.cpu().numpy(), exp)
            order = argsort_descending(got).cpu().numpy()
            assert np.array_equal(order, np.argsort(-exp, kind='stable'))
    v = torch.from_numpy(rng.standard_normal(6 * 37).astype(np.float32)).cuda()
    x = torch.arange(6 * 37, device='cuda', dtype=torch.float32).reshape(-1, 1, 1)
    bx, bv, bc = max_by_values_in_batch(x, v, v, 6)
    exp = v.reshape(6, 37).argmax(1) + torch.arange(6, device='cuda') * 37
    assert torch.equal(bx.reshape(-1).long(), exp) and torch.equal(bv, v[exp])


def test_selection_at_the_global_size_with_ties_and_nan():
    """The selection every rank runs on the all-gathered scores of configs[3] (65536 candidates): descending argsort with
    massive ties (infeasible scores are quantised by the -1e8 shift, functions.py:129) and NaN, which torch.argsort
    (sort_by_values, diffusion.py:44-49) orders BEFORE every number; torch.argmax (max_by_values_in_batch) returns the first
    NaN of a group.  Bit-exact against torch on the host (stable sort: the kernel breaks ties by index)."""
    import time
    from trebi_amd.diffusion import argsort_descending, max_by_values_in_batch
    rng = np.random.default_rng(5)
    n = 65536
    v = rng.standard_normal(n).astype(np.float32)
    v[rng.integers(0, n, n // 3)] = np.float32(-1e8)          # ties
    v[rng.integers(0, n, n // 8)] = 0.25
    v[rng.integers(0, n, 37)] = np.nan
    tv = torch.from_numpy(v)
    want = torch.argsort(tv, descending=True, stable=True)
    dv = tv.cuda()
    got = argsort_descending(dv)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    t0 = time.perf_counter()
    for _ in range(5):
        argsort_descending(dv)
    torch.cuda.synchronize()
    print(f'argsort_desc_kernel at n = {n}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms')
    for rows, per in ((64, 1024), (7, 333)):
        g = v[:rows * per].copy()
        g[3 * per + 5] = np.nan
        g[3 * per + 9] = np.nan
        gt = torch.from_numpy(g)
        x = torch.arange(rows * per, device='cuda', dtype=torch.float32).reshape(-1, 1, 1)
        bx, bv, bc = max_by_values_in_batch(x, gt.cuda(), gt.cuda(), rows)
        exp = gt.reshape(rows, per).argmax(1) + torch.arange(rows) * per
        assert torch.equal(bx.reshape(-1).long().cpu(), exp)


def test_sharded_planner_single_rank_equals_loop(setup):
    """world_size 1: ShardedPlanner.plan == p_sample_loop's best candidate; and slicing the batch in two
    (a fake 2-rank world on one device) reproduces the unsharded result (candidates are independent)."""
    from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
    from trebi_amd.parallel import ShardedPlanner
    fx, diff, guide, cguide = setup
    B, H, D = int(fx['B']), int(fx['H']), int(fx['obs']) + int(fx['act'])
    _, kw = _args(fx, 'constrained', guide, cguide)
    thr = kw.pop('cost_threshold')
    noise = torch.from_numpy(fx['noise']).cuda()
    obs = torch.from_numpy(fx['cond0']).cuda()
    plan = ShardedPlanner(diff).plan(obs, B, H, fn, cost_threshold=thr, noise=noise, **kw)
    cond = {0: obs[None].repeat(B, 1)}
    s = diff.p_sample_loop((B, H, D), cond, verbose=False, sample_fn=fn, noise=noise, cost_threshold=thr, **kw)
    assert torch.equal(plan.best_trajectory, s.trajectories[0])
    assert torch.equal(plan.values[plan.order], s.values)
    if B >= 2:
        h = B // 2
        parts = []
        for lo, hi in ((0, h), (h, B)):
            x, v, c, _ = diff.sample_raw((hi - lo, H, D), {0: obs[None].repeat(hi - lo, 1)}, sample_fn=fn,
                                         noise=noise[:, lo:hi], cost_threshold=thr[lo:hi], **kw)
            parts.append((x.clone(), v.clone(), c.clone()))
        xs = torch.cat([p[0] for p in parts]); vs = torch.cat([p[1] for p in parts])
        assert rel_err(xs.cpu().numpy(), plan.local_trajectories.cpu().numpy()) < 1e-5
        assert torch.allclose(vs, diff.sample_raw((B, H, D), cond, sample_fn=fn, noise=noise, cost_threshold=thr, **kw)[1], atol=1e-5)


def test_epsilon_parameterised_and_clipped_model_follows_the_reference_formulas():
    """predict_epsilon=True / clip_denoised=True (the constructor defaults, diffusion.py:75) never take the fused
    x0 loop: sample_raw refuses them, and the per-step route builds x_recon as p_mean_variance does
    (diffusion.py:153-164,176-181).  Expected values: the same torch expressions on the HIP net's own output
    (that output is pinned by the golden UNet vectors)."""
    from diffuser.models import TemporalUnet, GaussianDiffusion
    from diffuser.models.diffusion import default_sample_fn
    fx = load_fixture('halfcheetah')
    sd_u, _, _, um, _ = fixture_weights(fx)
    H, obs, act, dim, T = int(fx['H']), int(fx['obs']), int(fx['act']), int(fx['dim']), int(fx['T'])
    D, B = obs + act, int(fx['B'])
    unet = _mk(TemporalUnet, sd_u, H, D, 0, dim=dim, dim_mults=um)
    d = GaussianDiffusion(unet, H, obs, act, n_timesteps=T, loss_type='l2', clip_denoised=True,
                          predict_epsilon=True).cuda()
    x = torch.from_numpy(fx['net_x']).cuda()
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    nz = torch.from_numpy(fx['noise'][1]).cuda()
    with pytest.raises(RuntimeError, match='x0-parameterised'):
        d.sample_raw((B, H, D), cond, sample_fn=default_sample_fn)
    for tt in (0, 7, 19):
        t = torch.full((B,), tt, device='cuda', dtype=torch.long)
        out, _, _ = default_sample_fn(d, x, cond, t, noise=nz)
        eps = unet(x, cond, t)
        xr = (d.sqrt_recip_alphas_cumprod[tt] * x - d.sqrt_recipm1_alphas_cumprod[tt] * eps).clamp(-1., 1.)
        want = d.posterior_mean_coef1[tt] * xr + d.posterior_mean_coef2[tt] * x
        if tt != 0:
            want = want + torch.exp(0.5 * d.posterior_log_variance_clipped[tt]) * nz
        want[:, 0, act:] = cond[0]
        assert rel_err(out.cpu().numpy(), want.cpu().numpy()) < 1e-6, tt
    # the whole loop runs step by step for such a model and stays finite
    s = d.p_sample_loop((B, H, D), cond, verbose=False, sample_fn=default_sample_fn,
                        noise=torch.from_numpy(fx['noise']).cuda())
    assert torch.isfinite(s.trajectories).all()


def test_new_weights_after_a_captured_loop_are_used():
    """load_state_dict after a planning call frees the packed weights the captured loop pointed into: the stale
    samplers must go with them, and the next call must equal a freshly built model's (use-after-free)."""
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide, n_step_guided_and_constrained_p_sample as fn
    from trebi_amd import spec, synth
    fx = load_fixture('halfcheetah')
    sd_u, sd_r, sd_c, um, vm = fixture_weights(fx)
    H, obs, act, dim, T = int(fx['H']), int(fx['obs']), int(fx['act']), int(fx['dim']), int(fx['T'])
    D, B = obs + act, int(fx['B'])
    dk = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)

    def build(su, sr, sc):
        diff = GaussianDiffusion(_mk(TemporalUnet, su, H, D, 0, dim=dim, dim_mults=um), H, obs, act, **dk).cuda()
        g = ValueGuide(ValueDiffusion(_mk(ValueFunction, sr, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **dk)).cuda()
        c = ValueGuide(ValueDiffusion(_mk(ValueFunction, sc, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **dk)).cuda()
        return diff, g, c
    new_u = synth.synth_state_dict(spec.unet_param_shapes(H, D, dim, um), 900)
    new_r = synth.synth_state_dict(spec.value_param_shapes(H, D, dim, vm), 901)
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    noise = torch.from_numpy(fx['noise']).cuda()
    thr = torch.full((B,), float(fx['cost_threshold']), device='cuda')
    kw = fixture_kwargs(fx)
    run = lambda d, g, c: d.p_sample_loop((B, H, D), cond, verbose=False, sample_fn=fn, noise=noise, guide=g,
                                          cost_guide=c, cost_threshold=thr, **kw)
    diff, g, c = build(sd_u, sd_r, sd_c)
    first = run(diff, g, c)
    diff.model.load_state_dict({n: torch.from_numpy(v) for n, v in new_u.items()})
    g.model.model.load_state_dict({n: torch.from_numpy(v) for n, v in new_r.items()})
    second = run(diff, g, c)
    fresh = run(*build(new_u, new_r, sd_c))
    assert torch.equal(second.trajectories, fresh.trajectories) and torch.equal(second.values, fresh.values)
    assert not torch.equal(first.trajectories, second.trajectories)
    assert len(diff._samplers) == 1                  # the loop captured on the old weights is gone


def test_reference_checkpoint_route_reaches_the_kernels(tmp_path):
    """weights written in the reference Trainer's file format ({'step','model','ema'} of
    state_dicts, diffuser/utils/training.py:141-155; the value / cost models with loss_type 'value_l2', i.e. WITHOUT a
    loss_fn.weights entry, config/locomotion.py:94,137) -> checkpoint.load_into (weights_only) -> re-pack -> the fused
    loop reproduces the reference's golden chain.  No pretrained TREBI file exists offline, so the file's CONTENT is
    the golden fixture's seeded weights: the route is pinned, end-task scores are not ('parity unpinned' for those)."""
    import os
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide
    from trebi_amd import checkpoint
    fx = load_fixture('halfcheetah')
    sd_u, sd_r, sd_c, um, vm = fixture_weights(fx)
    H, obs, act, dim, T = int(fx['H']), int(fx['obs']), int(fx['act']), int(fx['dim']), int(fx['T'])
    D, B = obs + act, int(fx['B'])
    dk = dict(n_timesteps=T, clip_denoised=False, predict_epsilon=False)

    def fresh():
        return (GaussianDiffusion(TemporalUnet(H, D, 0, dim=dim, dim_mults=um), H, obs, act, loss_type='l2', **dk),
                ValueDiffusion(ValueFunction(H, D, 0, dim=dim, dim_mults=vm), H, obs, act, loss_type='value_l2', **dk),
                ValueDiffusion(ValueFunction(H, D, 0, dim=dim, dim_mults=vm), H, obs, act, loss_type='value_l2', **dk))
    # "training run": modules holding the fixture weights, saved the way Trainer.save does
    src = fresh()
    for m, sd in zip(src, (sd_u, sd_r, sd_c)):
        m.model.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()}, strict=True)
    paths = []
    for i, m in enumerate(src):
        p = os.path.join(tmp_path, f'run{i}', 'state_200000.pt')
        os.makedirs(os.path.dirname(p))
        noise_sd = {k: torch.randn_like(v) if v.dtype.is_floating_point and k.startswith('model.') else v.clone()
                    for k, v in m.state_dict().items()}                      # 'model' != 'ema': planning must take 'ema'
        torch.save({'step': 200000, 'model': noise_sd, 'ema': m.state_dict()}, p)
        paths.append(p)
    assert not any(k.startswith('loss_fn') for k in src[1].state_dict())
    # "planning run": default-initialised modules + the files
    dst = fresh()
    for m, p in zip(dst, paths):
        assert checkpoint.load_into(m, p) == 200000
    diff, guide, cguide = dst[0].cuda(), ValueGuide(dst[1]).cuda(), ValueGuide(dst[2]).cuda()
    fn, kw = _args(fx, 'constrained', guide, cguide)
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    s = diff.p_sample_loop((B, H, D), cond, verbose=False, return_chain=True, sample_fn=fn,
                           noise=torch.from_numpy(fx['noise']).cuda(), **kw)
    _check(fx, 'constrained', s)


def test_sampler_run_only_enqueues_and_is_capturable():
    """include/trebi_hip.h: after trebi_sampler_bind, trebi_sampler_run neither allocates nor synchronises -- so a
    use_graph=0 sampler can be captured into the CALLER's graph (here torch's) and replayed; and run refuses buffers
    it was not bound to."""
    import ctypes as C
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide, n_step_guided_and_constrained_p_sample as fn
    from trebi_amd import _lib
    fx = load_fixture('pendulum')
    sd_u, sd_r, sd_c, um, vm = fixture_weights(fx)
    H, obs, act, dim, T = int(fx['H']), int(fx['obs']), int(fx['act']), int(fx['dim']), int(fx['T'])
    D, B = obs + act, int(fx['B'])
    dk = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)
    diff = GaussianDiffusion(_mk(TemporalUnet, sd_u, H, D, 0, dim=dim, dim_mults=um), H, obs, act, **dk).cuda()
    guide = ValueGuide(ValueDiffusion(_mk(ValueFunction, sd_r, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **dk)).cuda()
    cguide = ValueGuide(ValueDiffusion(_mk(ValueFunction, sd_c, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **dk)).cuda()
    cond = {0: torch.from_numpy(fx['cond0']).cuda()[None].repeat(B, 1)}
    noise = torch.from_numpy(fx['noise']).cuda()
    thr = torch.full((B,), float(fx['cost_threshold']), device='cuda')
    kw = dict(guide=guide, cost_guide=cguide, cost_threshold=thr, **fixture_kwargs(fx))
    x, v, c, _ = diff.sample_raw((B, H, D), cond, sample_fn=fn, noise=noise, use_graph=False, **kw)   # creates + binds
    torch.cuda.synchronize()
    want = (x.clone(), v.clone(), c.clone())
    s = next(iter(diff._samplers.values()))
    lib = _lib.lib()

    def run():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return lib.trebi_sampler_run(s['h'], _lib.ptr(s['noise']), _lib.ptr_array(s['cond']), _lib.ptr(s['thr']),
                                     _lib.ptr(s['x']), _lib.ptr(s['values']), _lib.ptr(s['costs']), None, st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = run()
    assert rc == 0, lib.trebi_last_error()
    for t in (s['x'], s['values'], s['costs']):
        t.fill_(float('nan'))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(s['x'], want[0]) and torch.equal(s['values'], want[1]) and torch.equal(s['costs'], want[2])
    # unbound pointers are refused, not silently re-planned
    other = torch.empty_like(s['x'])
    rc = lib.trebi_sampler_run(s['h'], _lib.ptr(s['noise']), _lib.ptr_array(s['cond']), _lib.ptr(s['thr']), _lib.ptr(other),
                               _lib.ptr(s['values']), _lib.ptr(s['costs']), None, None)
    assert rc == -1 and b'trebi_sampler_bind' in lib.trebi_last_error()
