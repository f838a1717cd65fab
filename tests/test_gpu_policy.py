"""GuidedPolicy (policies.py:11-82) on the HIP loop: one control step reproduces the reference's golden plan
(same weights, same injected noise), through normalisation and un-normalisation.  Tolerance 5e-4 relative as in
test_gpu_sampling.py."""
import numpy as np
import pytest
import torch

from _common import fixture_kwargs, fixture_weights, load_fixture, rel_err

pytestmark = pytest.mark.gpu
TOL = 5e-4


def _mk(cls, sd, *a, **k):
    net = cls(*a, **k)
    net.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()}, strict=True)
    return net


@pytest.fixture(scope='module')
def setup():
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide
    from trebi_amd.policy import GaussianNormalizer
    fx = load_fixture('halfcheetah')
    sd_u, sd_r, sd_c, um, vm = fixture_weights(fx)
    H, obs, act, dim, T = int(fx['H']), int(fx['obs']), int(fx['act']), int(fx['dim']), int(fx['T'])
    D = obs + act
    kw = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)
    diff = GaussianDiffusion(_mk(TemporalUnet, sd_u, H, D, 0, dim=dim, dim_mults=um), H, obs, act, **kw).cuda()
    guide = ValueGuide(ValueDiffusion(_mk(ValueFunction, sd_r, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **kw)).cuda()
    cguide = ValueGuide(ValueDiffusion(_mk(ValueFunction, sd_c, H, D, 0, dim=dim, dim_mults=vm), H, obs, act, **kw)).cuda()
    rng = np.random.default_rng(3)
    norm = GaussianNormalizer({'observations': (rng.standard_normal(obs), 0.5 + rng.random(obs)),
                               'actions': (rng.standard_normal(act), 0.5 + rng.random(act))})
    return fx, diff, guide, cguide, norm


def test_single_observation_step_matches_golden(setup):
    from diffuser.sampling import GuidedPolicy, n_step_guided_and_constrained_p_sample as fn
    fx, diff, guide, cguide, norm = setup
    B, act, obs = int(fx['B']), int(fx['act']), int(fx['obs'])
    noise = torch.from_numpy(fx['noise']).cuda()
    policy = GuidedPolicy(guide, diff, norm, [], sample_fn=fn, cost_guide=cguide,
                          cost_threshold=float(fx['cost_threshold']), noise=noise, **fixture_kwargs(fx))
    raw_obs = norm.unnormalize(fx['cond0'], 'observations')          # the environment's observation
    action, traj = policy({0: raw_obs}, batch_size=B, verbose=False)
    g = fx['constrained_traj']                                       # sorted golden plans [B, H, D]
    assert action.shape == (act,)
    assert rel_err(action, norm.unnormalize(g[0, 0, :act], 'actions')) < TOL
    assert rel_err(traj.actions, norm.unnormalize(g[:, :, :act], 'actions')) < TOL
    assert rel_err(traj.observations, norm.unnormalize(g[:, :, act:act + obs], 'observations')) < TOL
    np.testing.assert_allclose(traj.costs.cpu().numpy(), fx['constrained_costs'], rtol=0, atol=TOL)
    assert traj.rewards.shape == (B, g.shape[1], 1) and not traj.terminals.any()
    # a threshold passed at call time replaces the stored one (policies.py:32), here: nothing is feasible
    action2, traj2 = policy({0: raw_obs}, batch_size=B, verbose=False, cost_threshold=-1e3)
    assert rel_err(traj2.actions, norm.unnormalize(fx['constrained_infeasible_traj'][:, :, :act], 'actions')) < TOL


def test_batched_observations_pick_per_group_maximum(setup):
    """conditions[0] of shape [n_obs, obs]: the reference repeats with the repeat index outermost and then takes the
    arg-max of `values.reshape(n_obs, -1)` (diffusion.py:51-62) -- mirrored as is."""
    from diffuser.sampling import GuidedPolicy, n_step_guided_p_sample as fn
    fx, diff, guide, cguide, norm = setup
    act, obs, H = int(fx['act']), int(fx['obs']), int(fx['H'])
    n_obs, rep = 2, 3
    rng = np.random.default_rng(5)
    raw = rng.standard_normal((n_obs, obs)).astype(np.float32)
    noise = torch.randn(int(fx['T']) + 1, n_obs * rep, H, act + obs, device='cuda',
                        generator=torch.Generator('cuda').manual_seed(0))
    gkw = {k: v for k, v in fixture_kwargs(fx).items() if k not in ('cost_grad_weight', 'guide_type')}
    policy = GuidedPolicy(guide, diff, norm, [], sample_fn=fn, noise=noise, **gkw)
    action, traj = policy({0: raw}, batch_size=rep, verbose=False)
    assert action.shape == (n_obs, act) and traj.actions.shape == (n_obs, H, act)
    # the same loop without the selection
    cond = {0: torch.as_tensor(norm.normalize(raw, 'observations'), dtype=torch.float32, device='cuda').repeat(rep, 1)}
    x, v, c, _ = diff.sample_raw((n_obs * rep, H, act + obs), cond, sample_fn=fn, guide=guide, noise=noise, **gkw)
    best = v.reshape(n_obs, rep).argmax(dim=1)
    sel = x.reshape(n_obs, rep, H, -1)[torch.arange(n_obs), best]
    assert rel_err(traj.actions, norm.unnormalize(sel[:, :, :act].cpu().numpy(), 'actions')) < 1e-6
    assert torch.equal(traj.values, v.reshape(n_obs, rep).max(dim=1).values)


def test_one_control_step_moves_under_1kb_to_the_host(setup, monkeypatch):
    """at 8192 candidates the reference copies 24 MB to the host per control step to read one
    action; here only the action leaves the device unless the caller reads the plans.  Every device->host route a
    torch tensor has is counted."""
    from diffuser.sampling import GuidedPolicy, n_step_guided_and_constrained_p_sample as fn
    fx, diff, guide, cguide, norm = setup
    act, obs = int(fx['act']), int(fx['obs'])
    B = 8192
    policy = GuidedPolicy(guide, diff, norm, [], sample_fn=fn, cost_guide=cguide,
                          cost_threshold=float(fx['cost_threshold']), **fixture_kwargs(fx))
    raw_obs = norm.unnormalize(fx['cond0'], 'observations')
    policy({0: raw_obs}, batch_size=B, verbose=False)                 # first call: builds the captured loop
    moved = [0]

    def counted(name):
        orig = getattr(torch.Tensor, name)

        def wrapper(self, *a, **k):
            to_host = self.is_cuda and (name != 'to' or any(str(v) == 'cpu' for v in list(a) + list(k.values())))
            if to_host:
                moved[0] += self.numel() * self.element_size()
            return orig(self, *a, **k)
        monkeypatch.setattr(torch.Tensor, name, wrapper)
    for name in ('cpu', 'item', 'tolist', 'numpy', 'to', '__array__'):
        counted(name)
    action, traj = policy({0: raw_obs}, batch_size=B, verbose=False)
    torch.cuda.synchronize()
    assert action.shape == (act,) and np.isfinite(action).all()
    assert 0 < moved[0] < 1024, moved[0]
    # the plans are still there for a caller that wants them (and reading them is what moves the bytes)
    assert traj.actions.shape == (B, int(fx['H']), act)
    assert moved[0] >= B * int(fx['H']) * (act + obs) * 4
    assert rel_err(action, traj.actions[0, 0]) < 1e-6


def test_host_normalizer_and_custom_sample_fn_routes(setup):
    """A normalizer without Gaussian statistics keeps its own arithmetic (on the observation and the selected action
    only), and a custom sample_fn falls back to the reference's step-by-step flow: both give the device path's plan."""
    from diffuser.sampling import GuidedPolicy, n_step_guided_and_constrained_p_sample as fn
    fx, diff, guide, cguide, norm = setup
    B = int(fx['B'])

    class Opaque:                                                    # same arithmetic, no statistics to take to the device
        def normalize(self, x, key):
            return norm.normalize(x, key)

        def unnormalize(self, x, key):
            return norm.unnormalize(x, key)
    noise = torch.from_numpy(fx['noise']).cuda()
    kw = dict(cost_guide=cguide, cost_threshold=float(fx['cost_threshold']), noise=noise, **fixture_kwargs(fx))
    raw_obs = norm.unnormalize(fx['cond0'], 'observations')
    a0, t0 = GuidedPolicy(guide, diff, norm, [], sample_fn=fn, **kw)({0: raw_obs}, batch_size=B, verbose=False)
    a1, t1 = GuidedPolicy(guide, diff, Opaque(), [], sample_fn=fn, **kw)({0: raw_obs}, batch_size=B, verbose=False)
    a2, t2 = GuidedPolicy(guide, diff, norm, [], sample_fn=lambda *a, **k: fn(*a, **k), **kw)({0: raw_obs}, batch_size=B,
                                                                                              verbose=False)
    assert rel_err(a1, a0) < 1e-5 and rel_err(a2, a0) < TOL
    assert rel_err(t1.actions, t0.actions) < 1e-5 and rel_err(t2.observations, t0.observations) < TOL
