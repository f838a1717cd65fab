"""Generate tests/golden/*.npz by running the REFERENCE on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference cannot travel to the GPU box, so what is committed is data only:
inputs (noise, conditions, thresholds, seeds) and the reference's outputs.  The
weights are NOT stored; they are regenerated from trebi_amd.synth (numpy PCG64)
from the seeds recorded in each fixture, and this script proves the regenerated
state_dict loads into the reference modules with strict=True.

How the reference is imported: `import diffuser` fails on
absent simulators (bullet_safety_gym, mujoco_py, ...), so empty stub packages are
registered for `diffuser`, `diffuser.models`, `diffuser.sampling` and a stub
`diffuser.utils` exposing the three names the five hot-path files touch; the
five files are then loaded unmodified from their paths.  Noise is injected by
temporarily replacing torch.randn / torch.randn_like during the call.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))

from trebi_amd import spec, synth  # noqa: E402


def load_reference():
    def stub(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    stub('diffuser')
    stub('diffuser.models')
    stub('diffuser.sampling')
    u = stub('diffuser.utils')
    u.to_np = lambda x: x.detach().cpu().numpy() if torch.is_tensor(x) else x

    class Silent:
        def __init__(self, *a, **k): pass
        def update(self, *a, **k): pass
        def stamp(self, *a, **k): pass
    u.Silent = Silent
    u.Progress = Silent
    sys.modules['diffuser'].utils = u

    def load(modname, rel):
        sp = importlib.util.spec_from_file_location(modname, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(sp)
        sys.modules[modname] = m
        sp.loader.exec_module(m)
        return m

    helpers = load('diffuser.models.helpers', 'diffuser/models/helpers.py')
    temporal = load('diffuser.models.temporal', 'diffuser/models/temporal.py')
    diffusion = load('diffuser.models.diffusion', 'diffuser/models/diffusion.py')
    guides = load('diffuser.sampling.guides', 'diffuser/sampling/guides.py')
    functions = load('diffuser.sampling.functions', 'diffuser/sampling/functions.py')
    return helpers, temporal, diffusion, guides, functions


class inject_noise:
    """Feed torch.randn(shape) then every torch.randn_like(x) from a pre-drawn tensor."""

    def __init__(self, noise):
        self.noise = noise
        self.i = 0

    def _next(self, *a, **k):
        n = self.noise[self.i].clone()
        self.i += 1
        return n

    def __enter__(self):
        self.saved = (torch.randn, torch.randn_like)
        torch.randn = self._next
        torch.randn_like = self._next
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self.saved


def to_torch_sd(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


CONFIGS = {
    # name: obs, act, unet_mults, value_mults, B, sampler kwargs  
    'pendulum': dict(obs=3, act=1, unet_mults=(1, 2, 4, 8), B=4,
                     kw=dict(scale=0.1, t_stopgrad=2, n_guide_steps=2, cost_grad_weight=20.0, guide_type=0)),
    'halfcheetah': dict(obs=17, act=6, unet_mults=(1, 2, 4, 8), B=4,
                        kw=dict(scale=0.001, t_stopgrad=4, n_guide_steps=2, cost_grad_weight=5.0, guide_type=0)),
    'hopper': dict(obs=11, act=3, unet_mults=(1, 2, 4, 8), B=4,
                   kw=dict(scale=0.1, t_stopgrad=2, n_guide_steps=2, cost_grad_weight=5.0, guide_type=1)),
    'walker148': dict(obs=17, act=6, unet_mults=(1, 4, 8), B=3,
                      kw=dict(scale=0.1, t_stopgrad=2, n_guide_steps=1, cost_grad_weight=5.0, guide_type=0,
                              state_grad_mask=True)),
    # the reference's in-repo halfcheetah shape (config/locomotion.py:295-314): horizon 4, dim_mults (1,4,8) for
    # all three nets; the TemporalUnet then skips its last Upsample1d (temporal.py:95,109), SURVEY.md 8f rank 4
    'halfcheetah_h4': dict(obs=17, act=6, unet_mults=(1, 4, 8), value_mults=(1, 4, 8), H=4, B=4,
                           kw=dict(scale=0.001, t_stopgrad=4, n_guide_steps=2, cost_grad_weight=5.0, guide_type=0)),
    # long-horizon toy run (test_cost_toyenv.sh:5 plans with horizon 128), SURVEY.md 8f rank 4.  The reference's
    # ValueFunction only runs when its head sees one position (temporal.py:191 halves the horizon one time fewer
    # than forward :225-228 downsamples), i.e. horizon <= 2^(len(dim_mults)+1): six levels for horizon 128
    'pendulum_h128': dict(obs=3, act=1, unet_mults=(1, 2, 4, 8), value_mults=(1, 2, 4, 8, 8, 8), H=128, B=2,
                          kw=dict(scale=0.1, t_stopgrad=2, n_guide_steps=2, cost_grad_weight=20.0, guide_type=0)),
}
H, T, DIM = 32, 20, 32
VMULTS = (1, 2, 4, 8)
SEED_UNET, SEED_REWARD, SEED_COST, SEED_NOISE, SEED_COND = 100, 101, 102, 7, 8


def main():
    helpers, temporal, diffusion, guides, functions = load_reference()
    torch.set_num_threads(8)

    # ---- schedule tables (diffusion.py:85-116) ----------------------------
    dummy = torch.nn.Linear(1, 1)
    gd = diffusion.GaussianDiffusion(dummy, H, 17, 6, n_timesteps=T, loss_type='l2',
                                     clip_denoised=False, predict_epsilon=False)
    sched = {k: v.numpy().copy() for k, v in gd.state_dict().items()
             if k.split('.')[0] in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
                                    'sqrt_one_minus_alphas_cumprod', 'posterior_variance',
                                    'posterior_log_variance_clipped', 'posterior_mean_coef1',
                                    'posterior_mean_coef2')}
    np.savez_compressed(os.path.join(HERE, 'schedule_T20.npz'), **sched)

    only = sys.argv[1:]                                       # optional: generate only the named configs
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        write_config(name, cfg, temporal, diffusion, guides, functions, helpers)


def write_config(name, cfg, temporal, diffusion, guides, functions, helpers):
    if True:
        H, VMULTS = cfg.get('H', globals()['H']), cfg.get('value_mults', globals()['VMULTS'])
        obs, act, B = cfg['obs'], cfg['act'], cfg['B']
        D = obs + act
        out = dict(obs=obs, act=act, B=B, H=H, T=T, dim=DIM, unet_mults=np.array(cfg['unet_mults']),
                   value_mults=np.array(VMULTS),
                   seeds=np.array([SEED_UNET, SEED_REWARD, SEED_COST, SEED_NOISE, SEED_COND]))

        unet = temporal.TemporalUnet(H, D, 0, dim=DIM, dim_mults=cfg['unet_mults'])
        rnet = temporal.ValueFunction(H, D, 0, dim=DIM, dim_mults=VMULTS)
        cnet = temporal.ValueFunction(H, D, 0, dim=DIM, dim_mults=VMULTS)
        sd_u = synth.synth_state_dict(spec.unet_param_shapes(H, D, DIM, cfg['unet_mults']), SEED_UNET)
        sd_r = synth.synth_state_dict(spec.value_param_shapes(H, D, DIM, VMULTS), SEED_REWARD)
        sd_c = synth.synth_state_dict(spec.value_param_shapes(H, D, DIM, VMULTS), SEED_COST)
        unet.load_state_dict(to_torch_sd(sd_u), strict=True)       # proves names + shapes == reference
        rnet.load_state_dict(to_torch_sd(sd_r), strict=True)
        cnet.load_state_dict(to_torch_sd(sd_c), strict=True)
        assert list(unet.state_dict().keys()) == list(sd_u.keys())
        assert list(rnet.state_dict().keys()) == list(sd_r.keys())

        mk = lambda m, cls: cls(m, H, obs, act, n_timesteps=T, loss_type='l2', clip_denoised=False,
                                predict_epsilon=False)
        diff = mk(unet, diffusion.GaussianDiffusion)
        guide = guides.ValueGuide(mk(rnet, diffusion.ValueDiffusion))
        cguide = guides.ValueGuide(mk(cnet, diffusion.ValueDiffusion))

        noise = torch.from_numpy(synth.synth_normal(SEED_NOISE, name, (T + 1, B, H, D)))
        cond0 = torch.from_numpy(synth.synth_normal(SEED_COND, name, (obs,)))
        cond = {0: cond0[None].repeat(B, 1)}                          # policies.py:77-82 (same obs for all)
        out['noise'] = noise.numpy()
        out['cond0'] = cond0.numpy()

        # ---- single-network vectors at t in {0,5,19} ----------------------
        xin = helpers.apply_conditioning(noise[0].clone(), cond, act)
        out['net_x'] = xin.numpy().copy()
        for tt in (0, 5, 19):
            t = torch.full((B,), tt, dtype=torch.long)
            with torch.no_grad():
                out[f'unet_out_t{tt}'] = unet(xin, cond, t).numpy()
            for tag, g in (('reward', guide), ('cost', cguide)):
                y, grad = g.gradients(xin.clone(), cond, t)
                out[f'{tag}_y_t{tt}'] = y.detach().numpy()
                out[f'{tag}_grad_t{tt}'] = grad.detach().numpy()
        # a few UNet intermediates at t=5, via forward hooks
        taps = {}
        hooks = [
            unet.downs[0][0].register_forward_hook(lambda m, i, o: taps.__setitem__('down0_block0', o.detach().numpy())),
            unet.mid_block2.register_forward_hook(lambda m, i, o: taps.__setitem__('mid_out', o.detach().numpy())),
        ]
        with torch.no_grad():
            unet(xin, cond, torch.full((B,), 5, dtype=torch.long))
        for hk in hooks:
            hk.remove()
        out['unet_tap_down0_block0_t5'] = taps['down0_block0']
        out['unet_tap_mid_out_t5'] = taps['mid_out']

        # ---- threshold that splits the batch ------
        with torch.no_grad():
            c19 = cguide(xin, cond, torch.full((B,), T - 1, dtype=torch.long)).numpy()
        thr = float(np.median(c19))
        out['cost_threshold'] = np.float32(thr)
        thr_t = torch.full((B,), thr, dtype=torch.float32)

        # ---- full loops ----------------------------------------------------
        kw = dict(cfg['kw'])
        def run(tag, sample_fn, **skw):
            with inject_noise(noise) as inj:
                s = diff.p_sample_loop((B, H, D), {0: cond[0].clone()}, verbose=False, return_chain=True,
                                       sample_fn=sample_fn, **skw)
                assert inj.i == T + 1
            out[f'{tag}_traj'] = s.trajectories.numpy()
            out[f'{tag}_values'] = s.values.detach().numpy()
            out[f'{tag}_costs'] = s.costs.detach().numpy()
            out[f'{tag}_chain'] = s.chains.numpy()
            print(name, tag, 'values', s.values.detach().numpy()[:4], 'costs', s.costs.detach().numpy()[:4])

        import io, contextlib
        buf = io.StringIO()                                   # functions.py:124 prints every step
        with contextlib.redirect_stdout(buf):
            run('constrained', functions.n_step_guided_and_constrained_p_sample, guide=guide, cost_guide=cguide,
                cost_threshold=thr_t, **kw)
            kw2 = dict(kw); kw2['guide_type'] = 1 - kw['guide_type']
            run('constrained_gt', functions.n_step_guided_and_constrained_p_sample, guide=guide,
                cost_guide=cguide, cost_threshold=thr_t, **kw2)
            # all-infeasible branch of the rule (functions.py:126-127): threshold below every cost
            run('constrained_infeasible', functions.n_step_guided_and_constrained_p_sample, guide=guide,
                cost_guide=cguide, cost_threshold=torch.full((B,), -1e3), **kw)
            gkw = {k: v for k, v in kw.items() if k not in ('cost_grad_weight', 'guide_type')}
            run('guided', functions.n_step_guided_p_sample, guide=guide, **gkw)
            run('default', diffusion.default_sample_fn)
        sys.stdout.write(''.join(l + '\n' for l in buf.getvalue().splitlines() if l.startswith(name)))
        for k, v in kw.items():
            out[f'kw_{k}'] = np.array(v)
        np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
        print('wrote', name, sum(v.nbytes for v in out.values() if hasattr(v, 'nbytes')) // 1024, 'KiB raw')


if __name__ == '__main__':
    main()
