"""Diagnostics: per-layer times of ONE profiled planning call (TREBI_PROFILE_DUMP) for a BASELINE config at its per-GPU size.
usage: python tools/layer_profile.py <2|3|1> <out.tsv> [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which, out = int(sys.argv[1]), sys.argv[2]
prec = sys.argv[3] if len(sys.argv) > 3 else 'f32'
os.environ['TREBI_PROFILE_DUMP'] = out
import numpy as np, torch
import bench
from trebi_amd import spec, synth
from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
from diffuser.sampling import ValueGuide, n_step_guided_and_constrained_p_sample as fn
H, T, DIM = bench.H, bench.T, bench.DIM
obs, act, um, vm, B = {4: (17, 6, (1, 2, 4, 8), (1, 2, 4, 8), 8192), 1: (17, 6, (1, 2, 4, 8), (1, 2, 4, 8), 64), 2: (11, 3, (1, 2, 4, 8), (1, 2, 4, 8), 512),
                       3: (17, 6, (1, 4, 8), (1, 2, 4, 8), 8192)}[which]
if len(sys.argv) > 4:
    B = int(sys.argv[4])
D = obs + act
dev = torch.device('cuda:0')
def mk(cls, shapes, seed, mults):
    net = cls(H, D, 0, dim=DIM, dim_mults=mults)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()})
    return net
dkw = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)
unet = mk(TemporalUnet, spec.unet_param_shapes(H, D, DIM, um), 100, um)
diff = GaussianDiffusion(unet, H, obs, act, **dkw).to(dev)
guide = ValueGuide(ValueDiffusion(mk(ValueFunction, spec.value_param_shapes(H, D, DIM, vm), 101, vm), H, obs, act, **dkw)).to(dev)
cguide = ValueGuide(ValueDiffusion(mk(ValueFunction, spec.value_param_shapes(H, D, DIM, vm), 102, vm), H, obs, act, **dkw)).to(dev)
if prec != 'f32':
    for n in (unet, guide._net(), cguide._net()):
        n.set_precision(prec)
o = torch.randn(obs, device=dev)
kw = dict(scale=0.1, t_stopgrad=2, n_guide_steps=2, cost_grad_weight=5.0, guide_type=0, guide=guide, cost_guide=cguide)
if os.environ.get('LP_UNET_ONLY') == '1':          # the denoiser alone: the unguided loop (diffusion.py:19-29)
    from trebi_amd.diffusion import default_sample_fn as fn
    kw = {}
for profile in (False, True):
    if kw:
        diff.sample_raw((B, H, D), {0: o[None].expand(B, -1)}, sample_fn=fn, cost_threshold=1e9, profile=profile, **kw)
    else:
        diff.sample_raw((B, H, D), {0: o[None].expand(B, -1)}, sample_fn=fn, profile=profile)
    torch.cuda.synchronize()
p = diff.last_profile
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in p.items()})
rows = [l.rstrip('\n').split('\t') for l in open(out)][1:]
rows.sort(key=lambda r: -float(r[1]))
tot = sum(float(r[1]) for r in rows)
print(f'total {tot / 1e3:.2f} ms over {len(rows)} distinct launches')
for r in rows[:45]:
    print(f'{float(r[1]) / 1e3:8.2f} ms {100 * float(r[1]) / tot:5.1f}%  n={r[0]:>4} avg {float(r[2]):8.1f} us {float(r[3]):6.1f} TF  {r[4]}')
