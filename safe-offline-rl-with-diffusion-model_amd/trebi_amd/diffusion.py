"""GaussianDiffusion / ValueDiffusion / ValueGuide and the sample functions, with the
reference's signatures, on the HIP library.

Reference: /root/reference/diffuser/models/diffusion.py:16-283 (sampling half),
/root/reference/diffuser/sampling/functions.py:9-37,75-137, guides.py:6-21,
helpers.py:124-145.

Two routes through the same kernels:
* `p_sample_loop` with one of the three sample functions below runs the whole
  reverse-diffusion loop inside the library (trebi_sampler_*, a captured HIP graph, no
  host synchronisation per step -- the reference syncs 6 times per step,
  diffusion.py:203, functions.py:124,126).
* any other `sample_fn` callable is driven step by step exactly like the reference does;
  the three functions below are then ordinary per-step functions built on
  trebi_value_gradients / trebi_guide_update / trebi_unet_forward / trebi_ddpm_update.

Extension to the reference API: `noise=` (a [n_timesteps+1, B, H, D] tensor) replaces the
loop's torch.randn draws, in the reference's draw order (noise[0] is x_T); parity tests need it
because CPU and GPU random streams differ.
"""
import ctypes as C
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .nets import TemporalUnet, ValueFunction

Sample = namedtuple('Sample', 'trajectories values costs chains')      # diffusion.py:16


# ----------------------------------------------------------------------------- helpers
def extract(a, t, x_shape):                                            # helpers.py:124-127
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def cosine_beta_schedule(timesteps, s=0.008, dtype=torch.float32):     # helpers.py:129-140
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.tensor(np.clip(betas, a_min=0, a_max=0.999), dtype=dtype)


def apply_conditioning(x, conditions, action_dim):                     # helpers.py:142-145
    for t, val in conditions.items():
        x[:, t, action_dim:action_dim + val.shape[-1]] = val.clone()
    return x


def make_timesteps(batch_size, i, device):                             # diffusion.py:65-67
    return torch.full((batch_size,), i, device=device, dtype=torch.long)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ctx(device):
    return _lib.ctx(device.index or 0)


def _step_desc(model, cond, scale=0.0, t_stopgrad=0, scale_grad_by_std=True, state_grad_mask=False,
               cost_grad_weight=1.0, guide_type=0):
    d = _lib.StepDesc()
    d.horizon = 0          # filled by caller
    d.transition_dim = model.transition_dim
    d.action_dim = model.action_dim
    d.observation_dim = model.observation_dim
    d.n_timesteps = model.n_timesteps
    d.scale = float(scale)
    d.t_stopgrad = int(t_stopgrad)
    d.scale_grad_by_std = int(bool(scale_grad_by_std))
    d.state_grad_mask = int(bool(state_grad_mask))
    d.cost_grad_weight = float(cost_grad_weight)
    d.guide_type = int(guide_type)
    keys = sorted(cond.keys())
    if len(keys) > 4:
        raise RuntimeError('at most 4 conditioned timesteps are supported')
    d.n_cond = len(keys)
    for i, k in enumerate(keys):
        d.cond_index[i] = int(k)
    return d, keys


def _cond_tensors(cond, keys, B, obs, device):
    out = []
    for k in keys:
        v = torch.as_tensor(cond[k], device=device, dtype=torch.float32)
        if v.dim() == 1:
            v = v[None].expand(B, -1)
        if v.shape != (B, obs):
            raise RuntimeError(f'cond[{k}] has shape {tuple(v.shape)}, expected {(B, obs)}')
        out.append(v.contiguous())
    return out


# ----------------------------------------------------------------------------- selection
def feasibility_rule(r_y, cost_y, cost_threshold):
    """functions.py:126-129 on the library (global over the batch)."""
    n = r_y.numel()
    if n == 0:
        return r_y.clone()
    thr = torch.as_tensor(cost_threshold, device=r_y.device, dtype=torch.float32).expand(n).contiguous()
    out = torch.empty_like(r_y)
    _lib.check(_lib.lib().trebi_feasibility_rule(_ctx(r_y.device), _lib.ptr(r_y.contiguous()),
                                                 _lib.ptr(cost_y.contiguous()), _lib.ptr(thr), n, _lib.ptr(out),
                                                 _stream(r_y.device)))
    return out


def argsort_descending(values):
    order = torch.empty(values.numel(), device=values.device, dtype=torch.int64)
    if values.numel() == 0:
        return order
    _lib.check(_lib.lib().trebi_argsort_desc(_ctx(values.device), _lib.ptr(values.contiguous()), values.numel(),
                                             _lib.ptr(order), _stream(values.device)))
    return order


def sort_by_values(x, values, costs):                                  # diffusion.py:44-49
    inds = argsort_descending(values)
    return x[inds], values[inds], costs[inds]


def max_by_values_in_batch(x, values, costs, sample_batch_size):       # diffusion.py:51-62
    B = x.shape[0]
    per = B // sample_batch_size
    best = torch.empty(sample_batch_size, device=x.device, dtype=torch.int64)
    _lib.check(_lib.lib().trebi_group_argmax(_ctx(x.device), _lib.ptr(values.contiguous()), sample_batch_size, per,
                                             _lib.ptr(best), _stream(x.device)))
    return x[best], values[best], costs[best]


# ----------------------------------------------------------------------------- guide
class ValueGuide(nn.Module):
    """guides.py:6-21.  `model` is a ValueDiffusion (it must expose transition_dim)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def _net(self):
        net = self.model.model if isinstance(self.model, GaussianDiffusion) else self.model
        if not isinstance(net, ValueFunction):
            raise RuntimeError('ValueGuide needs a trebi_amd ValueFunction behind it')
        return net

    def forward(self, x, cond, t, **kargs):
        output = self.model(x[:, :, 0:self.model.transition_dim], cond, t, **kargs)
        return output.squeeze(dim=-1)

    def gradients(self, x, *args, **kargs):
        """(y, d(sum y)/dx); the explicit backward-to-input replaces torch.autograd.grad (guides.py:19)."""
        t = args[1] if len(args) > 1 else kargs['t']
        y, g = self._net().value_and_input_grad(x[:, :, 0:self.model.transition_dim], t)
        if g.shape[-1] != x.shape[-1]:
            full = torch.zeros_like(x)
            full[:, :, :g.shape[-1]] = g
            g = full
        return y, g


# ----------------------------------------------------------------------------- per-step sample functions
def _noise_for(x, t, noise):
    return torch.randn_like(x) if noise is None else noise


def _ddpm_step(model, x, cond, t, noise):
    """p_mean_variance + posterior sample + re-conditioning, fused (diffusion.py:166-185, functions.py:131-137)."""
    B, H, D = x.shape
    # x_recon exactly as p_mean_variance builds it (diffusion.py:176-181): identity for predict_epsilon=False /
    # clip_denoised=False (the planning configs), otherwise the reference's two torch expressions on the net output
    x0 = model.predict_start_from_noise(x, t=t, noise=model.model(x, cond, t))
    if model.clip_denoised:
        x0 = x0.clamp(-1., 1.)
    x0 = x0.contiguous()
    d, keys = _step_desc(model, cond)
    d.horizon = H
    cts = _cond_tensors(cond, keys, B, model.observation_dim, x.device)
    out = x.contiguous().clone()
    _lib.check(_lib.lib().trebi_ddpm_update(
        _ctx(x.device), C.byref(d), _lib.ptr(out), _lib.ptr(x0), _lib.ptr(noise.contiguous()), _lib.ptr(t),
        _lib.ptr(model.posterior_log_variance_clipped), _lib.ptr(model.posterior_mean_coef1),
        _lib.ptr(model.posterior_mean_coef2), _lib.ptr_array(cts), B, _stream(x.device)))
    return out


@torch.no_grad()
def default_sample_fn(model, x, cond, t, guide=None, noise=None):      # diffusion.py:19-29
    out = _ddpm_step(model, x, cond, t, _noise_for(x, t, noise))
    values = torch.zeros(len(x), device=x.device)
    return out, values, values


def _guide_step(model, x, cond, t, r_grad, cost_grad, cost_y, thr, **kw):
    B, H, D = x.shape
    d, keys = _step_desc(model, cond, **kw)
    d.horizon = H
    cts = _cond_tensors(cond, keys, B, model.observation_dim, x.device)
    _lib.check(_lib.lib().trebi_guide_update(
        _ctx(x.device), C.byref(d), _lib.ptr(x), _lib.ptr(r_grad.contiguous()),
        _lib.ptr(None if cost_grad is None else cost_grad.contiguous()), _lib.ptr(cost_y), _lib.ptr(thr),
        _lib.ptr(t), _lib.ptr(model.posterior_log_variance_clipped), _lib.ptr_array(cts), B, _stream(x.device)))


@torch.no_grad()
def n_step_guided_p_sample(model, x, cond, t, guide, scale=0.001, t_stopgrad=0, n_guide_steps=1,
                           scale_grad_by_std=True, state_grad_mask=False, noise=None):   # functions.py:9-37
    x = x.contiguous().clone()
    for _ in range(n_guide_steps):
        y, grad = guide.gradients(x, cond, t)
        _guide_step(model, x, cond, t, grad, None, None, None, scale=scale, t_stopgrad=t_stopgrad,
                    scale_grad_by_std=scale_grad_by_std, state_grad_mask=state_grad_mask)
    out = _ddpm_step(model, x, cond, t, _noise_for(x, t, noise))
    return out, y, torch.zeros_like(y)


@torch.no_grad()
def n_step_guided_and_constrained_p_sample(model, x, cond, t, guide, scale=0.001, t_stopgrad=0, n_guide_steps=1,
                                           scale_grad_by_std=True, cost_guide=None, cost_threshold=0,
                                           cost_grad_weight=1.0, state_grad_mask=False, guide_type=0,
                                           noise=None):                                   # functions.py:75-137
    assert cost_guide is not None
    B = len(x)
    thr = torch.as_tensor(cost_threshold, device=x.device, dtype=torch.float32).expand(B).contiguous()
    x = x.contiguous().clone()
    for _ in range(n_guide_steps):
        cost_y, cost_grad = cost_guide.gradients(x, cond, t)
        r_y, r_grad = guide.gradients(x, cond, t)
        _guide_step(model, x, cond, t, r_grad, cost_grad, cost_y, thr, scale=scale, t_stopgrad=t_stopgrad,
                    scale_grad_by_std=scale_grad_by_std, state_grad_mask=state_grad_mask,
                    cost_grad_weight=cost_grad_weight, guide_type=guide_type)
    r_y = feasibility_rule(r_y, cost_y, thr)
    out = _ddpm_step(model, x, cond, t, _noise_for(x, t, noise))
    return out, r_y, cost_y


_FUSED_KINDS = {default_sample_fn: 0, n_step_guided_p_sample: 1, n_step_guided_and_constrained_p_sample: 2}


# ----------------------------------------------------------------------------- diffusion
class WeightedLoss(nn.Module):
    """helpers.py:152-167,193-201: holds the `weights` buffer (so reference checkpoints load strictly).  The
    arithmetic of the weighted loss runs inside the HIP training step (trebi_amd/training.py, trebi_train_*);
    calling the module directly is refused -- there is no torch path."""

    def __init__(self, weights, action_dim, kind):
        super().__init__()
        self.register_buffer('weights', weights)
        self.action_dim = action_dim
        self.kind = kind                     # 'l1' | 'l2'

    def forward(self, pred, targ):
        raise RuntimeError('the weighted loss is fused into the HIP training step: use GaussianDiffusion.loss() / '
                           'trebi_amd.training.TrainStep')


class ValueLoss(nn.Module):
    """helpers.py:169-191,203-211: no buffer (a reference value / cost checkpoint has no `loss_fn.weights` key).
    Value-net training is outside the hot path (SURVEY 8f covers the denoiser's training step only)."""

    def __init__(self, *args):
        super().__init__()
        self.kind = args[-1] if args else 'value_l2'

    def forward(self, pred, targ):
        raise NotImplementedError('value / cost network training is not part of this build')


def _make_loss(loss_type, weights, action_dim):
    """Losses[loss_type](weights, action_dim), helpers.py:213-218 (KeyError for an unknown type, like the dict)."""
    table = {'l1': WeightedLoss, 'l2': WeightedLoss, 'value_l1': ValueLoss, 'value_l2': ValueLoss}
    return table[loss_type](weights, action_dim, loss_type)


class GaussianDiffusion(nn.Module):
    def __init__(self, model, horizon, observation_dim, action_dim, n_timesteps=1000, loss_type='l1',
                 clip_denoised=False, predict_epsilon=True, action_weight=1.0, loss_discount=1.0, loss_weights=None,
                 predict_reward_done=False):                           # diffusion.py:73-120
        super().__init__()
        self.horizon = horizon
        self.predict_reward_done = predict_reward_done
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.transition_dim = observation_dim + action_dim + 2 * predict_reward_done
        self.model = model
        if isinstance(model, (TemporalUnet, ValueFunction)):
            model.set_max_timesteps(max(int(n_timesteps), 1))

        betas = cosine_beta_schedule(n_timesteps)
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, axis=0)
        alphas_cumprod_prev = torch.cat([torch.ones(1), alphas_cumprod[:-1]])
        self.n_timesteps = int(n_timesteps)
        self.clip_denoised = clip_denoised
        self.predict_epsilon = predict_epsilon
        rb = self.register_buffer
        rb('betas', betas)
        rb('alphas_cumprod', alphas_cumprod)
        rb('alphas_cumprod_prev', alphas_cumprod_prev)
        rb('sqrt_alphas_cumprod', torch.sqrt(alphas_cumprod))
        rb('sqrt_one_minus_alphas_cumprod', torch.sqrt(1. - alphas_cumprod))
        rb('log_one_minus_alphas_cumprod', torch.log(1. - alphas_cumprod))
        rb('sqrt_recip_alphas_cumprod', torch.sqrt(1. / alphas_cumprod))
        rb('sqrt_recipm1_alphas_cumprod', torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        rb('posterior_variance', posterior_variance)
        rb('posterior_log_variance_clipped', torch.log(torch.clamp(posterior_variance, min=1e-20)))
        # np.sqrt on a torch tensor (as the reference does, diffusion.py:113-116) is not bit-identical to
        # torch.sqrt for every entry; the tables must be, so the same mixed expression is kept
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', DeprecationWarning)       # numpy 2 on Tensor.__array_wrap__
            rb('posterior_mean_coef1', betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
            rb('posterior_mean_coef2', (1. - alphas_cumprod_prev) * np.sqrt(alphas) / (1. - alphas_cumprod))
        self.loss_type = loss_type
        self.loss_fn = _make_loss(loss_type, self.get_loss_weights(action_weight, loss_discount, loss_weights),
                                  self.action_dim)
        self._samplers = {}
        self._buffers_for = {}

    def get_loss_weights(self, action_weight, discount, weights_dict):  # diffusion.py:122-149
        self.action_weight = action_weight
        dim_weights = torch.ones(self.transition_dim, dtype=torch.float32)
        for ind, w in (weights_dict or {}).items():
            dim_weights[self.action_dim + ind] *= w
        discounts = discount ** torch.arange(self.horizon, dtype=torch.float)
        discounts = discounts / discounts.mean()
        loss_weights = torch.einsum('h,t->ht', discounts, dim_weights)
        loss_weights[0, :self.action_dim] = action_weight
        return loss_weights

    # ------------------------------------------------------------------ sampling
    def predict_start_from_noise(self, x_t, t, noise):                 # diffusion.py:153-164
        if self.predict_epsilon:
            return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t -
                    extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)
        return noise

    def q_posterior(self, x_start, x_t, t):                            # diffusion.py:166-173
        mean = (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start +
                extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, extract(self.posterior_variance, t, x_t.shape),
                extract(self.posterior_log_variance_clipped, t, x_t.shape))

    def p_mean_variance(self, x, cond, t):                             # diffusion.py:175-185
        x_recon = self.predict_start_from_noise(x, t=t, noise=self.model(x, cond, t))
        if self.clip_denoised:
            x_recon.clamp_(-1., 1.)
        return self.q_posterior(x_start=x_recon, x_t=x, t=t)

    def _fusable(self, sample_fn, kw):
        if sample_fn not in _FUSED_KINDS or self.predict_epsilon or self.clip_denoised:
            return False
        if not isinstance(self.model, TemporalUnet):
            return False
        kind = _FUSED_KINDS[sample_fn]
        if kind >= 1 and not isinstance(kw.get('guide'), ValueGuide):
            return False
        if kind == 2 and not isinstance(kw.get('cost_guide'), ValueGuide):
            return False
        return True

    def _sampler(self, kind, B, H, keys, kw, device, use_graph, skip_dead_evals):
        guide, cguide = kw.get('guide'), kw.get('cost_guide')
        rnet = guide._net() if kind >= 1 else None
        cnet = cguide._net() if kind == 2 else None
        hu = self.model.net_handle(device)
        hr = rnet.net_handle(device) if rnet is not None else None
        hc = cnet.net_handle(device) if cnet is not None else None
        gen = lambda n: None if n is None else (id(n), n.generation)
        key = (kind, B, H, tuple(keys), str(device), use_graph, skip_dead_evals,
               gen(self.model), gen(rnet), gen(cnet),
               float(kw.get('scale', 0.001)), int(kw.get('t_stopgrad', 0)), int(kw.get('n_guide_steps', 1)),
               bool(kw.get('scale_grad_by_std', True)), bool(kw.get('state_grad_mask', False)),
               float(kw.get('cost_grad_weight', 1.0)), int(kw.get('guide_type', 0)))
        if key not in self._samplers:
            d = _lib.SamplerDesc()
            sd, _ = _step_desc(self, {k: None for k in keys}, scale=key[10], t_stopgrad=key[11],
                               scale_grad_by_std=key[13], state_grad_mask=key[14], cost_grad_weight=key[15],
                               guide_type=key[16])
            sd.horizon = H
            d.step = sd
            d.kind = kind
            d.n_guide_steps = key[12]
            d.use_graph = int(use_graph)
            d.skip_dead_evals = int(skip_dead_evals)
            tabs = [getattr(self, n).detach().to('cpu', torch.float32).contiguous() for n in
                    ('posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2')]
            fp = lambda t: C.cast(t.data_ptr(), _lib.c_float_p)
            h = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(_lib.lib().trebi_sampler_create(_ctx(device), hu, hr, hc, C.byref(d), fp(tabs[0]),
                                                           fp(tabs[1]), fp(tabs[2]), B, C.byref(h)))
            if len(self._samplers) >= 8:            # keep a handful of captured loops
                self._drop_sampler(next(iter(self._samplers)))
            for n in (self.model, rnet, cnet):      # a net frees its samplers before it frees its packed weights
                if n is not None:
                    n.add_dependent(self)
            D = self.transition_dim
            T = self.n_timesteps
            self._samplers[key] = dict(
                h=h,
                noise=torch.empty(T + 1, B, H, D, device=device),
                x=torch.empty(B, H, D, device=device),
                values=torch.empty(B, device=device), costs=torch.empty(B, device=device),
                thr=torch.zeros(B, device=device),
                cond=[torch.empty(B, self.observation_dim, device=device) for _ in keys],
                chain=None)
        return self._samplers[key]

    def _drop_sampler(self, key):
        s = self._samplers.pop(key, None)
        if s is not None:
            torch.cuda.synchronize()                # a replay of its graph may still be in flight
            _lib.lib().trebi_sampler_destroy(s['h'])

    def drop_samplers_of(self, net):
        """Called by a network right before it frees its packed weights (new parameters were loaded): every captured
        loop built on that handle holds raw device pointers into them."""
        for key in [k for k in self._samplers if any(g is not None and g[0] == id(net) for g in k[7:10])]:
            self._drop_sampler(key)

    def __del__(self):
        try:
            for key in list(self._samplers):
                self._drop_sampler(key)
        except Exception:
            pass

    @torch.no_grad()
    def sample_raw(self, shape, cond, sample_fn=default_sample_fn, return_chain=False, noise=None,
                   use_graph=True, skip_dead_evals=False, profile=False, prepared=False, **sample_kwargs):
        """The fused loop without the final selection: returns (x[B,H,D], values_raw[B], costs[B],
        chain|None) for this rank's candidates.  ShardedPlanner all-gathers values/costs.
        The returned tensors are the sampler's persistent buffers: the next call with the same batch size and
        arguments overwrites them (clone what must outlive it).  prepared=True: the caller has already filled the
        sampler's cond / threshold buffers on the device (GuidedPolicy does, with trebi_policy_prepare); `cond` then
        only names the conditioned timesteps."""
        if self.predict_epsilon or self.clip_denoised:
            raise RuntimeError('the fused loop computes x0-parameterised, unclipped steps only (predict_epsilon=False, '
                               'clip_denoised=False, config/locomotion.py:30,39); use p_sample_loop, which runs such '
                               'a model step by step through the reference formulas')
        device = self.betas.device
        B, H, D = shape
        if B == 0:                                  # empty batch: nothing to plan (torch would return empty tensors)
            e = torch.empty(0, device=device)
            chain = torch.empty(self.n_timesteps + 1, 0, H, D, device=device) if return_chain else None
            return torch.empty(0, H, D, device=device), e, e.clone(), chain
        kind = _FUSED_KINDS[sample_fn]
        keys = sorted(cond.keys())
        s = self._sampler(kind, B, H, keys, sample_kwargs, device, use_graph, skip_dead_evals)
        if noise is None:
            s['noise'].normal_()                    # one generator call instead of T+1 (diffusion.py:192, functions.py:134)
        else:
            s['noise'].copy_(torch.as_tensor(noise, device=device).reshape(s['noise'].shape))
        if not prepared:
            for buf, v in zip(s['cond'], _cond_tensors(cond, keys, B, self.observation_dim, device)):
                buf.copy_(v)
            if kind == 2:
                thr = sample_kwargs.get('cost_threshold', 0)
                s['thr'].copy_(torch.as_tensor(thr, device=device, dtype=torch.float32).expand(B))
        if return_chain and s['chain'] is None:
            s['chain'] = torch.empty(self.n_timesteps + 1, B, H, D, device=device)
        chain = s['chain'] if return_chain else None
        bound = (s['noise'].data_ptr(), None if chain is None else chain.data_ptr())
        if s.get('bound') != bound:                 # set-up for this set of buffers: launch plans + graph capture
            with torch.cuda.device(device):
                _lib.check(_lib.lib().trebi_sampler_bind(s['h'], _lib.ptr(s['noise']), _lib.ptr_array(s['cond']),
                                                         _lib.ptr(s['thr']), _lib.ptr(s['x']), _lib.ptr(s['values']),
                                                         _lib.ptr(s['costs']), _lib.ptr(chain)))
            s['bound'] = bound
        if profile:
            _lib.check(_lib.lib().trebi_sampler_set_profiling(s['h'], 1))
        _lib.check(_lib.lib().trebi_sampler_run(s['h'], _lib.ptr(s['noise']), _lib.ptr_array(s['cond']),
                                                _lib.ptr(s['thr']), _lib.ptr(s['x']), _lib.ptr(s['values']),
                                                _lib.ptr(s['costs']), _lib.ptr(chain), _stream(device)))
        if profile:
            _lib.check(_lib.lib().trebi_sampler_set_profiling(s['h'], 0))
            us, n, fl, loop = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(_lib.lib().trebi_sampler_get_profile(s['h'], C.byref(us), C.byref(n), C.byref(fl),
                                                            C.byref(loop)))
            self.last_profile = dict(gemm_us=us.value, gemm_launches=n.value, gemm_flops=fl.value,
                                     loop_us=loop.value)
        return s['x'], s['values'], s['costs'], chain

    @torch.no_grad()
    def p_sample_loop(self, shape, cond, verbose=True, return_chain=False, sample_fn=default_sample_fn,
                      sample_batch_size=None, noise=None, **sample_kwargs):              # diffusion.py:187-213
        device = self.betas.device
        batch_size = shape[0]
        if self._fusable(sample_fn, sample_kwargs):
            x, values, costs, chain = self.sample_raw(shape, cond, sample_fn=sample_fn, return_chain=return_chain,
                                                      noise=noise, **sample_kwargs)
            if _FUSED_KINDS[sample_fn] == 2:
                values = feasibility_rule(values, costs, sample_kwargs.get('cost_threshold', 0))
            if return_chain:
                chain = chain.permute(1, 0, 2, 3).clone()                                # [B, T+1, H, D]
        else:
            draws = iter(noise) if noise is not None else None
            x = next(draws).to(device).clone() if draws else torch.randn(shape, device=device)
            x = apply_conditioning(x, cond, self.action_dim)
            chain = [x] if return_chain else None
            for i in reversed(range(0, self.n_timesteps)):
                t = make_timesteps(batch_size, i, device)
                extra = dict(noise=next(draws).to(device)) if draws else {}
                x, values, costs = sample_fn(self, x, cond, t, **sample_kwargs, **extra)
                x = apply_conditioning(x, cond, self.action_dim)
                if return_chain:
                    chain.append(x)
            if return_chain:
                chain = torch.stack(chain, dim=1)
        if sample_batch_size is None:
            x, values, costs = sort_by_values(x, values, costs)
        else:
            x, values, costs = max_by_values_in_batch(x, values, costs, sample_batch_size)
        return Sample(x, values, costs, chain)

    @torch.no_grad()
    def conditional_sample(self, cond, horizon=None, **sample_kwargs):                    # diffusion.py:215-225
        batch_size = len(cond[0])
        horizon = horizon or self.horizon
        shape = (batch_size, horizon, self.transition_dim)
        return self.p_sample_loop(shape, cond, **sample_kwargs)

    # ------------------------------------------------------------------ training
    def train_engine(self, max_batch=None, ema_model=None):
        """The library object behind loss(): flat parameter / gradient buffers + the fused forward-loss-backward
        (trebi_amd/training.py).  One per model: the TemporalUnet's parameters become views of its flat buffer."""
        from .training import TrainEngine
        eng = getattr(self, '_engine', None)
        if eng is None:
            if max_batch is None:
                raise RuntimeError('no training engine yet: call train_engine(max_batch) or build a training.Trainer')
            eng = self._engine = TrainEngine(self, max_batch, ema_model=ema_model)
        elif max_batch is not None and max_batch > eng.max_batch:
            raise RuntimeError(f'the training engine was sized for batches of {eng.max_batch}, got {max_batch}')
        return eng

    def p_losses(self, x_start, cond, t, noise=None, grad_scale=1.0):                     # diffusion.py:240-256
        """Returns (loss, {'a0_loss': ...}) like the reference -- and, because forward, loss and backward are ONE
        fused library call here, ALSO adds grad_scale * dloss/dparams to the engine's flat gradient (what the
        reference does later in loss.backward(), training.py:115-116)."""
        eng = self.train_engine(len(x_start))
        if noise is None:
            noise = torch.randn_like(x_start)
        out = eng.accumulate(x_start, cond, t, noise, grad_scale=grad_scale)
        return out[0], {'a0_loss': out[1]}

    def loss(self, x, *args, **kw):                                                       # diffusion.py:258-261
        t = torch.randint(0, self.n_timesteps, (len(x),), device=x.device).long()
        return self.p_losses(x, *args, t, **kw)

    def forward(self, cond, *args, **kwargs):                                             # diffusion.py:263-264
        return self.conditional_sample(cond, *args, **kwargs)


class ValueDiffusion(GaussianDiffusion):                                                  # diffusion.py:269-283
    def forward(self, x, cond, t, policy_step=True):
        return self.model(x, cond, t)
