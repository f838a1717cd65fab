"""ORACLE -- test infrastructure, not product code.

A plain numpy (float32) restatement of TREBI's cost-guided diffusion-planning hot
path, written from the reference's source.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the shipped path
(safe-offline-rl-with-diffusion-model_amd/) never does and fails loudly without
its HIP library.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
against vectors produced by the reference itself (tests/golden/make_golden.py
imports /root/reference/diffuser/{models,sampling}/*.py on CPU and writes
tests/golden/*.npz; the reference has no tests or fixtures of its own, SURVEY.md
section 4).

Each function cites the reference file:line (relative to /root/reference) it
follows.  Layout is the reference's: activations are [batch, channels, length]
inside the networks and [batch, horizon, transition] at the boundary.  The
arithmetic primitives (conv1d, group_norm, mish, linear) restate the PyTorch
operator definitions the reference delegates to (torch.nn.Conv1d /
ConvTranspose1d / GroupNorm(eps=1e-5, biased variance) / Mish / Linear); the
backward-to-input functions restate what torch.autograd.grad computes for them.
"""
import math
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# primitives (PyTorch operator semantics)
# --------------------------------------------------------------------------- #

def softplus(x):
    # torch.nn.functional.softplus(beta=1, threshold=20), used by Mish
    x = np.asarray(x, F32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, F32(20))))).astype(F32)


def mish(x):
    # torch.nn.Mish: x * tanh(softplus(x))            (helpers.py:59, temporal.py:29,73)
    return (x * np.tanh(softplus(x))).astype(F32)


def mish_grad(x):
    # d/dx [x tanh(sp(x))] = tanh(sp) + x * sigmoid(x) * (1 - tanh(sp)^2)
    sp = softplus(x)
    th = np.tanh(sp)
    sig = F32(1) / (F32(1) + np.exp(-x))
    return (th + x * sig * (F32(1) - th * th)).astype(F32)


def linear(x, w, b):
    return (x @ w.T + b).astype(F32)


def conv1d(x, w, b, stride=1, pad=0):
    """torch.nn.Conv1d forward.  x [B,Ci,L], w [Co,Ci,k], b [Co] -> [B,Co,Lout].
    y[b,:,lo] = sum_tap w[:,:,tap] @ xpad[b,:,lo*stride+tap]  -- one sgemm per tap."""
    Co, Ci, k = w.shape
    B, _, L = x.shape
    Lout = (L + 2 * pad - k) // stride + 1
    xp = np.zeros((B, L + 2 * pad, Ci), F32)                # channels-last copy so each tap is a GEMM
    xp[:, pad:pad + L, :] = x.transpose(0, 2, 1)
    y = np.zeros((B * Lout, Co), F32)
    wt = np.ascontiguousarray(w.transpose(2, 1, 0))          # [k, Ci, Co], contiguous so BLAS is used
    for tap in range(k):
        rows = np.ascontiguousarray(xp[:, tap:tap + (Lout - 1) * stride + 1:stride, :]).reshape(B * Lout, Ci)
        y += rows @ wt[tap]
    y = y.reshape(B, Lout, Co)
    if b is not None:
        y += b
    return np.ascontiguousarray(y.transpose(0, 2, 1), dtype=F32)


def conv1d_input_grad(dy, w, L_in, stride=1, pad=0):
    """d(sum)/dx of conv1d: scatter-add of dy[b,:,lo] @ w[:,:,tap] to x position lo*stride+tap-pad."""
    Co, Ci, k = w.shape
    B, _, Lout = dy.shape
    dyl = np.ascontiguousarray(dy.transpose(0, 2, 1)).reshape(B * Lout, Co)
    wt = np.ascontiguousarray(w.transpose(2, 0, 1))          # [k, Co, Ci]
    dxp = np.zeros((B, L_in + 2 * pad + stride, Ci), F32)
    for tap in range(k):                                     # positions lo*stride+tap are distinct per tap
        dxp[:, tap:tap + (Lout - 1) * stride + 1:stride, :] += (dyl @ wt[tap]).reshape(B, Lout, Ci)
    return np.ascontiguousarray(dxp[:, pad:pad + L_in, :].transpose(0, 2, 1), dtype=F32)


def conv_transpose1d(x, w, b, stride=2, pad=1):
    """torch.nn.ConvTranspose1d(k,stride,pad).  x [B,Ci,L], w [Ci,Co,k] -> [B,Co,(L-1)*stride-2*pad+k]."""
    Ci, Co, k = w.shape
    B, _, L = x.shape
    Lfull = (L - 1) * stride + k
    xl = np.ascontiguousarray(x.transpose(0, 2, 1)).reshape(B * L, Ci)
    wt = np.ascontiguousarray(w.transpose(2, 0, 1))          # [k, Ci, Co]
    y = np.zeros((B, Lfull, Co), F32)
    for tap in range(k):
        y[:, tap:tap + (L - 1) * stride + 1:stride, :] += (xl @ wt[tap]).reshape(B, L, Co)
    y = y[:, pad:Lfull - pad, :] + b
    return np.ascontiguousarray(y.transpose(0, 2, 1), dtype=F32)


def group_norm(x, gamma, beta, groups=8, eps=1e-5):
    """torch.nn.GroupNorm on [B,C,L] (the reference inserts a dummy dim, helpers.py:56-58).
    Returns (out, xhat, rstd) -- the extra two are what the backward needs."""
    B, C, L = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(axis=2, keepdims=True, dtype=F32)
    var = ((xg - mean) ** 2).mean(axis=2, keepdims=True, dtype=F32)
    rstd = (F32(1) / np.sqrt(var + F32(eps))).astype(F32)
    xhat = ((xg - mean) * rstd).reshape(B, C, L).astype(F32)
    out = xhat * gamma[None, :, None] + beta[None, :, None]
    return out.astype(F32), xhat, rstd


def group_norm_input_grad(dout, xhat, rstd, gamma, groups=8):
    """Backward-to-input of GroupNorm: dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))."""
    B, C, L = dout.shape
    dxhat = (dout * gamma[None, :, None]).reshape(B, groups, -1)
    xh = xhat.reshape(B, groups, -1)
    m1 = dxhat.mean(axis=2, keepdims=True, dtype=F32)
    m2 = (dxhat * xh).mean(axis=2, keepdims=True, dtype=F32)
    dx = rstd * (dxhat - m1 - xh * m2)
    return dx.reshape(B, C, L).astype(F32)


def sinusoidal_pos_emb(t, dim):
    # helpers.py:21-28
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freqs = np.exp(np.arange(half, dtype=F32) * F32(-e)).astype(F32)
    ang = np.asarray(t, F32)[:, None] * freqs[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=-1).astype(F32)


# --------------------------------------------------------------------------- #
# building blocks (helpers.py:30-63, temporal.py:18-46)
# --------------------------------------------------------------------------- #

class _Net:
    def __init__(self, sd):
        self.sd = {k: np.asarray(v, F32) for k, v in sd.items()}

    def p(self, name):
        return self.sd[name]

    def time_embed(self, time, dim):
        # temporal.py:70-75 : SinusoidalPosEmb -> Linear(dim,4dim) -> Mish -> Linear(4dim,dim)
        e = sinusoidal_pos_emb(time, dim)
        e = mish(linear(e, self.p('time_mlp.1.weight'), self.p('time_mlp.1.bias')))
        return linear(e, self.p('time_mlp.3.weight'), self.p('time_mlp.3.bias'))

    # Conv1dBlock: Conv1d(k, pad=k//2) -> GroupNorm(8) -> Mish          (helpers.py:46-63)
    def conv_block(self, prefix, x, cache=None):
        w = self.p(f'{prefix}.block.0.weight')
        y = conv1d(x, w, self.p(f'{prefix}.block.0.bias'), 1, w.shape[2] // 2)
        z, xhat, rstd = group_norm(y, self.p(f'{prefix}.block.2.weight'), self.p(f'{prefix}.block.2.bias'))
        if cache is not None:
            cache[prefix] = (x.shape[2], xhat, rstd, z)
        return mish(z)

    def conv_block_input_grad(self, prefix, dout, cache):
        L_in, xhat, rstd, z = cache[prefix]
        w = self.p(f'{prefix}.block.0.weight')
        dz = dout * mish_grad(z)
        dy = group_norm_input_grad(dz, xhat, rstd, self.p(f'{prefix}.block.2.weight'))
        return conv1d_input_grad(dy, w, L_in, 1, w.shape[2] // 2)

    # ResidualTemporalBlock.forward                                       (temporal.py:37-46)
    def res_block(self, prefix, x, temb, cache=None):
        tb = linear(mish(temb), self.p(f'{prefix}.time_mlp.1.weight'), self.p(f'{prefix}.time_mlp.1.bias'))
        out = self.conv_block(f'{prefix}.blocks.0', x, cache) + tb[:, :, None]
        out = self.conv_block(f'{prefix}.blocks.1', out, cache)
        if f'{prefix}.residual_conv.weight' in self.sd:
            res = conv1d(x, self.p(f'{prefix}.residual_conv.weight'), self.p(f'{prefix}.residual_conv.bias'))
        else:
            res = x
        return (out + res).astype(F32)

    def res_block_input_grad(self, prefix, dout, cache):
        d = self.conv_block_input_grad(f'{prefix}.blocks.1', dout, cache)
        d = self.conv_block_input_grad(f'{prefix}.blocks.0', d, cache)
        if f'{prefix}.residual_conv.weight' in self.sd:
            d = d + conv1d_input_grad(dout, self.p(f'{prefix}.residual_conv.weight'), dout.shape[2])
        else:
            d = d + dout
        return d.astype(F32)

    def downsample(self, prefix, x):        # helpers.py:30-36  Conv1d(dim,dim,3,2,1)
        return conv1d(x, self.p(f'{prefix}.conv.weight'), self.p(f'{prefix}.conv.bias'), 2, 1)

    def downsample_input_grad(self, prefix, dout, L_in):
        return conv1d_input_grad(dout, self.p(f'{prefix}.conv.weight'), L_in, 2, 1)

    def upsample(self, prefix, x):          # helpers.py:38-44  ConvTranspose1d(dim,dim,4,2,1)
        return conv_transpose1d(x, self.p(f'{prefix}.conv.weight'), self.p(f'{prefix}.conv.bias'), 2, 1)


class TemporalUnet(_Net):
    """temporal.py:49-151.  ``attention`` is never enabled on this path."""

    def __init__(self, sd, horizon, transition_dim, dim=32, dim_mults=(1, 2, 4, 8)):
        super().__init__(sd)
        self.dim = dim
        self.n_levels = len(dim_mults)
        self.transition_dim = transition_dim

    def forward(self, x, cond, time, taps=None):
        x = np.ascontiguousarray(np.asarray(x, F32).transpose(0, 2, 1))     # b h t -> b t h  (:125)
        t = self.time_embed(time, self.dim)                                 # :127
        h = []
        for i in range(self.n_levels):                                      # :130-135
            x = self.res_block(f'downs.{i}.0', x, t)
            if taps is not None and i == 0:
                taps['down0_block0'] = x
            x = self.res_block(f'downs.{i}.1', x, t)
            h.append(x)
            if f'downs.{i}.3.conv.weight' in self.sd:
                x = self.downsample(f'downs.{i}.3', x)
        if taps is not None:
            taps['down_out'] = x
        x = self.res_block('mid_block1', x, t)                              # :137-139
        x = self.res_block('mid_block2', x, t)
        if taps is not None:
            taps['mid_out'] = x
        for i in range(self.n_levels - 1):                                  # :141-146
            x = np.concatenate([x, h.pop()], axis=1)
            x = self.res_block(f'ups.{i}.0', x, t)
            x = self.res_block(f'ups.{i}.1', x, t)
            if f'ups.{i}.3.conv.weight' in self.sd:
                x = self.upsample(f'ups.{i}.3', x)
        x = self.conv_block('final_conv.0', x)                              # :148 (final_conv[0])
        x = conv1d(x, self.p('final_conv.1.weight'), self.p('final_conv.1.bias'))
        return np.ascontiguousarray(x.transpose(0, 2, 1))                   # b t h -> b h t  (:150)

    __call__ = forward


class ValueFunction(_Net):
    """temporal.py:153-239."""

    def __init__(self, sd, horizon, transition_dim, dim=32, dim_mults=(1, 2, 4, 8)):
        super().__init__(sd)
        self.dim = dim
        self.n_levels = len(dim_mults)
        self.transition_dim = transition_dim

    def forward(self, x, cond, time, cache=None):
        x = np.ascontiguousarray(np.asarray(x, F32).transpose(0, 2, 1))     # :218
        t = self.time_embed(time, self.dim)                                 # :223
        lens = []
        for i in range(self.n_levels):                                      # :225-228
            x = self.res_block(f'blocks.{i}.0', x, t, cache)
            x = self.res_block(f'blocks.{i}.1', x, t, cache)
            lens.append(x.shape[2])
            x = self.downsample(f'blocks.{i}.2', x)
        x = self.res_block('mid_block1', x, t, cache)                       # :231-232
        lens.append(x.shape[2])
        x = self.downsample('mid_down1', x)
        x = self.res_block('mid_block2', x, t, cache)                       # :234-235
        lens.append(x.shape[2])
        x = self.downsample('mid_down2', x)
        shp = x.shape
        flat = x.reshape(len(x), -1)                                        # :237
        hin = np.concatenate([flat, t], axis=-1)                            # :238
        h1 = linear(hin, self.p('final_block.0.weight'), self.p('final_block.0.bias'))
        out = linear(mish(h1), self.p('final_block.2.weight'), self.p('final_block.2.bias'))
        if cache is not None:
            cache['__head__'] = (lens, shp, h1)
        return out                                                          # [B, out_dim]

    __call__ = forward

    def value_and_input_grad(self, x, cond, time):
        """What ValueGuide.gradients computes (guides.py:16-21): y = V(x).squeeze(-1),
        grad = d(sum y)/dx, by explicit backward-to-input (no weight gradients)."""
        cache = {}
        out = self.forward(x, cond, time, cache)
        lens, shp, h1 = cache['__head__']
        B = len(out)
        dout = np.ones((B, out.shape[1]), F32)                              # d(sum y)/dy
        dh = (dout @ self.p('final_block.2.weight')) * mish_grad(h1)
        dhin = dh @ self.p('final_block.0.weight')
        d = dhin[:, :shp[1] * shp[2]].reshape(shp).astype(F32)              # time-embedding part gets no grad to x
        d = self.downsample_input_grad('mid_down2', d, lens[-1])
        d = self.res_block_input_grad('mid_block2', d, cache)
        d = self.downsample_input_grad('mid_down1', d, lens[-2])
        d = self.res_block_input_grad('mid_block1', d, cache)
        for i in reversed(range(self.n_levels)):
            d = self.downsample_input_grad(f'blocks.{i}.2', d, lens[i])
            d = self.res_block_input_grad(f'blocks.{i}.1', d, cache)
            d = self.res_block_input_grad(f'blocks.{i}.0', d, cache)
        return out[:, 0].copy(), np.ascontiguousarray(d.transpose(0, 2, 1))


# --------------------------------------------------------------------------- #
# diffusion schedule (helpers.py:129-140, diffusion.py:85-116)
# --------------------------------------------------------------------------- #

def cosine_beta_schedule(timesteps, s=0.008):
    # helpers.py:129-140: float64 numpy, cast to float32 at the end
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return np.clip(betas, 0, 0.999).astype(F32)


def make_schedule(n_timesteps):
    """diffusion.py:85-116.  The reference mixes float32 torch tensors with np.sqrt
    on them (still float32); all arithmetic below is float32 in the same order."""
    betas = cosine_beta_schedule(n_timesteps)
    alphas = (F32(1) - betas).astype(F32)
    ac = np.cumprod(alphas, dtype=F32)                                      # :87
    ac_prev = np.concatenate([np.ones(1, F32), ac[:-1]])                    # :88
    post_var = (betas * (F32(1) - ac_prev) / (F32(1) - ac)).astype(F32)     # :106
    return dict(
        betas=betas, alphas_cumprod=ac, alphas_cumprod_prev=ac_prev,
        sqrt_alphas_cumprod=np.sqrt(ac), sqrt_one_minus_alphas_cumprod=np.sqrt(F32(1) - ac),
        posterior_variance=post_var,
        posterior_log_variance_clipped=np.log(np.maximum(post_var, F32(1e-20))).astype(F32),   # :111-112
        posterior_mean_coef1=(betas * np.sqrt(ac_prev) / (F32(1) - ac)).astype(F32),          # :113-114
        posterior_mean_coef2=((F32(1) - ac_prev) * np.sqrt(alphas) / (F32(1) - ac)).astype(F32),  # :115-116
    )


def apply_conditioning(x, cond, action_dim):
    # helpers.py:142-145 (in place)
    for t, val in cond.items():
        x[:, t, action_dim:action_dim + val.shape[-1]] = val
    return x


# --------------------------------------------------------------------------- #
# guide, sample functions, loop
# --------------------------------------------------------------------------- #

class ValueGuide:
    """guides.py:6-21 wrapping a ValueDiffusion whose forward is the bare net (diffusion.py:282-283)."""

    def __init__(self, value_net):
        self.net = value_net

    def gradients(self, x, cond, t):
        D = self.net.transition_dim
        y, g = self.net.value_and_input_grad(x[:, :, :D], cond, t)          # guides.py:13
        grad = np.zeros_like(x)
        grad[:, :, :D] = g
        return y, grad


class GaussianDiffusion:
    """Sampling half of diffusion.py:72-225 with predict_epsilon=False, clip_denoised=False
    (config/locomotion.py:30,39) and *injected* noise: ``noise[0]`` replaces torch.randn(shape)
    (:192) and ``noise[1 + k]`` the k-th randn_like call (functions.py:134), SURVEY.md 8a(iv)."""

    def __init__(self, model, horizon, observation_dim, action_dim, n_timesteps=20):
        self.model = model
        self.horizon = horizon
        self.observation_dim = observation_dim
        self.action_dim = action_dim
        self.transition_dim = observation_dim + action_dim
        self.n_timesteps = n_timesteps
        self.sched = make_schedule(n_timesteps)

    def p_mean_variance(self, x, cond, t):
        # diffusion.py:175-185 with predict_start_from_noise == identity (:163-164)
        x_recon = self.model(x, cond, t)
        s = self.sched
        c1 = s['posterior_mean_coef1'][t][:, None, None]
        c2 = s['posterior_mean_coef2'][t][:, None, None]
        mean = (c1 * x_recon + c2 * x).astype(F32)                          # :166-170
        return mean, s['posterior_variance'][t], s['posterior_log_variance_clipped'][t]

    def p_sample_loop(self, shape, cond, noise, sample_fn, return_chain=False,
                      sample_batch_size=None, **kw):
        # diffusion.py:187-213
        B = shape[0]
        x = np.array(noise[0], F32)                                         # :192
        x = apply_conditioning(x, cond, self.action_dim)                    # :193
        chain = [x.copy()] if return_chain else None
        draw = 1
        for i in reversed(range(self.n_timesteps)):                         # :198
            t = np.full((B,), i, np.int64)                                  # :199
            x, values, costs = sample_fn(self, x, cond, t, noise=noise[draw], **kw)   # :200
            draw += 1
            x = apply_conditioning(x, cond, self.action_dim)                # :201
            if return_chain:
                chain.append(x.copy())
        if sample_batch_size is None:
            x, values, costs = sort_by_values(x, values, costs)             # :208-209
        else:
            x, values, costs = max_by_values_in_batch(x, values, costs, sample_batch_size)
        if return_chain:
            chain = np.stack(chain, axis=1)                                 # :212 (unsorted)
        return x, values, costs, chain


def sort_by_values(x, values, costs):
    # diffusion.py:44-49.  torch.argsort is not stable; ties are broken by index here.
    inds = np.argsort(-values, kind='stable')
    return x[inds], values[inds], costs[inds]


def max_by_values_in_batch(x, values, costs, sample_batch_size):
    # diffusion.py:51-62
    B = x.shape[0]
    per = B // sample_batch_size
    xr = x.reshape(sample_batch_size, per, x.shape[-2], x.shape[-1])
    vr = values.reshape(sample_batch_size, per)
    cr = costs.reshape(sample_batch_size, per)
    inds = np.argmax(vr, axis=1)
    ar = np.arange(sample_batch_size)
    return xr[ar, inds], vr[ar, inds], cr[ar, inds]


def default_sample_fn(model, x, cond, t, noise, guide=None):
    # diffusion.py:19-29
    mean, _, logvar = model.p_mean_variance(x, cond, t)
    std = np.exp(F32(0.5) * logvar)[:, None, None]
    noise = np.array(noise, F32)
    noise[t == 0] = 0
    values = np.zeros(len(x), F32)
    return (mean + std * noise).astype(F32), values, values


def n_step_guided_p_sample(model, x, cond, t, guide, noise, scale=0.001, t_stopgrad=0,
                           n_guide_steps=1, scale_grad_by_std=True, state_grad_mask=False):
    # functions.py:9-37
    logvar = model.sched['posterior_log_variance_clipped'][t][:, None, None]
    std = np.exp(F32(0.5) * logvar).astype(F32)
    var = np.exp(logvar).astype(F32)
    for _ in range(n_guide_steps):
        y, grad = guide.gradients(x, cond, t)
        if scale_grad_by_std:
            grad = var * grad
        if state_grad_mask:
            grad[:, :, model.action_dim:] = 0
        grad[t < t_stopgrad] = 0
        x = (x + F32(scale) * grad).astype(F32)
        x = apply_conditioning(x, cond, model.action_dim)
    mean, _, _ = model.p_mean_variance(x, cond, t)
    noise = np.array(noise, F32)
    noise[t == 0] = 0
    return (mean + std * noise).astype(F32), y, np.zeros_like(y)


def feasibility_rule(r_y, cost_y, cost_threshold):
    # functions.py:126-129 -- global over the batch
    if np.sum(cost_y <= cost_threshold) == 0:
        return (-cost_y).astype(F32)
    return np.where(cost_y > cost_threshold, r_y - F32(1e8), r_y).astype(F32)


def n_step_guided_and_constrained_p_sample(model, x, cond, t, guide, noise, scale=0.001, t_stopgrad=0,
                                           n_guide_steps=1, scale_grad_by_std=True, cost_guide=None,
                                           cost_threshold=0, cost_grad_weight=1.0, state_grad_mask=False,
                                           guide_type=0):
    # functions.py:75-137
    assert cost_guide is not None
    logvar = model.sched['posterior_log_variance_clipped'][t][:, None, None]
    std = np.exp(F32(0.5) * logvar).astype(F32)
    var = np.exp(logvar).astype(F32)
    thr = np.asarray(cost_threshold, F32)
    for _ in range(n_guide_steps):
        cost_y, cost_grad = cost_guide.gradients(x, cond, t)                # :88
        r_y, r_grad = guide.gradients(x, cond, t)                           # :91
        over = (cost_y > thr)
        if guide_type == 0:                                                 # :95-99
            with np.errstate(all='ignore'):
                cw = (cost_y - thr) / thr
            cw = F32(cost_grad_weight) * np.exp(np.clip(cw, 0, 1.0)).astype(F32)
            grad = r_grad - cost_grad * cw[:, None, None] * over[:, None, None]
        else:                                                               # :100-102
            cw = F32(cost_grad_weight) * over.astype(F32)
            grad = r_grad - cost_grad * cw[:, None, None]
        grad = grad.astype(F32)
        if scale_grad_by_std:
            grad = var * grad
        if state_grad_mask:
            grad[:, :, model.action_dim:] = 0
        grad[t < t_stopgrad] = 0
        x = (x + F32(scale) * grad).astype(F32)
        x = apply_conditioning(x, cond, model.action_dim)
    r_y = feasibility_rule(r_y, cost_y, thr)                                # :126-129
    mean, _, _ = model.p_mean_variance(x, cond, t)                          # :131
    noise = np.array(noise, F32)
    noise[t == 0] = 0                                                       # :134-135
    return (mean + std * noise).astype(F32), r_y, cost_y
