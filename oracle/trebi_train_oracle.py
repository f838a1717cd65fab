"""ORACLE -- test infrastructure, not product code.

numpy (float32) restatement of TREBI's diffusion TRAINING step (BASELINE config 5):
q_sample + conditioning, TemporalUnet forward, weighted-L2 loss, the full backward (input, weight, bias,
GroupNorm and time-MLP gradients), gradient accumulation, Adam and the EMA copy -- what
`Trainer.train`'s inner loop does for one optimiser step.  Same rules as trebi_oracle.py: only tests/ (and a
future bench cpu_baseline leg) may import it.

Parity status: PINNED.  tests/test_train_oracle_golden.py checks loss, every parameter gradient, the Adam update
and the EMA copy against vectors produced by the reference modules themselves under torch.autograd /
torch.optim.Adam on CPU (tests/golden/make_train_golden.py -> tests/golden/train_*.npz).

Reference lines followed (relative to /root/reference):
  diffuser/models/diffusion.py:122-149 (loss weights), :229-261 (q_sample, p_losses, loss)
  diffuser/models/helpers.py:152-167,198-201 (WeightedLoss / WeightedL2), :142-145 (apply_conditioning)
  diffuser/models/temporal.py:18-46,49-151 (ResidualTemporalBlock, TemporalUnet)
  diffuser/utils/training.py:19-35 (EMA), :94-98 (step_ema), :104-122 (the optimiser step)
The backward functions restate what torch.autograd computes for Conv1d / ConvTranspose1d / GroupNorm / Mish /
Linear; Adam restates torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay, bias-corrected).
"""
import numpy as np

import trebi_oracle as O

F32 = np.float32


# --------------------------------------------------------------------------- #
# parameter gradients of the primitives
# --------------------------------------------------------------------------- #

def conv1d_param_grad(x, dy, k, stride=1, pad=0):
    """dW[co,ci,tap] = sum_{b,lo} dy[b,co,lo] * xpad[b,ci,lo*stride+tap];  db[co] = sum dy."""
    B, Ci, L = x.shape
    _, Co, Lout = dy.shape
    xp = np.zeros((B, L + 2 * pad, Ci), F32)
    xp[:, pad:pad + L, :] = x.transpose(0, 2, 1)
    dyl = np.ascontiguousarray(dy.transpose(0, 2, 1)).reshape(B * Lout, Co)
    dw = np.zeros((Co, Ci, k), F32)
    for tap in range(k):
        rows = np.ascontiguousarray(xp[:, tap:tap + (Lout - 1) * stride + 1:stride, :]).reshape(B * Lout, Ci)
        dw[:, :, tap] = dyl.T @ rows
    return dw, dy.sum(axis=(0, 2), dtype=F32)


def conv_transpose1d_grads(x, w, dy, stride=2, pad=1):
    """Backward of torch.nn.ConvTranspose1d (x [B,Ci,L], w [Ci,Co,k], dy [B,Co,Lout]) -> (dx, dw, db)."""
    Ci, Co, k = w.shape
    B, _, L = x.shape
    Lfull = (L - 1) * stride + k
    dyf = np.zeros((B, Lfull, Co), F32)
    dyf[:, pad:Lfull - pad, :] = dy.transpose(0, 2, 1)
    xl = np.ascontiguousarray(x.transpose(0, 2, 1)).reshape(B * L, Ci)
    dx = np.zeros((B * L, Ci), F32)
    dw = np.zeros((Ci, Co, k), F32)
    for tap in range(k):
        d = np.ascontiguousarray(dyf[:, tap:tap + (L - 1) * stride + 1:stride, :]).reshape(B * L, Co)
        dx += d @ w[:, :, tap].T
        dw[:, :, tap] = xl.T @ d
    dx = np.ascontiguousarray(dx.reshape(B, L, Ci).transpose(0, 2, 1), dtype=F32)
    return dx, dw, dy.sum(axis=(0, 2), dtype=F32)


def linear_grads(x, w, dy):
    """y = x @ w.T + b  ->  (dx, dw, db)."""
    return (dy @ w).astype(F32), (dy.T @ x).astype(F32), dy.sum(axis=0, dtype=F32)


# --------------------------------------------------------------------------- #
# TemporalUnet forward with a cache, and the full backward
# --------------------------------------------------------------------------- #

class TrainableUnet(O.TemporalUnet):
    """temporal.py:49-151 with everything the backward needs kept in `cache`."""

    # ---- Conv1dBlock (helpers.py:46-63) ----
    def _cb_fwd(self, prefix, x, cache):
        w = self.p(f'{prefix}.block.0.weight')
        y = O.conv1d(x, w, self.p(f'{prefix}.block.0.bias'), 1, w.shape[2] // 2)
        z, xhat, rstd = O.group_norm(y, self.p(f'{prefix}.block.2.weight'), self.p(f'{prefix}.block.2.bias'))
        cache[prefix] = (x, xhat, rstd, z)
        return O.mish(z)

    def _cb_bwd(self, prefix, dout, cache, g):
        x, xhat, rstd, z = cache[prefix]
        w = self.p(f'{prefix}.block.0.weight')
        k = w.shape[2]
        dz = (dout * O.mish_grad(z)).astype(F32)
        g[f'{prefix}.block.2.weight'] = (dz * xhat).sum(axis=(0, 2), dtype=F32)
        g[f'{prefix}.block.2.bias'] = dz.sum(axis=(0, 2), dtype=F32)
        dy = O.group_norm_input_grad(dz, xhat, rstd, self.p(f'{prefix}.block.2.weight'))
        g[f'{prefix}.block.0.weight'], g[f'{prefix}.block.0.bias'] = conv1d_param_grad(x, dy, k, 1, k // 2)
        return O.conv1d_input_grad(dy, w, x.shape[2], 1, k // 2)

    # ---- ResidualTemporalBlock (temporal.py:18-46) ----
    def _rb_fwd(self, prefix, x, temb, cache):
        mt = O.mish(temb)
        tb = O.linear(mt, self.p(f'{prefix}.time_mlp.1.weight'), self.p(f'{prefix}.time_mlp.1.bias'))
        h0 = self._cb_fwd(f'{prefix}.blocks.0', x, cache) + tb[:, :, None]
        h1 = self._cb_fwd(f'{prefix}.blocks.1', h0.astype(F32), cache)
        if f'{prefix}.residual_conv.weight' in self.sd:
            res = O.conv1d(x, self.p(f'{prefix}.residual_conv.weight'), self.p(f'{prefix}.residual_conv.bias'))
        else:
            res = x
        cache[prefix] = (x, mt)
        return (h1 + res).astype(F32)

    def _rb_bwd(self, prefix, dout, cache, g, dtemb, temb):
        x, mt = cache[prefix]
        dh0 = self._cb_bwd(f'{prefix}.blocks.1', dout, cache, g)
        dtb = dh0.sum(axis=2, dtype=F32)                                   # the bias broadcast over positions
        wt = self.p(f'{prefix}.time_mlp.1.weight')
        dmt, g[f'{prefix}.time_mlp.1.weight'], g[f'{prefix}.time_mlp.1.bias'] = linear_grads(mt, wt, dtb)
        dtemb += dmt * O.mish_grad(temb)
        dx = self._cb_bwd(f'{prefix}.blocks.0', dh0, cache, g)
        if f'{prefix}.residual_conv.weight' in self.sd:
            wr = self.p(f'{prefix}.residual_conv.weight')
            g[f'{prefix}.residual_conv.weight'], g[f'{prefix}.residual_conv.bias'] = conv1d_param_grad(x, dout, 1)
            dx = dx + O.conv1d_input_grad(dout, wr, x.shape[2])
        else:
            dx = dx + dout
        return dx.astype(F32)

    # ---- whole network ----
    def forward_train(self, x, time):
        cache = {}
        x = np.ascontiguousarray(np.asarray(x, F32).transpose(0, 2, 1))
        e = O.sinusoidal_pos_emb(time, self.dim)                            # temporal.py:70-75
        a1 = O.linear(e, self.p('time_mlp.1.weight'), self.p('time_mlp.1.bias'))
        m = O.mish(a1)
        temb = O.linear(m, self.p('time_mlp.3.weight'), self.p('time_mlp.3.bias'))
        cache['time'] = (e, a1, m, temb)
        h = []
        for i in range(self.n_levels):
            x = self._rb_fwd(f'downs.{i}.0', x, temb, cache)
            x = self._rb_fwd(f'downs.{i}.1', x, temb, cache)
            h.append(x)
            if f'downs.{i}.3.conv.weight' in self.sd:
                cache[f'downs.{i}.3'] = x
                x = self.downsample(f'downs.{i}.3', x)
        x = self._rb_fwd('mid_block1', x, temb, cache)
        x = self._rb_fwd('mid_block2', x, temb, cache)
        for i in range(self.n_levels - 1):
            skip = h.pop()
            cache[f'ups.{i}.cat'] = x.shape[1]
            x = np.concatenate([x, skip], axis=1)
            x = self._rb_fwd(f'ups.{i}.0', x, temb, cache)
            x = self._rb_fwd(f'ups.{i}.1', x, temb, cache)
            if f'ups.{i}.3.conv.weight' in self.sd:
                cache[f'ups.{i}.3'] = x
                x = self.upsample(f'ups.{i}.3', x)
        x = self._cb_fwd('final_conv.0', x, cache)
        cache['final_conv.1'] = x
        x = O.conv1d(x, self.p('final_conv.1.weight'), self.p('final_conv.1.bias'))
        return np.ascontiguousarray(x.transpose(0, 2, 1)), cache

    def backward(self, dout, cache):
        """dout: dL/d(output) [B, H, D].  Returns {state_dict name: gradient}."""
        g = {}
        e, a1, m, temb = cache['time']
        dtemb = np.zeros_like(temb)
        d = np.ascontiguousarray(np.asarray(dout, F32).transpose(0, 2, 1))
        xin = cache['final_conv.1']
        g['final_conv.1.weight'], g['final_conv.1.bias'] = conv1d_param_grad(xin, d, 1)
        d = O.conv1d_input_grad(d, self.p('final_conv.1.weight'), xin.shape[2])
        d = self._cb_bwd('final_conv.0', d, cache, g)
        dskip = []                                                          # gradients of h[...] in pop order reversed
        for i in reversed(range(self.n_levels - 1)):
            if f'ups.{i}.3.conv.weight' in self.sd:
                xu = cache[f'ups.{i}.3']
                d, g[f'ups.{i}.3.conv.weight'], g[f'ups.{i}.3.conv.bias'] = conv_transpose1d_grads(
                    xu, self.p(f'ups.{i}.3.conv.weight'), d)
            d = self._rb_bwd(f'ups.{i}.1', d, cache, g, dtemb, temb)
            d = self._rb_bwd(f'ups.{i}.0', d, cache, g, dtemb, temb)
            c = cache[f'ups.{i}.cat']
            dskip.append(d[:, c:, :])                                       # skip popped for level i
            d = np.ascontiguousarray(d[:, :c, :])
        d = self._rb_bwd('mid_block2', d, cache, g, dtemb, temb)
        d = self._rb_bwd('mid_block1', d, cache, g, dtemb, temb)
        # ups.i consumed h[n_levels-1-i]; dskip was appended for i = n-2 .. 0, i.e. h[1], h[2], ..., h[n-1]
        hgrad = {self.n_levels - 1 - i: dskip[j] for j, i in enumerate(reversed(range(self.n_levels - 1)))}
        for i in reversed(range(self.n_levels)):
            if f'downs.{i}.3.conv.weight' in self.sd:
                xd = cache[f'downs.{i}.3']
                w = self.p(f'downs.{i}.3.conv.weight')
                g[f'downs.{i}.3.conv.weight'], g[f'downs.{i}.3.conv.bias'] = conv1d_param_grad(xd, d, 3, 2, 1)
                d = O.conv1d_input_grad(d, w, xd.shape[2], 2, 1)
            if i in hgrad:
                d = (d + hgrad[i]).astype(F32)
            d = self._rb_bwd(f'downs.{i}.1', d, cache, g, dtemb, temb)
            d = self._rb_bwd(f'downs.{i}.0', d, cache, g, dtemb, temb)
        dm, g['time_mlp.3.weight'], g['time_mlp.3.bias'] = linear_grads(m, self.p('time_mlp.3.weight'), dtemb)
        da1 = (dm * O.mish_grad(a1)).astype(F32)
        _, g['time_mlp.1.weight'], g['time_mlp.1.bias'] = linear_grads(e, self.p('time_mlp.1.weight'), da1)
        return {k: np.asarray(v, F32) for k, v in g.items()}


# --------------------------------------------------------------------------- #
# loss (diffusion.py:122-149, 229-261; helpers.py:152-167, 198-201)
# --------------------------------------------------------------------------- #

def loss_weights(horizon, transition_dim, action_dim, action_weight=10.0, discount=1.0, weights_dict=None):
    dim_w = np.ones(transition_dim, F32)
    for ind, w in (weights_dict or {}).items():
        dim_w[action_dim + ind] *= w
    disc = (F32(discount) ** np.arange(horizon, dtype=F32)).astype(F32)
    disc = disc / disc.mean(dtype=F32)
    lw = np.einsum('h,t->ht', disc, dim_w).astype(F32)
    lw[0, :action_dim] = action_weight
    return lw


def p_losses(unet, x_start, cond, t, noise, lw, action_dim, sched):
    """diffusion.py:240-256 with predict_epsilon=False, loss_type 'l2'.
    Returns (loss, a0_loss, grads) with grads = d loss / d parameter."""
    x_start = np.asarray(x_start, F32)
    sa = sched['sqrt_alphas_cumprod'][t][:, None, None]
    sb = sched['sqrt_one_minus_alphas_cumprod'][t][:, None, None]
    x_noisy = (sa * x_start + sb * noise).astype(F32)                       # q_sample :229-238
    x_noisy = O.apply_conditioning(x_noisy, cond, action_dim)
    x_recon, cache = unet.forward_train(x_noisy, t)
    x_recon = O.apply_conditioning(x_recon.copy(), cond, action_dim)
    diff = (x_recon - x_start).astype(F32)
    l = diff * diff                                                         # F.mse_loss(reduction='none')
    loss = (l * lw).mean(dtype=F32)
    a0 = (l[:, 0, :action_dim] / lw[0, :action_dim]).mean(dtype=F32)
    dout = (F32(2) * diff * lw / F32(l.size)).astype(F32)
    for k, val in cond.items():                                             # overwritten outputs carry no gradient
        dout[:, k, action_dim:action_dim + val.shape[-1]] = 0
    return float(loss), float(a0), unet.backward(dout, cache)


# --------------------------------------------------------------------------- #
# optimiser step (training.py:104-122), torch.optim.Adam, EMA (training.py:19-35, 94-98)
# --------------------------------------------------------------------------- #

class Adam:
    def __init__(self, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.m, self.v, self.t = {}, {}, 0

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for k, g in grads.items():
            m = self.m.get(k, np.zeros_like(g))
            v = self.v.get(k, np.zeros_like(g))
            m = (F32(self.b1) * m + F32(1 - self.b1) * g).astype(F32)
            v = (F32(self.b2) * v + F32(1 - self.b2) * g * g).astype(F32)
            self.m[k], self.v[k] = m, v
            denom = (np.sqrt(v) / F32(np.sqrt(bc2)) + F32(self.eps)).astype(F32)
            params[k] = (params[k] - F32(self.lr / bc1) * (m / denom)).astype(F32)


def train_step(sd, ema_sd, opt, batches, horizon, obs_dim, action_dim, n_timesteps, step,
               action_weight=10.0, ema_decay=0.995, step_start_ema=2000, update_ema_every=10, dim=32,
               dim_mults=(1, 2, 4, 8)):
    """One pass of the loop body training.py:108-122.  `batches`: one (x_start, cond, t, noise) per accumulation
    micro-batch (the reference draws t and noise with torch RNG; the golden run injects them).
    Updates sd / ema_sd in place and returns the per-micro-batch losses."""
    D = obs_dim + action_dim
    sched = O.make_schedule(n_timesteps)
    lw = loss_weights(horizon, D, action_dim, action_weight)
    unet = TrainableUnet(sd, horizon, D, dim, dim_mults)
    acc, losses = None, []
    for x0, cond, t, noise in batches:
        loss, a0, g = p_losses(unet, x0, cond, t, noise, lw, action_dim, sched)
        losses.append((loss, a0))
        g = {k: (v / F32(len(batches))).astype(F32) for k, v in g.items()}  # loss / gradient_accumulate_every (:115)
        acc = g if acc is None else {k: (acc[k] + g[k]).astype(F32) for k in g}
    opt.step(sd, acc)                                                       # :118
    if step % update_ema_every == 0:                                        # :121-122, :94-98
        for k in acc:
            if step < step_start_ema:
                ema_sd[k] = sd[k].copy()
            else:
                ema_sd[k] = (ema_sd[k] * F32(ema_decay) + F32(1 - ema_decay) * sd[k]).astype(F32)
    return losses, acc
