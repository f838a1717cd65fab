"""Shared helpers for the parity tests: fixture loading and weight regeneration."""
import os

import numpy as np

from trebi_amd import spec, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CONFIG_NAMES = ('pendulum', 'halfcheetah', 'hopper', 'walker148', 'halfcheetah_h4', 'pendulum_h128')


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, f'{name}.npz')))


def fixture_weights(fx):
    """Rebuild the three synthetic state_dicts a fixture was generated with."""
    H, D, dim = int(fx['H']), int(fx['obs']) + int(fx['act']), int(fx['dim'])
    um = tuple(int(m) for m in fx['unet_mults'])
    vm = tuple(int(m) for m in fx['value_mults'])
    s_u, s_r, s_c = (int(s) for s in fx['seeds'][:3])
    sd_u = synth.synth_state_dict(spec.unet_param_shapes(H, D, dim, um), s_u)
    sd_r = synth.synth_state_dict(spec.value_param_shapes(H, D, dim, vm), s_r)
    sd_c = synth.synth_state_dict(spec.value_param_shapes(H, D, dim, vm), s_c)
    return sd_u, sd_r, sd_c, um, vm


def fixture_kwargs(fx):
    kw = {}
    for k in fx:
        if k.startswith('kw_'):
            v = fx[k]
            kw[k[3:]] = (bool(v) if v.dtype == bool else int(v) if np.issubdtype(v.dtype, np.integer) else float(v))
    return kw


def rel_err(a, b):
    """max |a-b| / max |b| -- the relative error the parity bars are stated in."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def sample_elems(v, n):
    """The evenly strided elements make_train_golden.py's pin fixture stores (float32)."""
    v = np.asarray(v, np.float32).reshape(-1)
    return v[np.linspace(0, v.size - 1, min(v.size, n)).astype(np.int64)]


def check_optimizer_pin(fx, params_by_step, emas_by_step):
    """tests/golden/train_ema_pin.npz (4 optimiser steps, lr 1e-3, ema_decay 0.9, step_start_ema 1) against the
    parameters / EMA copies a training path produced after each step (dicts name -> array).  Stated bars:
      * Adam (training.py:79,118; torch.optim.Adam): on the sampled elements whose gradient is above rounding noise
        (|g| > 1e-3 x the tensor's largest sampled |g| at every step so far), the parameter after EVERY step within
        1e-3 x lr of the reference -- steps 1..3 depend on the moments, the bias corrections and where eps sits;
      * EMA (training.py:19-35,94-98): step 0 copies the parameters bit for bit; at steps 1..3, on the elements where the
        parameter and the previous average differ by more than 0.2 x lr,
            (ema_k - ema_{k-1}) / (p_k - ema_{k-1}) == 1 - decay within 2 %
        -- a rule that did nothing (ratio 0), copied (ratio 1) or used another decay fails.
    Returns the worst figures for the test to print."""
    lr, decay, ns = float(fx['lr']), float(fx['ema_decay']), int(fx['nsample'])
    names = [str(n) for n in fx['names']]
    worst_adam = worst_ratio = 0.0
    n_adam = n_ema = 0
    for k in names:
        live = np.ones(min(int(np.prod(np.shape(params_by_step[0][k]))), ns), bool)
        for step in range(int(fx['steps'])):
            g = np.abs(fx[f'g{step}.{k}'].astype(np.float64))
            live &= g > 1e-3 * max(g.max(), 1e-30)
            got_p = sample_elems(params_by_step[step][k], ns).astype(np.float64)
            want_p = fx[f'p{step}.{k}'].astype(np.float64)
            if live.any():
                d = np.abs(got_p - want_p)[live].max()
                worst_adam = max(worst_adam, d / lr)
                n_adam += int(live.sum())
                assert d <= 1e-3 * lr, ('adam', k, step, d / lr)
            got_e = sample_elems(emas_by_step[step][k], ns).astype(np.float64)
            if step == 0:
                assert np.array_equal(sample_elems(emas_by_step[0][k], ns), sample_elems(params_by_step[0][k], ns)), ('ema copy', k)
                continue
            prev = sample_elems(emas_by_step[step - 1][k], ns).astype(np.float64)
            gap = got_p - prev
            sel = np.abs(gap) > 0.2 * lr
            if sel.any():
                ratio = (got_e - prev)[sel] / gap[sel]
                dev = np.abs(ratio - (1.0 - decay)).max() / (1.0 - decay)
                worst_ratio = max(worst_ratio, dev)
                n_ema += int(sel.sum())
                assert dev <= 0.02, ('ema', k, step, float(ratio.min()), float(ratio.max()))
            # and against the reference's own average, where the parameters themselves are pinned
            if live.any():
                assert np.abs(got_e - fx[f'ema{step}.{k}'].astype(np.float64))[live].max() <= 2e-3 * lr * (step + 1), ('ema value', k, step)
    assert n_adam > 1000 and n_ema > 1000, (n_adam, n_ema)          # the pins did look at something
    return worst_adam, worst_ratio, n_adam, n_ema
