"""GuidedPolicy: the caller of the planning hot path, with the reference's signature and semantics.

Reference: /root/reference/diffuser/sampling/policies.py:11-82.  One control step = normalise
the observation, repeat it `batch_size` times, build the per-candidate cost-threshold tensor, run the
reverse-diffusion loop, pick the best plan, un-normalise, return its first action plus the batch of plans.

On the device.  The reference moves the whole [batch, horizon, transition] batch to the host every control step to
read ONE action (policies.py:38-47; 24 MB at 8192 candidates).  Here one control step moves the raw observation and
the budget up (one pinned staging copy, < 100 bytes) and the selected action(s) down (action_dim floats per
observation): trebi_policy_prepare normalises / repeats / fills on the device straight into the captured loop's
buffers, the loop and the selection run there, trebi_policy_first_action un-normalises what the caller reads.  The
`Trajectories` the reference also returns are built LAZILY: nothing else crosses PCIe unless a field is read.

`normalizer` is duck-typed like the reference's DatasetNormalizer (`.normalize(x, key)` / `.unnormalize(x, key)` on
numpy arrays, diffuser/datasets/normalization.py:52-66).  Gaussian statistics (the configured normalizer,
config/locomotion.py:37) are taken to the device; any other normalizer keeps its own numpy arithmetic on the few
numbers involved (the observation, the selected action).  `preprocess_fns` is a list of callables (the reference
passes names that it `eval`s inside diffuser/datasets/preprocessing.py:26-28; its planning configs pass [])."""
import ctypes as C

import numpy as np
import torch

from . import _lib

FIELDS = ('actions', 'observations', 'rewards', 'terminals', 'values', 'costs')


class Trajectories:
    """The reference's namedtuple('Trajectories', 'actions observations rewards terminals values costs')
    (policies.py:11), with the numpy fields produced on first access from a device-side snapshot of the plans
    (sorted like the reference's `samples.trajectories`)."""
    _fields = FIELDS

    def __init__(self, plans, values, costs, normalizer, action_dim, observation_dim):
        self._plans, self.values, self.costs = plans, values, costs
        self._normalizer, self._act, self._obs = normalizer, action_dim, observation_dim
        self._cache = {}

    def _host(self):
        if 'host' not in self._cache:
            self._cache['host'] = self._plans.detach().cpu().numpy()        # the reference's utils.to_np(samples.trajectories)
        return self._cache['host']

    @property
    def actions(self):
        if 'actions' not in self._cache:
            self._cache['actions'] = self._normalizer.unnormalize(self._host()[:, :, :self._act], 'actions')
        return self._cache['actions']

    @property
    def observations(self):
        if 'observations' not in self._cache:
            t = self._host()
            self._cache['observations'] = self._normalizer.unnormalize(t[:, :, self._act:self._act + self._obs],
                                                                       'observations')
        return self._cache['observations']

    @property
    def rewards(self):
        t, d = self._host(), self._act + self._obs
        return t[:, :, d:d + 1] if t.shape[-1] > d else np.zeros(t.shape[:-1] + (1,))

    @property
    def terminals(self):
        t, d = self._host(), self._act + self._obs
        if t.shape[-1] > d:
            return self._normalizer.unnormalize(t[:, :, d + 1:d + 2], 'terminals')
        return np.zeros(t.shape[:-1] + (1,))

    def __iter__(self):
        return iter(getattr(self, f) for f in FIELDS)

    def __getitem__(self, i):
        return getattr(self, FIELDS[i])

    def __len__(self):
        return len(FIELDS)


class GaussianNormalizer:
    """x -> (x - mean) / std per key (normalization.py:131-149), from given statistics."""

    def __init__(self, stats):
        self.stats = {k: (np.asarray(m, np.float32), np.asarray(s, np.float32)) for k, (m, s) in stats.items()}

    def normalize(self, x, key):
        m, s = self.stats[key]
        return (x - m) / s

    def unnormalize(self, x, key):
        m, s = self.stats[key]
        return x * s + m


def gaussian_stats(normalizer, key):
    """(mean, std) if `normalizer` is Gaussian for `key` (this package's class, or the reference's DatasetNormalizer
    holding GaussianNormalizers: normalization.py:127-149), else None."""
    if isinstance(normalizer, GaussianNormalizer):
        return normalizer.stats.get(key)
    sub = getattr(normalizer, 'normalizers', {}).get(key) if hasattr(normalizer, 'normalizers') else None
    if sub is not None and type(sub).__name__ == 'GaussianNormalizer' and hasattr(sub, 'means') and hasattr(sub, 'stds'):
        return np.asarray(sub.means, np.float32), np.asarray(sub.stds, np.float32)
    return None


def _compose(fns):
    for f in fns:
        if not callable(f):
            raise RuntimeError(f'preprocess_fns must be callables here (got {f!r}); the reference resolves names '
                               'inside diffuser.datasets.preprocessing, which is outside this package')

    def _fn(x):
        for f in fns:
            x = f(x)
        return x
    return _fn


class GuidedPolicy:

    def __init__(self, guide, diffusion_model, normalizer, preprocess_fns, **sample_kwargs):    # policies.py:16-23
        self.guide = guide
        self.diffusion_model = diffusion_model
        self.normalizer = normalizer
        self.action_dim = diffusion_model.action_dim
        self.observation_dim = diffusion_model.observation_dim
        self.preprocess_fn = _compose(list(preprocess_fns or []))
        self.sample_kwargs = sample_kwargs
        self._dev = {}

    @property
    def device(self):                                                                           # policies.py:65-68
        return next(self.diffusion_model.parameters()).device

    # ------------------------------------------------------------------ device-side glue
    def _stats(self, key, n):
        """Gaussian statistics of `key` on the device (None: that normalizer stays on the host)."""
        if key not in self._dev:
            st = gaussian_stats(self.normalizer, key)
            if st is None or len(st[0]) != n:
                self._dev[key] = None
            else:
                self._dev[key] = tuple(torch.as_tensor(np.ascontiguousarray(a, np.float32), device=self.device) for a in st)
        return self._dev[key]

    def _staging(self, n):
        key = ('stage', n)
        if key not in self._dev:
            self._dev[key] = (torch.empty(n, dtype=torch.float32).pin_memory(), torch.empty(n, device=self.device))
        return self._dev[key]

    def __call__(self, conditions, batch_size=1, verbose=True, cost_threshold=None, plan_horizon=None):
        # policies.py:25-63
        from . import diffusion as Dm
        sample_batch_size = None if len(conditions[0].shape) == 1 else conditions[0].shape[0]
        conditions = {k: np.asarray(self.preprocess_fn(v), np.float32) for k, v in conditions.items()}
        if cost_threshold is not None:
            self.sample_kwargs['cost_threshold'] = cost_threshold
        dm, dev = self.diffusion_model, self.device
        kw = dict(self.sample_kwargs)
        sample_fn = kw.pop('sample_fn', Dm.default_sample_fn)
        noise = kw.pop('noise', None)
        kw['guide'] = self.guide
        H = plan_horizon or dm.horizon
        n_obs = sample_batch_size or 1
        B = batch_size * n_obs
        if not dm._fusable(sample_fn, kw) or B == 0:
            return self._call_stepwise(conditions, batch_size, verbose, plan_horizon, sample_batch_size)
        kind = Dm._FUSED_KINDS[sample_fn]
        keys = sorted(conditions.keys())
        lib, ctx = _lib.lib(), _lib.ctx(dev.index or 0)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        s = dm._sampler(kind, B, H, keys, kw, dev, True, False)

        # ---- up: raw observation(s) + budget, one pinned staging copy ----
        if kind == 2 and 'cost_threshold' not in kw:
            raise KeyError('cost_threshold')                                         # policies.py:33 indexes it
        thr = kw.get('cost_threshold') if kind == 2 else None
        if torch.is_tensor(thr):
            thr = thr.detach().cpu().numpy()
        thr = None if thr is None else np.atleast_1d(np.asarray(thr, np.float32)).reshape(-1)
        st_obs = self._stats('observations', self.observation_dim)
        vals = []
        for k in keys:
            v = conditions[k].reshape(n_obs, self.observation_dim)
            if st_obs is None:                      # a normalizer without device statistics: its own arithmetic, 17 numbers
                v = np.asarray(self.normalizer.normalize(v, 'observations'), np.float32)
            vals.append(v.reshape(-1))
        if thr is not None:
            vals.append(thr)
        flat = np.concatenate(vals)
        pin, stage = self._staging(flat.size)
        pin.numpy()[:] = flat
        stage.copy_(pin, non_blocking=True)
        off = 0
        for i, k in enumerate(keys):
            n = n_obs * self.observation_dim
            last = i == len(keys) - 1
            _lib.check(lib.trebi_policy_prepare(
                ctx, _lib.ptr(stage[off:off + n]), n_obs, self.observation_dim,
                _lib.ptr(st_obs[0]) if st_obs else None, _lib.ptr(st_obs[1]) if st_obs else None, B, _lib.ptr(s['cond'][i]),
                _lib.ptr(stage[len(keys) * n:]) if (thr is not None and last) else None, 0 if thr is None else thr.size,
                _lib.ptr(s['thr']) if (thr is not None and last) else None, st))
            off += n

        # ---- the captured loop + selection, all on the device ----
        x, values, costs, _ = dm.sample_raw((B, H, dm.transition_dim), {k: None for k in keys}, sample_fn=sample_fn,
                                            noise=noise, prepared=True, **kw)
        if kind == 2:
            values = Dm.feasibility_rule(values, costs, s['thr'])
        if sample_batch_size is None:
            order = Dm.argsort_descending(values)                                    # diffusion.py:44-49
            index = order[:1]
        else:
            order = torch.empty(sample_batch_size, device=dev, dtype=torch.int64)    # diffusion.py:51-62
            _lib.check(lib.trebi_group_argmax(ctx, _lib.ptr(values.contiguous()), sample_batch_size,
                                              B // sample_batch_size, _lib.ptr(order), st))
            index = order
        st_act = self._stats('actions', self.action_dim)
        act_dev = torch.empty(index.numel(), self.action_dim, device=dev)
        _lib.check(lib.trebi_policy_first_action(
            ctx, _lib.ptr(x), B, H, dm.transition_dim, self.action_dim, _lib.ptr(index), index.numel(),
            _lib.ptr(st_act[0]) if st_act else None, _lib.ptr(st_act[1]) if st_act else None, _lib.ptr(act_dev), st))
        # ---- down: the selected action(s), action_dim floats per observation ----
        action = act_dev.cpu().numpy()
        if st_act is None:
            action = np.asarray(self.normalizer.unnormalize(action, 'actions'))
        action = action[0] if sample_batch_size is None else action
        # the plans stay on the device (a snapshot in the reference's order: the loop's buffers are reused next call)
        trajectories = Trajectories(x[order], values[order], costs[order], self.normalizer, self.action_dim,
                                    self.observation_dim)
        return action, trajectories

    def _call_stepwise(self, conditions, batch_size, verbose, plan_horizon, sample_batch_size):
        """Anything the fused loop does not take (a custom sample_fn, an epsilon-parameterised model): the reference's
        own flow, policies.py:25-63, through diffusion_model(...)."""
        conditions = self._format_conditions(conditions, batch_size)
        if 'cost_threshold' in self.sample_kwargs:
            thr = self.sample_kwargs['cost_threshold']
            if torch.is_tensor(thr):
                thr = thr.detach().cpu().numpy()
            if hasattr(thr, '__len__'):
                thr = torch.tensor(np.asarray(thr, np.float32)).repeat(batch_size)
            else:
                thr = torch.tensor([float(thr)]).repeat(batch_size)
            self.sample_kwargs['cost_threshold'] = thr.to(self.device)
        samples = self.diffusion_model(conditions, guide=self.guide, verbose=verbose, horizon=plan_horizon,
                                       sample_batch_size=sample_batch_size, **self.sample_kwargs)
        traj = Trajectories(samples.trajectories, samples.values, samples.costs, self.normalizer, self.action_dim,
                            self.observation_dim)
        action = traj.actions[0, 0] if sample_batch_size is None else traj.actions[:, 0]
        return action, traj

    def _format_conditions(self, conditions, batch_size):                                       # policies.py:70-82
        out = {}
        for k, v in conditions.items():
            v = self.normalizer.normalize(np.asarray(v), 'observations')
            v = torch.as_tensor(v, dtype=torch.float32, device=self.device)
            # einops 'd -> repeat d' / 'd w -> (repeat d) w': the repeat index is the OUTER one
            out[k] = v[None].repeat(batch_size, 1) if v.dim() == 1 else v.repeat(batch_size, 1)
        return out
