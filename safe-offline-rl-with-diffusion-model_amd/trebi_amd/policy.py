"""GuidedPolicy: the caller of the planning hot path, with the reference's signature and semantics.

Reference: /root/reference/diffuser/sampling/policies.py:11-82.  One control step =
normalise the observation, repeat it `batch_size` times, build the per-candidate cost-threshold tensor, run the
reverse-diffusion loop (here: the captured HIP loop, diffusion.GaussianDiffusion), un-normalise, return the first
action of the best candidate plus the whole batch of plans.

`normalizer` is duck-typed like the reference's DatasetNormalizer: `.normalize(x, key)` / `.unnormalize(x, key)`
on numpy arrays (diffuser/datasets/normalization.py:52-66).  `GaussianNormalizer` below is a dataset-free stand-in
with the arithmetic of normalization.py:145-149 for callers that only have the statistics.
`preprocess_fns` is a list of callables (the reference passes names that it `eval`s inside
diffuser/datasets/preprocessing.py:26-28; its planning configs pass an empty list)."""
from collections import namedtuple

import numpy as np
import torch

Trajectories = namedtuple('Trajectories', 'actions observations rewards terminals values costs')   # policies.py:11


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

    @property
    def device(self):                                                                           # policies.py:65-68
        return next(self.diffusion_model.parameters()).device

    def __call__(self, conditions, batch_size=1, verbose=True, cost_threshold=None, plan_horizon=None):
        # policies.py:25-63, statement for statement
        sample_batch_size = None if len(conditions[0].shape) == 1 else conditions[0].shape[0]
        conditions = {k: self.preprocess_fn(v) for k, v in conditions.items()}
        conditions = self._format_conditions(conditions, batch_size)

        if cost_threshold is not None:
            self.sample_kwargs['cost_threshold'] = cost_threshold
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
        trajectories = samples.trajectories.detach().cpu().numpy()

        actions = trajectories[:, :, :self.action_dim]
        actions = self.normalizer.unnormalize(actions, 'actions')
        action = actions[0, 0] if sample_batch_size is None else actions[:, 0]

        tran_dim = self.action_dim + self.observation_dim
        normed_observations = trajectories[:, :, self.action_dim:tran_dim]
        observations = self.normalizer.unnormalize(normed_observations, 'observations')
        if trajectories.shape[-1] > tran_dim:
            rewards = trajectories[:, :, tran_dim:tran_dim + 1]
            terminals = self.normalizer.unnormalize(trajectories[:, :, tran_dim + 1:tran_dim + 2], 'terminals')
        else:
            rewards = np.zeros(trajectories.shape[:-1] + (1,))
            terminals = np.zeros(trajectories.shape[:-1] + (1,))
        return action, Trajectories(actions, observations, rewards, terminals, samples.values, samples.costs)

    def _format_conditions(self, conditions, batch_size):                                       # policies.py:70-82
        out = {}
        for k, v in conditions.items():
            v = self.normalizer.normalize(np.asarray(v), 'observations')
            v = torch.as_tensor(v, dtype=torch.float32, device=self.device)
            # einops 'd -> repeat d' / 'd w -> (repeat d) w': the repeat index is the OUTER one
            out[k] = v[None].repeat(batch_size, 1) if v.dim() == 1 else v.repeat(batch_size, 1)
        return out
