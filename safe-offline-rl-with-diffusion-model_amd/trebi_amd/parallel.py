"""Candidate sharding across the GPUs of one node.

Candidates are independent through all denoise steps (GroupNorm is per sample), so each rank
runs the captured loop on its contiguous slice of the global batch with NO communication.  The
only global step is the feasibility rule + argmax at the end (functions.py:126-129,
diffusion.py:44-49): one all-gather of 2*B_local floats (values, costs) over RCCL/xGMI --
latency-bound (<= 64 KiB at 8192 candidates per GPU) -- after which every rank applies the rule
and the sort to the global arrays; the owner of the winning candidate broadcasts its [H, D]
trajectory.  One process per GPU; backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import os
from collections import namedtuple

import torch
import torch.distributed as dist

Plan = namedtuple('Plan', 'best_index best_trajectory values costs order local_trajectories local_slice')


def shard_bounds(n_global, world_size, rank):
    """Contiguous, balanced slices: the first n_global % world_size ranks get one extra candidate."""
    q, r = divmod(n_global, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class ShardedPlanner:
    def __init__(self, diffusion, group=None, local_fn=None, rule_fn=None, argsort_fn=None):
        """`local_fn(shape, cond, noise=..., **kw) -> (x, values_raw, costs)` defaults to the fused HIP loop;
        `rule_fn` / `argsort_fn` default to the library's selection kernels.  The hooks exist so the
        sharding logic can be exercised on CPU ranks (gloo) where there is no GPU."""
        self.diffusion = diffusion
        self.group = group
        if local_fn is None:
            def local_fn(shape, cond, noise=None, **kw):
                x, v, c, _ = diffusion.sample_raw(shape, cond, noise=noise, **kw)
                return x, v, c
        if rule_fn is None or argsort_fn is None:
            from . import diffusion as D
            rule_fn = rule_fn or D.feasibility_rule
            argsort_fn = argsort_fn or D.argsort_descending
        self.local_fn, self.rule_fn, self.argsort_fn = local_fn, rule_fn, argsort_fn

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def _all_gather_ragged(self, t, sizes):
        """all_gather of 1-D tensors whose length differs by at most one across ranks (pad to the max)."""
        world, _ = self._world()
        if world == 1:
            return t
        m = max(sizes)
        pad = torch.zeros(m, dtype=t.dtype, device=t.device)
        pad[:t.numel()] = t
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad, group=self.group)
        return torch.cat([o[:n] for o, n in zip(outs, sizes)])

    @torch.no_grad()
    def plan(self, observation, n_candidates, horizon, sample_fn, cost_threshold=None, noise=None,
             constrained=True, **sample_kwargs):
        """observation: [obs_dim] normalised observation shared by all candidates (policies.py:77-82);
        noise: optional GLOBAL [T+1, n_candidates, H, D] tensor (each rank takes its slice)."""
        world, rank = self._world()
        lo, hi = shard_bounds(n_candidates, world, rank)
        sizes = [shard_bounds(n_candidates, world, r)[1] - shard_bounds(n_candidates, world, r)[0]
                 for r in range(world)]
        B = hi - lo
        D = self.diffusion.transition_dim
        obs = torch.as_tensor(observation, dtype=torch.float32)
        cond = {0: obs[None].expand(B, -1)}
        kw = dict(sample_kwargs)
        if constrained:
            thr = torch.as_tensor(cost_threshold, dtype=torch.float32)
            kw['cost_threshold'] = thr if thr.dim() == 0 else thr[lo:hi]
        nz = None if noise is None else noise[:, lo:hi]
        x, v_raw, c = self.local_fn((B, horizon, D), cond, noise=nz, sample_fn=sample_fn, **kw)
        # ---- the one exchange: per-candidate scores ----
        packed = torch.cat([v_raw.reshape(-1), c.reshape(-1)])
        if world > 1:
            m = max(sizes)
            pad = torch.zeros(2 * m, dtype=packed.dtype, device=packed.device)
            pad[:B] = v_raw.reshape(-1)
            pad[m:m + B] = c.reshape(-1)
            if pad.is_cuda and os.environ.get('TREBI_PLAN_SYNC', '1') == '1':
                # the scores are needed on the host right after the exchange anyway; draining the launch stream first
                # keeps the backend's side stream from queueing behind the captured loop (two ranks sharing one GPU
                # over gloo stalled there; see tools/rehearse_ranks.py)
                torch.cuda.current_stream(pad.device).synchronize()
            outs = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(outs, pad, group=self.group)          # single fused all-gather (values | costs)
            v_all = torch.cat([o[:n] for o, n in zip(outs, sizes)])
            c_all = torch.cat([o[m:m + n] for o, n in zip(outs, sizes)])
        else:
            v_all, c_all = v_raw.reshape(-1), c.reshape(-1)
        if constrained:
            thr = torch.as_tensor(cost_threshold, dtype=torch.float32, device=v_all.device)
            thr_all = thr.expand(n_candidates) if thr.dim() == 0 else thr
            v_all = self.rule_fn(v_all, c_all, thr_all)
        order = self.argsort_fn(v_all)
        best = int(order[0])
        owner = next(r for r in range(world) if shard_bounds(n_candidates, world, r)[0] <= best
                     < shard_bounds(n_candidates, world, r)[1])
        traj = x[best - lo].clone() if rank == owner else torch.empty(horizon, D, dtype=x.dtype, device=x.device)
        if world > 1:
            src = dist.get_global_rank(self.group, owner) if self.group is not None else owner
            dist.broadcast(traj, src=src, group=self.group)
        return Plan(best, traj, v_all, c_all, order, x, (lo, hi))
