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
    def __init__(self, diffusion, group=None, local_fn=None, rule_fn=None, argsort_fn=None, fence=None):
        """`local_fn(shape, cond, noise=..., **kw) -> (x, values_raw, costs)` defaults to the fused HIP loop;
        `rule_fn` / `argsort_fn` default to the library's selection kernels.  The hooks exist so the
        sharding logic can be exercised on CPU ranks (gloo) where there is no GPU."""
        self.diffusion = diffusion
        self.group = group
        self._bufs = {}
        # How the exchange is ordered behind the captured loop.  'host' (default, every backend): drain the launch
        # stream first -- free, the scores are read on the host right after the exchange.  'stream' (TREBI_PLAN_SYNC=0):
        # leave the ordering to the backend's own event wait.  Measured on a one-GPU box with two gloo ranks
        # (tools/rehearse_two_samplers.py, profiles/r02_stall_diagnostic.txt): with 'stream' the FIRST exchange queued
        # behind a pending hipGraphLaunch never completes (gloo records an event behind the graph, lets a pool stream
        # wait for it and copies to the host from there); with 'host' 24 of 24 exchanges pass.  RCCL could not be run
        # on that box (one GPU), so the safe fence is the default for it too.
        if fence is None:
            fence = 'stream' if os.environ.get('TREBI_PLAN_SYNC') == '0' else 'host'
        self.fence = fence
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

    def _buffers(self, world, m, like):
        """Exchange buffers, allocated once per (world, shard size, device): send [2, m] = (values | costs) padded to
        the largest shard, receive [world, 2, m]."""
        key = (world, m, str(like.device), like.dtype)
        if self._bufs.get('key') != key:
            self._bufs = dict(key=key, send=torch.zeros(2, m, dtype=like.dtype, device=like.device),
                              recv=torch.empty(world, 2, m, dtype=like.dtype, device=like.device))
        return self._bufs['send'], self._bufs['recv']

    @torch.no_grad()
    def plan(self, observation, n_candidates, horizon, sample_fn, cost_threshold=None, noise=None,
             constrained=True, copy=True, **sample_kwargs):
        """observation: [obs_dim] normalised observation shared by all candidates (policies.py:77-82);
        noise: optional GLOBAL [T+1, n_candidates, H, D] tensor (each rank takes its slice).
        copy=False returns views of the sampler's persistent buffers in `local_trajectories` / `costs` (the next
        plan() overwrites them; the reference returns fresh tensors, which is what the default does)."""
        world, rank = self._world()
        bounds = [shard_bounds(n_candidates, world, r) for r in range(world)]
        sizes = [hi - lo for lo, hi in bounds]
        lo, hi = bounds[rank]
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
        # ---- the one exchange: per-candidate scores, ONE all-gather of (values | costs) ----
        if world > 1:
            m = max(sizes)
            send, recv = self._buffers(world, m, v_raw)
            send[0, :B].copy_(v_raw.reshape(-1))
            send[1, :B].copy_(c.reshape(-1))
            if send.is_cuda and self.fence == 'host':
                # drain the launch stream first (free: the scores are read on the host right after the exchange).
                # Needed with gloo, whose device-tensor path stages through the host; RCCL enqueues its kernel on a
                # stream that waits for the current stream through an event (fence 'stream': nothing to do here).
                torch.cuda.current_stream(send.device).synchronize()
            if send.is_cuda and dist.get_backend(self.group) == 'gloo':
                # gloo moves device tensors through its own staging copies; with two processes on one device (the one-GPU
                # rehearsal of bench --gpus N) that path stalled intermittently.  The scores are 2 x m floats: exchange
                # them from the host instead.  RCCL (backend 'nccl', one process per GPU) takes the device buffers.
                send_h = send.cpu()
                recv_h = torch.empty(recv.shape, dtype=recv.dtype)
                dist.all_gather_into_tensor(recv_h.view(-1), send_h.view(-1), group=self.group)
                recv.copy_(recv_h)
            else:
                dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
            if min(sizes) == m:
                v_all, c_all = recv[:, 0, :].reshape(-1), recv[:, 1, :].reshape(-1)
            else:
                v_all = torch.cat([recv[r, 0, :n] for r, n in enumerate(sizes)])
                c_all = torch.cat([recv[r, 1, :n] for r, n in enumerate(sizes)])
        else:
            v_all, c_all = v_raw.reshape(-1), c.reshape(-1)
            if copy:
                v_all, c_all = v_all.clone(), c_all.clone()
        if constrained:
            thr = torch.as_tensor(cost_threshold, dtype=torch.float32, device=v_all.device)
            thr_all = thr.expand(n_candidates) if thr.dim() == 0 else thr
            v_all = self.rule_fn(v_all, c_all, thr_all)
        order = self.argsort_fn(v_all)
        best = int(order[0])
        owner = next(r for r, (a, b) in enumerate(bounds) if a <= best < b)
        traj = x[best - lo].clone() if rank == owner else torch.empty(horizon, D, dtype=x.dtype, device=x.device)
        if world > 1:
            src = dist.get_global_rank(self.group, owner) if self.group is not None else owner
            if traj.is_cuda and dist.get_backend(self.group) == 'gloo':      # as above: host tensors under gloo
                traj_h = traj.cpu()
                dist.broadcast(traj_h, src=src, group=self.group)
                traj.copy_(traj_h)
            else:
                dist.broadcast(traj, src=src, group=self.group)
        return Plan(best, traj, v_all, c_all, order, x.clone() if copy else x, (lo, hi))
