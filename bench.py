"""bench.py -- cost-guided trajectories/sec of the TREBI planning hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one complete planning call of BASELINE.json configs[1]: halfcheetah (obs 17 + act 6),
H=32, 20 denoise steps, 64 candidates per GPU, reward + cost guide (n_guide_steps=2, scale=0.001,
t_stopgrad=4, cost_grad_weight=5, guide_type=0; config/locomotion.py:153-203,305-309): noise draw,
the captured loop (20 x (2 x [cost+reward ValueFunction forward+backward, guide update] + TemporalUnet
forward + DDPM update)), then the all-gather of (values, costs), feasibility rule and argsort.  Weights
are seeded random (no pretrained checkpoints offline), data is synthetic.  Candidates are sharded over
ranks with no collective inside the loop (weak scaling: 64 candidates per GPU).

Prints ONE JSON line on rank 0 (see the "Measurement" section of DESIGN.md for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, T, OBS, ACT, DIM = 32, 20, 17, 6, 32
MULTS = (1, 2, 4, 8)
PLAN_KW = dict(scale=0.001, t_stopgrad=4, n_guide_steps=2, cost_grad_weight=5.0, guide_type=0)
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 matrix peak (no sparsity)


def gflop_per_trajectory(F_unet, F_vf_fwd, F_vf_bwd, n_guide_steps, t_stopgrad, n_nets=2):
    """(nominal, executed) algorithmic GFLOP per trajectory of one planning call.  Nominal = SURVEY.md 8d's formula
    T x (F_unet + n_nets x n_guide_steps x (F_fwd + F_bwd)); executed = the same minus the value-net backward passes of
    the denoise steps with t < t_stopgrad, whose gradient the reference zeroes by assignment (functions.py:116) and the
    captured loop therefore does not compute (enqueue_loop in csrc/trebi.hip).  Every `frac` in the line uses EXECUTED."""
    nominal = T * (F_unet + n_nets * n_guide_steps * (F_vf_fwd + F_vf_bwd))
    executed = nominal - min(t_stopgrad, T) * n_nets * n_guide_steps * F_vf_bwd
    return nominal / 1e9, executed / 1e9


def _coll_device(device):
    """Tensors of the bench's own tiny collectives: on the GPU for RCCL (it needs device tensors), on the host for gloo
    (the one-GPU rehearsal; gloo's device-tensor path stalled intermittently with two processes on one GPU)."""
    return device if dist.get_backend() == 'nccl' else torch.device('cpu')


def _max_over_ranks(v, device):
    td = torch.tensor([v], device=_coll_device(device), dtype=torch.float64)
    dist.all_reduce(td, op=dist.ReduceOp.MAX)
    return float(td)


def _from_rank0(v, device):
    tt = torch.tensor([float(v)], device=_coll_device(device), dtype=torch.float64)
    dist.broadcast(tt, 0)
    return float(tt)


def build_models(device):
    from trebi_amd import spec, synth
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide
    D = OBS + ACT

    def mk(cls, shapes, seed):
        net = cls(H, D, 0, dim=DIM, dim_mults=MULTS)
        sd = synth.synth_state_dict(shapes, seed)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        return net, sd
    kw = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)
    unet, sd_u = mk(TemporalUnet, spec.unet_param_shapes(H, D, DIM, MULTS), 100)
    rnet, sd_r = mk(ValueFunction, spec.value_param_shapes(H, D, DIM, MULTS), 101)
    cnet, sd_c = mk(ValueFunction, spec.value_param_shapes(H, D, DIM, MULTS), 102)
    diff = GaussianDiffusion(unet, H, OBS, ACT, **kw).to(device)
    guide = ValueGuide(ValueDiffusion(rnet, H, OBS, ACT, **kw)).to(device)
    cguide = ValueGuide(ValueDiffusion(cnet, H, OBS, ACT, **kw)).to(device)
    return diff, guide, cguide, (sd_u, sd_r, sd_c)


def other_configs(device, steps=3, world=1, rank=0, rehearse=False, only=None, bf16=True):
    """BASELINE configs 3 and 4, whole planning calls through the same ShardedPlanner path, synthetic seeded weights.
    One GPU: at their PER-GPU size (512 and 8192 candidates).  world > 1: at their GLOBAL size (4096 and 65536 candidates)
    sharded over the ranks -- strong scaling, with the score all-gather in the loop.  Reported next to the headline, not as `value`."""
    from trebi_amd import spec, synth
    from trebi_amd.parallel import ShardedPlanner, shard_bounds
    from diffuser.models import TemporalUnet, ValueFunction, GaussianDiffusion, ValueDiffusion
    from diffuser.sampling import ValueGuide, n_step_guided_and_constrained_p_sample as fn
    out = []
    for idx, (name, obs, act, um, vm, Bgpu, Bglobal, kw) in enumerate((
            ('configs[2]: hopper-medium-v2 shapes, 4096 candidates / 8 GPUs', 11, 3, (1, 2, 4, 8), (1, 2, 4, 8), 512, 4096,
             dict(scale=0.1, t_stopgrad=2, n_guide_steps=2, cost_grad_weight=5.0, guide_type=0)),
            ('configs[3]: walker2d shapes, UNet dim_mults (1,4,8), 65536 candidates / 8 GPUs', 17, 6, (1, 4, 8),
             (1, 2, 4, 8), 8192, 65536, dict(scale=0.1, t_stopgrad=2, n_guide_steps=2, cost_grad_weight=5.0, guide_type=0)))):
        if only is not None and idx != only:
            continue
        D = obs + act
        n_global = Bgpu if world == 1 else (min(Bglobal, 8192) if rehearse else Bglobal)
        lo, hi = shard_bounds(n_global, world, rank)
        B = hi - lo

        def mk(cls, shapes, seed, mults):
            net = cls(H, D, 0, dim=DIM, dim_mults=mults)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed).items()})
            return net
        dkw = dict(n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False)
        unet = mk(TemporalUnet, spec.unet_param_shapes(H, D, DIM, um), 100, um)
        diff = GaussianDiffusion(unet, H, obs, act, **dkw).to(device)
        guide = ValueGuide(ValueDiffusion(mk(ValueFunction, spec.value_param_shapes(H, D, DIM, vm), 101, vm), H, obs, act,
                                          **dkw)).to(device)
        cguide = ValueGuide(ValueDiffusion(mk(ValueFunction, spec.value_param_shapes(H, D, DIM, vm), 102, vm), H, obs, act,
                                           **dkw)).to(device)
        o = torch.from_numpy(np.random.default_rng(1).standard_normal(obs).astype(np.float32)).to(device)
        nb = min(max(B, 1), 512)
        x_T = torch.randn(nb, H, D, device=device, generator=torch.Generator(device).manual_seed(2))
        x_T[:, 0, act:] = o
        tt = cguide(x_T, None, torch.full((nb,), T - 1, device=device, dtype=torch.long)).median().reshape(1)
        thr = float(tt)
        if world > 1:
            thr = _from_rank0(thr, device)
        planner = ShardedPlanner(diff)
        run = lambda: planner.plan(o, n_global, H, fn, cost_threshold=thr, guide=guide, cost_guide=cguide, copy=False, **kw)

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(device)
        run()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        fence()
        dt = (time.perf_counter() - t0) / steps
        if world > 1:
            dt = _max_over_ranks(dt, device)
        g_nom, g_exe = gflop_per_trajectory(unet.flops(), guide._net().flops(), guide._net().flops(backward=True),
                                            kw['n_guide_steps'], kw['t_stopgrad'])
        fl = g_exe * 1e9
        tf = fl * n_global / dt / 1e12
        entry = {'workload': f'{name}: {n_global} candidates over {world} GPU(s), {B} on this rank',
                 'trajectories_per_s': n_global / dt, 'ms_per_step': 1e3 * dt, 'scaling': 'strong' if world > 1 else 'per-GPU size',
                 'algorithmic_gflop_per_trajectory': g_nom, 'executed_gflop_per_trajectory': g_exe,
                 'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS * world, 'unit': 'TFLOP/s (executed FLOPs)',
                 'frac': tf / (PEAK_F32_MFMA_TFLOPS * world)}
        if not bf16:                       # profiling runs of one config (tools/bench_other.py): the fp32 path only
            out.append(entry)
            continue
        # the same call with the opt-in bf16x3 operands (engaged from 512 candidates per launch up), reported apart
        for net in (unet, guide._net(), cguide._net()):
            net.set_precision('bf16x3')
        run()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        fence()
        dt = (time.perf_counter() - t0) / steps
        if world > 1:
            dt = _max_over_ranks(dt, device)
        tf = fl * n_global / dt / 1e12
        entry['bf16x3'] = {'trajectories_per_s': n_global / dt, 'ms_per_step': 1e3 * dt, 'achieved': tf,
                           'peak': PEAK_BF16_MFMA_TFLOPS * world, 'unit': 'TFLOP/s (algorithmic)',
                           'frac': tf / (PEAK_BF16_MFMA_TFLOPS * world),
                           'frac_of_f32_matrix_peak': tf / (PEAK_F32_MFMA_TFLOPS * world)}
        out.append(entry)
        del planner, diff, guide, cguide, unet
        torch.cuda.empty_cache()
    return out


def unet_utilisation(unet, device, batch=8192, iters=10):
    """north_star's second target: matrix-core utilisation of the TemporalUnet conv kernels where they are not
    latency-bound -- the UNet forward alone at config 4's per-GPU size (8192 candidates), algorithmic FLOPs (zero-padding
    taps excluded) over HIP-event time on the launch stream, against the fp32 matrix peak."""
    x = torch.randn(batch, H, OBS + ACT, device=device)
    t = 5                                  # a Python int: range-checked on the host, no device read-back per call
    for _ in range(2):
        unet(x, None, t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        unet(x, None, t)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    tf = unet.flops() * batch / (ms * 1e-3) / 1e12
    out = {'workload': f'TemporalUnet forward alone, {batch} candidates, dim_mults {tuple(MULTS)}', 'ms': ms,
           'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_F32_MFMA_TFLOPS}
    # the opt-in bf16x3 operand mode (3 bf16 MFMAs per product, fp32 accumulate; 6e-5 of the reference), reported apart:
    # algorithmic FLOPs (one per multiply-add, not three) against the bf16 dense matrix peak
    unet.set_precision('bf16x3')
    for _ in range(2):
        unet(x, None, t)
    e0.record()
    for _ in range(iters):
        unet(x, None, t)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    tf = unet.flops() * batch / (ms * 1e-3) / 1e12
    out['bf16x3'] = {'ms': ms, 'achieved': tf, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s (algorithmic; 3 bf16 MFMA flops each)',
                     'frac': tf / PEAK_BF16_MFMA_TFLOPS, 'frac_of_f32_matrix_peak': tf / PEAK_F32_MFMA_TFLOPS}
    unet.set_precision('f32')
    return out


def train_step_extra(device, world, rank, batch=4096, accum=2, steps=4, cpu=True):
    """BASELINE configs[4]: the denoiser's training step (halfcheetah shapes, micro-batch 4096, gradient_accumulate_every 2 as in
    config/locomotion.py:49-50) -- q_sample, TemporalUnet forward, weighted L2, full backward, Adam, EMA.  Data-parallel under
    world > 1: the GLOBAL micro-batch stays 4096 (strong scaling), each rank takes 4096 / world samples, one all-reduce of the
    flat gradient per optimiser step.  Reported next to the headline, never as `value`."""
    from trebi_amd import spec, synth
    from trebi_amd.training import Trainer
    from diffuser.models import TemporalUnet, GaussianDiffusion
    D = OBS + ACT
    unet = TemporalUnet(H, D, 0, dim=DIM, dim_mults=MULTS)
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, D, DIM, MULTS), 100)
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    diff = GaussianDiffusion(unet, H, OBS, ACT, n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False,
                             action_weight=10.0).to(device)
    from trebi_amd.parallel import shard_bounds
    lo, hi = shard_bounds(batch, world, rank)
    Bl = hi - lo
    tr = Trainer(diff, train_batch_size=max(Bl, 1), train_lr=2e-4, gradient_accumulate_every=accum, distributed=world > 1)
    g = torch.Generator(device).manual_seed(rank)
    data = []
    for _ in range(accum):
        x = torch.randn(Bl, H, D, device=device, generator=g)
        data.append((x, {0: x[:, 0, ACT:].contiguous()}))
    sizes = [batch] * accum

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
    for _ in range(2):
        tr.train_step(data, global_batch_sizes=sizes)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.train_step(data, global_batch_sizes=sizes)     # t ~ randint, noise ~ randn drawn inside, like loss()
    fence()
    dt = (time.perf_counter() - t0) / steps
    if world > 1:
        dt = _max_over_ranks(dt, device)
    fl = (tr.engine.flops() + tr.engine.flops(backward=True)) * batch * accum
    out = {'workload': f'configs[4]: halfcheetah diffusion training step, global micro-batch {batch} x {accum} accumulations per '
                       f'optimiser step, data-parallel x{world} ({Bl} samples per rank and micro-batch)',
           'samples_per_s': batch * accum / dt, 'ms_per_optimiser_step': 1e3 * dt, 'achieved': fl / dt / 1e12,
           'peak': PEAK_F32_MFMA_TFLOPS * world, 'unit': 'TFLOP/s', 'frac': fl / dt / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
           'algorithmic_mflop_per_sample': {'forward': tr.engine.flops() / 1e6, 'backward': tr.engine.flops(backward=True) / 1e6},
           'loss': float(losses[-1][0]), 'scaling': 'strong', 'dtype': 'f32'}
    tr.engine.set_precision('bf16x3')          # opt-in operands for the forward / input-gradient convolutions, reported apart
    for _ in range(2):
        tr.train_step(data, global_batch_sizes=sizes)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(data, global_batch_sizes=sizes)
    fence()
    dtb = (time.perf_counter() - t0) / steps
    if world > 1:
        dtb = _max_over_ranks(dtb, device)
    out['bf16x3'] = {'samples_per_s': batch * accum / dtb, 'ms_per_optimiser_step': 1e3 * dtb, 'achieved': fl / dtb / 1e12,
                     'unit': 'TFLOP/s (algorithmic)', 'frac_of_f32_matrix_peak': fl / dtb / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
                     'note': 'forward and input-gradient convolutions on bf16x3 operands; weight gradients, loss, Adam, EMA fp32'}
    if cpu and rank == 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import trebi_train_oracle as TO
        Bc = 256
        sdc = {k: v.copy() for k, v in sd.items()}
        ema = {k: v.copy() for k, v in sd.items()}
        opt = TO.Adam(lr=2e-4)
        rng = np.random.default_rng(0)
        times = []
        for step in range(3):
            bs = []
            for _ in range(accum):
                x = rng.standard_normal((Bc, H, D)).astype(np.float32)
                bs.append((x, {0: x[:, 0, ACT:].copy()}, rng.integers(0, T, size=Bc), rng.standard_normal((Bc, H, D)).astype(np.float32)))
            t0 = time.perf_counter()
            TO.train_step(sdc, ema, opt, bs, H, OBS, ACT, T, step)
            times.append(time.perf_counter() - t0)
        try:
            from threadpoolctl import threadpool_info
            cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
        except Exception:
            cores = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
        out['cpu_baseline'] = {'value': Bc * accum / sorted(times)[1], 'unit': 'samples/s', 'cores': int(cores), 'kind': 'port',
                               'sample': f'3 optimiser steps of {accum} x {Bc} samples through oracle/trebi_train_oracle.py '
                                         '(numpy/OpenBLAS), median'}
    del tr, diff
    torch.cuda.empty_cache()
    return out


def cpu_baseline(sds, B, thr, runs=5, big=512):
    """The numpy oracle (a port of the reference path, oracle/trebi_oracle.py) on this box's host cores: the same workload
    (B candidates, full 20-step constrained loop), median of `runs` calls, plus ONE call at `big` candidates
    (the CPU path amortises its per-op overhead there); bounded to about half a minute of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import trebi_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    D = OBS + ACT
    u = O.TemporalUnet(sds[0], H, D, DIM, MULTS)
    r = O.ValueFunction(sds[1], H, D, DIM, MULTS)
    c = O.ValueFunction(sds[2], H, D, DIM, MULTS)
    diff = O.GaussianDiffusion(u, H, OBS, ACT, T)
    rng = np.random.default_rng(0)

    def call(n):
        cond = {0: np.repeat(rng.standard_normal((1, OBS)).astype(np.float32), n, 0)}
        noise = rng.standard_normal((T + 1, n, H, D)).astype(np.float32)
        t0 = time.perf_counter()
        diff.p_sample_loop((n, H, D), cond, noise, O.n_step_guided_and_constrained_p_sample,
                           guide=O.ValueGuide(r), cost_guide=O.ValueGuide(c),
                           cost_threshold=np.full(n, thr, np.float32), **PLAN_KW)
        return time.perf_counter() - t0
    times = sorted(call(B) for _ in range(runs))
    med = times[len(times) // 2]
    out = dict(value=B / med, unit='trajectories/s', cores=int(cores), kind='port',
               sample=f'{runs} full planning calls of the same workload ({B} candidates, H={H}, {T} steps, '
                      f'reward+cost guide) through oracle/trebi_oracle.py (numpy/OpenBLAS), median')
    if big:
        out['at_larger_batch'] = {'candidates': big, 'value': big / call(big), 'unit': 'trajectories/s', 'sample': 'one call'}
    return out


def control_step_extra(diff, guide, cguide, thr, device, B, steps=20):
    """whole control steps through GuidedPolicy (policies.py:25-63): raw observation in (numpy), normalise,
    repeat, budget tensor, the captured loop, selection, un-normalised first action out (numpy) -- what the episode loop of
    scripts/plan_cost_guided.py:208-210 calls once per environment step."""
    from diffuser.sampling import GuidedPolicy, n_step_guided_and_constrained_p_sample as fn
    from trebi_amd.policy import GaussianNormalizer
    rng = np.random.default_rng(3)
    norm = GaussianNormalizer({'observations': (rng.standard_normal(OBS), 0.5 + rng.random(OBS)),
                               'actions': (rng.standard_normal(ACT), 0.5 + rng.random(ACT))})
    policy = GuidedPolicy(guide, diff, norm, [], sample_fn=fn, cost_guide=cguide, cost_threshold=thr, **PLAN_KW)
    obs = rng.standard_normal(OBS).astype(np.float32)
    for _ in range(3):
        policy({0: obs}, batch_size=B, verbose=False)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        action, _ = policy({0: obs + 0.01 * i}, batch_size=B, verbose=False, cost_threshold=thr * (1 + 0.001 * i))
    dt = (time.perf_counter() - t0) / steps          # the action is on the host when the call returns
    return {'workload': f'GuidedPolicy control step, {B} candidates (numpy observation in, numpy action out)',
            'control_steps_per_s': 1.0 / dt, 'ms_per_control_step': 1e3 * dt,
            'device_to_host_bytes_per_step': ACT * 4, 'host_to_device_bytes_per_step': (OBS + 1) * 4}


def variants(diff, planner, obs, n_global, guide, cguide, thr, device, world, steps=10):
    """SURVEY.md 8d config 2's other two runs on the headline shapes: n_guide_steps = 1, and the budget
    730.746 * ratio_of_maxthreshold (cost.py:32; with random-init nets every candidate is feasible there: the unpenalised
    branch only).  Reported next to the headline, never as `value`."""
    from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
    out = []
    for label, kw, th in (('n_guide_steps=1', dict(PLAN_KW, n_guide_steps=1), thr),
                          ('cost_threshold = 730.746 * 0.5 (MAX_COST_DISCOUNT_THRESHOLD x ratio_of_maxthreshold)', dict(PLAN_KW),
                           730.746 * 0.5)):
        run = lambda: planner.plan(obs, n_global, H, fn, cost_threshold=th, guide=guide, cost_guide=cguide, copy=False, **kw)
        for _ in range(2):
            run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        F_v = guide._net()
        g_nom, g_exe = gflop_per_trajectory(diff.model.flops(), F_v.flops(), F_v.flops(backward=True), kw['n_guide_steps'],
                                            kw['t_stopgrad'])
        tf = g_exe * 1e9 * n_global / dt / 1e12
        out.append({'variant': label, 'trajectories_per_s': n_global / dt, 'ms_per_step': 1e3 * dt,
                    'algorithmic_gflop_per_trajectory': g_nom, 'executed_gflop_per_trajectory': g_exe,
                    'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS * world, 'unit': 'TFLOP/s (executed FLOPs)',
                    'frac': tf / (PEAK_F32_MFMA_TFLOPS * world)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='candidates per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--guide-steps', type=int, default=PLAN_KW['n_guide_steps'],
                    help='n_guide_steps (headline 2; SURVEY 8d also asks for 1)')
    ap.add_argument('--no-unet-util', action='store_true', help='skip the large-batch UNet utilisation measurement')
    ap.add_argument('--no-train', action='store_true', help='skip the training-step extra (BASELINE configs[4])')
    ap.add_argument('--skip-dead-evals', action='store_true',
                    help='also drop guide evaluations whose result cannot reach an output (not the default)')
    args = ap.parse_args()

    PLAN_KW['n_guide_steps'] = args.guide_steps
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N with N > 1 must be launched through torch.distributed.run '
                     '(one process per GPU)')
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: there is no CPU path')
    # rehearsal on a one-GPU box: TREBI_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the driver's runs use one GPU per rank over RCCL
    rehearse = os.environ.get('TREBI_BENCH_REHEARSE') == '1'
    def phase(name):                          # rehearsal only: progress on stderr + a watchdog re-armed per phase
        if rehearse:
            import faulthandler
            faulthandler.dump_traceback_later(150, exit=True)  # a phase that hangs says where
            print(f'[rank {rank}] {name}', file=sys.stderr, flush=True)
    if rehearse:
        local = 0
        phase('start')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
    from trebi_amd.parallel import ShardedPlanner
    diff, guide, cguide, sds = build_models(device)
    planner = ShardedPlanner(diff)
    B = args.batch
    n_global = B * world
    obs = torch.from_numpy(np.random.default_rng(1).standard_normal(OBS).astype(np.float32)).to(device)
    D = OBS + ACT

    # synthetic budget: the median cost at t = T-1 of x_T, so about half the candidates take the
    # penalised branch (random-init nets give O(0.1) costs, not MuJoCo budgets)
    x_T = torch.randn(B, H, D, device=device, generator=torch.Generator(device).manual_seed(2))
    x_T[:, 0, ACT:] = obs
    c19 = cguide(x_T, None, torch.full((B,), T - 1, device=device, dtype=torch.long))
    thr = float(c19.median())
    if world > 1:
        thr = _from_rank0(thr, device)
    kw = dict(PLAN_KW, guide=guide, cost_guide=cguide, skip_dead_evals=args.skip_dead_evals)

    def step():
        return planner.plan(obs, n_global, H, fn, cost_threshold=thr, copy=False, **kw)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    phase('headline loop')
    for _ in range(max(args.warmup, 1)):
        plan = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = _max_over_ranks(dt, device)
    assert torch.isfinite(plan.best_trajectory).all()

    # ---- roofline of the dominant kernel (the implicit-GEMM conv), measured live with HIP events on
    # the launch stream: one profiled planning call with plain launches bracketed by events.
    phase('profiled call')
    lo, hi = plan.local_slice
    diff.sample_raw((hi - lo, H, D), {0: obs[None].expand(hi - lo, -1)}, sample_fn=fn, cost_threshold=thr,
                    profile=True, **kw)
    torch.cuda.synchronize(device)
    prof = diff.last_profile

    ctl = control_step_extra(diff, guide, cguide, thr, device, B) if (rank == 0 and world == 1 and not args.no_unet_util) else None
    phase('variants')
    var = None if args.no_unet_util else variants(diff, planner, obs, n_global, guide, cguide, thr, device, world)
    phase('training step')
    train = None if args.no_train else train_step_extra(device, world, rank, cpu=not args.no_cpu_baseline)
    # configs[2] / configs[3]: per-GPU sizes on one GPU; sharded at their global sizes under world > 1 (strong scaling)
    phase('other configs')
    others = None if args.no_unet_util else other_configs(device, world=world, rank=rank, rehearse=rehearse)

    phase('report')
    if rank == 0:
        gflop_nominal, gflop_executed = gflop_per_trajectory(diff.model.flops(), guide._net().flops(),
                                                             guide._net().flops(backward=True), PLAN_KW['n_guide_steps'],
                                                             PLAN_KW['t_stopgrad'])
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # figure is the committed rocprofv3 --pmc measurement of this same command (profiles/, tools/analyze_traffic.py)
        traffic = None
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_headline_hbm_traffic.json'))) or \
                sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_hbm_traffic.json')))
            with open(files[-1]) as f:                                             # newest round's headline-only run
                traffic = json.load(f)['conv_gemm_kernel']['hbm_bytes_per_launch']
        except Exception:
            pass
        avg_us = prof['gemm_us'] / max(prof['gemm_launches'], 1)
        flops_per_launch = prof['gemm_flops'] / max(prof['gemm_launches'], 1)
        achieved = flops_per_launch / (avg_us * 1e-6) / 1e12
        out = {
            'metric': 'cost-guided trajectories/sec (H=32, 20 denoise steps)',
            'value': n_global * args.steps / dt,
            'unit': 'trajectories/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic (seeded random weights, N(0,1) noise and observation)',
            'config': {'workload': 'halfcheetah-medium-expert-v2 shapes (obs 17 + act 6), H=32, T=20, '
                                   f'{B} candidates per GPU, reward+cost guide, n_guide_steps={args.guide_steps}, '
                                   'TemporalUnet/ValueFunction dim_mults (1,2,4,8)',
                       'candidates_per_gpu': B, 'global_candidates': n_global,
                       'parallelism': f'candidate-sharded x{world}', 'skip_dead_evals': bool(args.skip_dead_evals),
                       'algorithmic_gflop_per_trajectory': gflop_nominal,          # SURVEY.md 8d formula
                       'executed_gflop_per_trajectory': gflop_executed,            # what roofline.achieved counts
                       'whole_call_tflops_executed': gflop_executed * 1e9 * n_global * args.steps / dt / 1e12},
            'roofline': {'bound': 'mfma',
                         'kernel': 'implicit-GEMM Conv1d on f32 MFMA (conv_gemm_kernel + resident_chain_kernel)',
                         'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'traffic_note': 'bytes per conv_gemm_kernel launch leaving the L2s (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 '
                                         '--pmc, profiles/): fabric traffic, mostly served by the 256 MB Infinity Cache, not HBM.  '
                                         'Algorithmic ~1.3 MB (0.67 MB of weights for both nets + the candidate tile in and out); the '
                                         'grid is column-tile-major so that a weight slice is fetched by ONE XCD, which makes each of '
                                         'the 8 column tiles (8 XCDs, 8 L2s) read the 0.5 MB of input activations: ~0.7 + 8 x 0.5 MB',
                         'avg_launch_us': avg_us, 'launches_per_step': prof['gemm_launches'],
                         'algorithmic_flops_per_launch': flops_per_launch,
                         'eager_loop_ms': prof['loop_us'] / 1e3},
        }
        if world == 1 and not args.no_unet_util:
            out['unet_mfma'] = unet_utilisation(diff.model, device)
        if others is not None:
            out['other_configs'] = others
        if var is not None:
            out['variants'] = var
        if ctl is not None:
            out['control_step'] = ctl
        if train is not None:
            out['train_step'] = train
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sds, B, thr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
