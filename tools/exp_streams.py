"""Experiment: G independent sub-batches, each a captured loop on its own stream, vs one batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
import torch, numpy as np
import bench
from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
dev = torch.device('cuda:0')
H, D, OBS = 32, 23, 17
for G in (1, 2, 4, 8):
    Bg = 64 // G
    models = [bench.build_models(dev)[:3] for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    obs = torch.randn(OBS, device=dev)
    kw = dict(bench.PLAN_KW)
    def run():
        outs = []
        for g in range(G):
            diff, guide, cguide = models[g]
            with torch.cuda.stream(streams[g]):
                outs.append(diff.sample_raw((Bg, H, D), {0: obs[None].expand(Bg, -1)}, sample_fn=fn, cost_threshold=0.0,
                                            guide=guide, cost_guide=cguide, **kw))
        return outs
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'G={G} Bg={Bg}: {dt*1e3:.2f} ms per 64 candidates -> {64/dt:.0f} traj/s', flush=True)
