"""Diagnostics: the captured planning call repeated with the same noise -- bitwise identical results, stable memory.
usage: soak.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device('cuda:0')
diff, guide, cguide, sds = bench.build_models(dev)
B, H, D = 64, bench.H, bench.OBS + bench.ACT
obs = torch.zeros(bench.OBS, device=dev)
noise = torch.randn(bench.T + 1, B, H, D, device=dev, generator=torch.Generator(dev).manual_seed(3))
kw = dict(bench.PLAN_KW, guide=guide, cost_guide=cguide)
def call():
    x, v, c, _ = diff.sample_raw((B, H, D), {0: obs[None].expand(B, -1)}, noise=noise, sample_fn=fn, cost_threshold=0.1, **kw)
    return x.clone(), v.clone(), c.clone()
x0, v0, c0 = call()
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
bad = 0
t0 = time.perf_counter()
for i in range(n):
    x, v, c = call()
    if not (torch.equal(x, x0) and torch.equal(v, v0) and torch.equal(c, c0)):
        bad += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'{n} calls, {bad} differ from the first bitwise; {1e3 * dt / n:.2f} ms per call (with the comparisons); '
      f'allocated {m0} -> {torch.cuda.memory_allocated()} bytes; finite: {bool(torch.isfinite(x0).all())}')
