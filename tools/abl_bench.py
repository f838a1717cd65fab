"""Diagnostics: whole-call time of the headline workload for the library named by TREBI_LIB (ablated builds of
tools/abl_chain.sh give wrong results on purpose; only the time is read)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
from trebi_amd.parallel import ShardedPlanner
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
diff, guide, cguide, sds = bench.build_models(dev)
planner = ShardedPlanner(diff)
obs = torch.zeros(bench.OBS, device=dev)
kw = dict(bench.PLAN_KW, guide=guide, cost_guide=cguide)
run = lambda: planner.plan(obs, B, bench.H, fn, cost_threshold=0.1, copy=False, **kw)
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get('CALLS', '40'))
for _ in range(n):
    run()
torch.cuda.synchronize()
print(f"{os.environ.get('LABEL') or os.environ.get('TREBI_LIB', 'product')}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per call", flush=True)
