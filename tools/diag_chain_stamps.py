"""Diagnostic only: per-stage, per-phase cycle stamps of the resident chain kernel (debug build of the
library with -DTREBI_STAMPS in tools/dbg/).  Never part of a timed run."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
sys.path.insert(0, ROOT)
import numpy as np, torch
from trebi_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'dbg', 'libtrebi_hip.so')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
diff, guide, cguide, sds = bench.build_models(dev)
from diffuser.sampling import n_step_guided_and_constrained_p_sample as fn
obs = torch.zeros(bench.OBS, device=dev)
kw = dict(bench.PLAN_KW, guide=guide, cost_guide=cguide)
for _ in range(2):
    diff.sample_raw((B, bench.H, bench.OBS + bench.ACT), {0: obs[None].expand(B, -1)}, sample_fn=fn, cost_threshold=0.1, **kw)
torch.cuda.synchronize()
lib = _lib.lib()
n = 96 * 48 * 16
buf = (C.c_uint64 * n)()
lib.trebi_debug_read_cstamps.argtypes = [C.POINTER(C.c_uint64)]
lib.trebi_debug_read_cstamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(96, 48, 16).astype(np.int64)
print('phases: A(stage/loads) bar+prologue kloop partials+bar epilogue endbar | total')
for key in range(96):
    if a[key].any():
        ns, nz = key // 2, key % 2 + 1
        print(f'== chain of {ns} stages, nets per launch {nz}')
        for i in range(ns):
            s = a[key, i]
            d = [s[k + 1] - s[k] for k in range(6)]
            print(f'{i:3d}', ' '.join(f'{x:6d}' for x in d), '|', s[6] - s[0], '| A: pins', s[7] - s[0], 'pro-ld', s[8] - s[7], 'col-ld', s[9] - s[8], 'stage', s[10] - s[9], 'halo', s[1] - s[10], '| epi: stats', s[11] - s[4], 'xbar', s[12] - s[11], 'rest', s[5] - s[12])
