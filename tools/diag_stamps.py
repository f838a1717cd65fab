"""Diagnostic only: per-phase cycle stamps of the conv kernel, layer by layer (debug build of the library
with -DTREBI_STAMPS; see csrc/kernels.hip STAMP).  Never part of a timed run."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
import numpy as np, torch
from trebi_amd import _lib, spec, synth
_lib.LIB_PATH = os.environ.get('TREBI_DBG_LIB') or os.path.join(ROOT, 'tools', 'dbg', 'libtrebi_hip.so')
from trebi_amd.nets import TemporalUnet, ValueFunction
H, D, B = 32, 23, int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind = sys.argv[2] if len(sys.argv) > 2 else 'value'
if kind == 'value':
    net = ValueFunction(H, D, 0); shapes = spec.value_param_shapes(H, D)
else:
    net = TemporalUnet(H, D, 0); shapes = spec.unet_param_shapes(H, D)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net.set_max_timesteps(20); net = net.cuda()
x = torch.randn(B, H, D, device='cuda'); t = torch.full((B,), 5, device='cuda', dtype=torch.long)
lib = _lib.lib()
lib.trebi_debug_read_stamps.argtypes = [C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 32)()
run = (lambda: net.value_and_input_grad(x, t)) if kind == 'value' else (lambda: net(x, None, t))
nops = int(sys.argv[3]) if len(sys.argv) > 3 else (60 if kind == 'value' else 45)
print('op  total_cyc  us(100MHz clk)  clkGHz | phases: preload stage wait_bar prologue kloop bar+partials epilogue')
for n in range(nops):
    os.environ['TREBI_STOP_AFTER'] = str(n)
    for _ in range(3):
        run(); torch.cuda.synchronize()
    lib.trebi_debug_read_stamps(buf)
    s = list(buf)
    cyc = [s[i + 1] - s[i] for i in range(6)] + [s[9] - s[6]]
    tot = s[9] - s[0]; real = (s[25] - s[16]) / 100.0
    print(f'{n:3d} {tot:8d} {real:7.2f} {tot / max(real, 1e-9) / 1e3:5.2f} |', ' '.join(f'{c:6d}' for c in cyc))
