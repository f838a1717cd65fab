"""Diagnostic only (-DTREBI_STAMPS build in tools/dbg/): per-WAVE cycle stamps of workgroup (0,0,0) of the conv kernel for chosen
layers: when each wave leaves staging, the K loop, the exchange and the epilogue.
usage: diag_wave_stamps.py <batch> <value|unet> <op> [<op> ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
import numpy as np, torch
from trebi_amd import _lib, spec, synth
_lib.LIB_PATH = os.environ.get('TREBI_DBG_LIB') or os.path.join(ROOT, 'tools', 'dbg', 'libtrebi_hip.so')
from trebi_amd.nets import TemporalUnet, ValueFunction
H, D, B = 32, 23, int(sys.argv[1])
kind = sys.argv[2]
ops = [int(v) for v in sys.argv[3:]]
if kind == 'value':
    net = ValueFunction(H, D, 0); shapes = spec.value_param_shapes(H, D)
else:
    net = TemporalUnet(H, D, 0); shapes = spec.unet_param_shapes(H, D)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 1).items()})
net.set_max_timesteps(20); net = net.cuda()
x = torch.randn(B, H, D, device='cuda'); t = torch.full((B,), 5, device='cuda', dtype=torch.long)
lib = _lib.lib()
lib.trebi_debug_read_wave_stamps.argtypes = [C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 128)()
run = (lambda: net.value_and_input_grad(x, t)) if kind == 'value' else (lambda: net(x, None, t))
for n in ops:
    os.environ['TREBI_STOP_AFTER'] = str(n)
    for _ in range(3):
        run(); torch.cuda.synchronize()
    lib.trebi_debug_read_wave_stamps(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 16).astype(np.int64)
    t0 = a[:, 0].min()
    print(f'op {n}: cycles since the first wave started; columns: start, weights requested, staged, past barrier, prologue done, K loop done, '
          f'exchange done, end')
    for w in range(8):
        s = a[w]
        print(f'  wave {w}: ' + ' '.join(f'{int(s[i] - t0):8d}' for i in (0, 1, 2, 3, 4, 5, 6, 9)) + f' | K loop {int(s[5] - s[4]):7d}')
