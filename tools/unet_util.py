"""TemporalUnet forward alone at large batches: achieved fp32 matrix rate (algorithmic FLOPs, zero-padding taps
excluded) against the 157.3 TFLOP/s peak -- north_star's ">= 40 % MFMA utilisation in the TemporalUnet conv kernels".
usage: unet_util.py [batch] [dim_mults e.g. 1,2,4,8] [iters] [f32|bf16x3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
import torch
from trebi_amd import spec, synth
from trebi_amd.nets import TemporalUnet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mults = tuple(int(v) for v in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 2, 4, 8)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prec = sys.argv[4] if len(sys.argv) > 4 else 'f32'
H, D = 32, 23
net = TemporalUnet(H, D, 0, dim=32, dim_mults=mults)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec.unet_param_shapes(H, D, 32, mults), 1).items()})
net.set_max_timesteps(20); net.set_precision(prec); net = net.cuda()
x = torch.randn(B, H, D, device='cuda'); t = 5
for _ in range(3):
    net(x, None, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    net(x, None, t)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = net.flops() * B
print(f'unet {mults} B={B} {prec}: {ms:.3f} ms per forward, {fl / ms / 1e9:.1f} TFLOP/s algorithmic = {fl / ms / 1e9 / 157.3 * 100:.1f} % of fp32 matrix peak')
