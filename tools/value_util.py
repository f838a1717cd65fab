"""Diagnostics: ValueFunction forward + backward-to-input alone at large batches (the guide's gradients call).
usage: value_util.py [batch] [iters] [f32|bf16x3] [obs+act dim]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
import torch
from trebi_amd import spec, synth
from trebi_amd.nets import ValueFunction
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else 'f32'
D = int(sys.argv[4]) if len(sys.argv) > 4 else 23
H, mults = 32, (1, 2, 4, 8)
net = ValueFunction(H, D, 0, dim=32, dim_mults=mults)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec.value_param_shapes(H, D, 32, mults), 1).items()})
net.set_max_timesteps(20); net.set_precision(prec); net = net.cuda()
x = torch.randn(B, H, D, device='cuda'); t = torch.full((B,), 5, device='cuda', dtype=torch.long)
for _ in range(2):
    net.value_and_input_grad(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    net.value_and_input_grad(x, t)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = (net.flops() + net.flops(backward=True)) * B
print(f'value net B={B} {prec}: {ms:.3f} ms per fwd+bwd, {fl / ms / 1e9:.1f} TFLOP/s algorithmic = {fl / ms / 1e9 / 157.3 * 100:.1f} % of fp32 matrix peak')
