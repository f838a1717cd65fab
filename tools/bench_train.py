"""Diagnostics: time the HIP training step (BASELINE configs[4] shapes) on one GPU.
    python tools/bench_train.py [--batch 4096] [--accum 2] [--steps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'safe-offline-rl-with-diffusion-model_amd'))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--accum', type=int, default=2)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--mults', default='1,2,4,8')
    ap.add_argument('--precision', default='f32')
    a = ap.parse_args()
    from trebi_amd import spec, synth
    from trebi_amd.training import Trainer
    from diffuser.models import TemporalUnet, GaussianDiffusion
    H, T, OBS, ACT, DIM = 32, 20, 17, 6, 32
    mults = tuple(int(m) for m in a.mults.split(','))
    D = OBS + ACT
    dev = torch.device('cuda:0')
    unet = TemporalUnet(H, D, 0, dim=DIM, dim_mults=mults)
    sd = synth.synth_state_dict(spec.unet_param_shapes(H, D, DIM, mults), 100)
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    diff = GaussianDiffusion(unet, H, OBS, ACT, n_timesteps=T, loss_type='l2', clip_denoised=False, predict_epsilon=False,
                             action_weight=10.0).to(dev)
    tr = Trainer(diff, train_batch_size=a.batch, train_lr=2e-4, gradient_accumulate_every=a.accum)
    tr.engine.set_precision(a.precision)
    g = torch.Generator(dev).manual_seed(0)
    data = []
    for _ in range(a.accum):
        x = torch.randn(a.batch, H, D, device=dev, generator=g)
        data.append((x, {0: x[:, 0, ACT:].contiguous()}))
    for _ in range(2):
        tr.train_step(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = tr.train_step(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fl = (tr.engine.flops() + tr.engine.flops(backward=True)) * a.batch * a.accum
    print(f'{a.precision} batch {a.batch} x accum {a.accum}: {1e3 * dt:.2f} ms per optimiser step, {a.batch * a.accum / dt:.0f} samples/s, '
          f'{fl / dt / 1e12:.1f} TFLOP/s algorithmic ({100 * fl / dt / 157.3e12:.1f} % of fp32 MFMA peak); '
          f'fwd {tr.engine.flops() / 1e6:.2f} + bwd {tr.engine.flops(backward=True) / 1e6:.2f} MFLOP/sample; '
          f'loss {float(losses[-1][0]):.4f}')


if __name__ == '__main__':
    main()
