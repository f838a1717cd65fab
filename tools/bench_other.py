"""Diagnostics: bench.other_configs alone (configs[2] at 512 and configs[3] at 8192 candidates on one GPU); environment
knobs (TREBI_CHAIN_MAXB, TREBI_PM_MINB, TREBI_OCC_MINB, ...) are read by the library.
usage: bench_other.py [2|3 = that config alone, fp32 only (profiling runs)] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
only = int(sys.argv[1]) - 2 if len(sys.argv) > 1 else None
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for e in bench.other_configs(dev, steps=steps, only=only, bf16=only is None):
    b = e.get('bf16x3', {})
    print(f"{e['workload'][:11]} f32 {e['ms_per_step']:.1f} ms frac {e['frac']:.3f} | bf16x3 {b.get('ms_per_step', 0):.1f} ms "
          f"{b.get('frac_of_f32_matrix_peak', 0):.3f} of f32 peak", flush=True)
