"""Deterministic synthetic weights / inputs shared by the golden-vector generator,
the tests and bench.py.

There are no pretrained TREBI checkpoints offline, so all
parity work uses seeded random weights.  They are drawn with numpy's PCG64 (stable
across platforms) from the *shape table* in spec.py, so the exact same tensors are
rebuilt in this container (where the reference is importable) and on the GPU box
(where it is not) without committing 34 MB of weights.

Distributions follow PyTorch's defaults for the layers involved (U(-1/sqrt(fan_in),
+1/sqrt(fan_in)) for Conv1d / Linear weight and bias) except that GroupNorm
gamma/beta are perturbed away from (1, 0) so that the affine part is exercised.
"""
import zlib
import numpy as np


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def synth_state_dict(shapes, seed, gn_jitter=0.2):
    """shapes: OrderedDict name -> shape (spec.unet_param_shapes / value_param_shapes)."""
    out = {}
    for name, shape in shapes.items():
        r = _rng(seed, name)
        is_gn = ('.block.2.' in name)
        if is_gn and name.endswith('weight'):
            w = 1.0 + gn_jitter * r.uniform(-1, 1, shape)
        elif is_gn and name.endswith('bias'):
            w = gn_jitter * r.uniform(-1, 1, shape)
        else:
            if name.endswith('bias'):
                wshape = shapes[name[:-4] + 'weight']
            else:
                wshape = shape
            fan_in = int(np.prod(wshape[1:]))
            if 'ups.' in name and '.3.conv.' in name:       # ConvTranspose1d: fan_in uses dim 0 x K
                fan_in = int(wshape[0] * wshape[2])
            bound = 1.0 / np.sqrt(fan_in)
            w = r.uniform(-bound, bound, shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synth_normal(seed, name, shape):
    return np.ascontiguousarray(_rng(seed, name).standard_normal(shape), dtype=np.float32)
