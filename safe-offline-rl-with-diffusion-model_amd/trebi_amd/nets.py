"""TemporalUnet / ValueFunction with the reference's constructor and forward signatures
(/root/reference/diffuser/models/temporal.py:49-151, 153-239) and state_dict keys, executing
on the HIP library.  The modules only *hold* parameters (so reference checkpoints load with
strict=True); all arithmetic happens in libtrebi_hip.so.

Inference only: outputs carry no autograd graph.  The one gradient the hot path needs --
d(sum V)/dx for guidance -- is `ValueFunction.value_and_input_grad`, an explicit
backward-to-input chain in the library.
"""
import ctypes as C
import math
import weakref

import torch
import torch.nn as nn

from . import _lib, spec

DEFAULT_MAX_TIMESTEPS = 1000      # diffusion.py:73 default n_timesteps


def _register(root, name, tensor):
    """Create nested containers so that state_dict() yields exactly `name`."""
    parts = name.split('.')
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def _default_init(name, shape, shapes):
    """PyTorch's default initialisers for the layer kinds involved (kaiming_uniform(a=sqrt(5)) ==
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Conv/Linear weight and bias; GroupNorm ones/zeros)."""
    if '.block.2.' in name:
        return torch.ones(shape) if name.endswith('weight') else torch.zeros(shape)
    wshape = shapes[name[:-4] + 'weight'] if name.endswith('bias') else shape
    fan_in = 1
    for s in wshape[1:]:
        fan_in *= s
    if 'ups.' in name and '.3.conv.' in name:
        fan_in = wshape[0] * wshape[2]
    bound = 1.0 / math.sqrt(fan_in)
    return torch.empty(shape).uniform_(-bound, bound)


class _HipNet(nn.Module):
    _kind = None

    def _setup(self, shapes, horizon, transition_dim, dim, dim_mults):
        self.horizon = horizon
        self.transition_dim = transition_dim
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.max_timesteps = DEFAULT_MAX_TIMESTEPS
        self._shapes = shapes
        for name, shape in shapes.items():
            _register(self, name, _default_init(name, shape, shapes))
        self._handle = None
        self._packed_key = None
        self._ws = {}
        self.precision = ('f32', 0)               # operand precision of the convolutions (set_precision)
        self.generation = 0                       # bumped whenever the packed weights are rebuilt
        self._dependents = weakref.WeakSet()      # GaussianDiffusion objects holding captured loops on the handle

    # ---- packed weights live in the library; rebuilt when parameters change -----------
    def _desc(self):
        d = _lib.NetDesc()
        d.kind = self._kind
        d.horizon, d.transition_dim, d.dim = self.horizon, self.transition_dim, self.dim
        d.n_mults = len(self.dim_mults)
        for i, m in enumerate(self.dim_mults):
            d.dim_mults[i] = m
        d.max_timesteps = int(self.max_timesteps)
        return d

    def _param_key(self, device):
        return (str(device), int(self.max_timesteps),
                tuple((p.data_ptr(), p._version) for p in self.parameters()))

    def set_max_timesteps(self, n):
        """Size of the per-timestep bias tables (GaussianDiffusion sets it to n_timesteps)."""
        if int(n) != self.max_timesteps:
            self.max_timesteps = int(n)

    def net_handle(self, device):
        if device.type != 'cuda':
            raise RuntimeError('trebi_amd networks run on an MI355X only (tensor is on %s); '
                               'there is no CPU path' % device)
        key = self._param_key(device)
        if self._handle is None or key != self._packed_key:
            self._free()
            sd = self.state_dict()
            names = list(sd.keys())
            host = [sd[k].detach().to('cpu', torch.float32).contiguous() for k in names]
            n = len(names)
            c_names = (C.c_char_p * n)(*[k.encode() for k in names])
            c_ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in host])
            c_numel = (C.c_int64 * n)(*[t.numel() for t in host])
            h = C.c_void_p()
            d = self._desc()
            with torch.cuda.device(device):
                _lib.check(_lib.lib().trebi_net_create(_lib.ctx(device.index or 0), C.byref(d), c_names, c_ptrs,
                                                       c_numel, n, C.byref(h)))
            self._handle, self._packed_key = h, key
            self._ws = {}
            self.generation += 1
            self._apply_precision(device)
        return self._handle

    def set_precision(self, mode='f32', min_batch=0):
        """'f32' (default: exact fp32 products on the fp32 matrix pipe) or 'bf16x3' (opt-in: fp32 operands split into
        two bf16 halves, three bf16 MFMAs per product, fp32 accumulation; ~2^-16 relative error per product, stated
        bound 1e-3 on network outputs; engaged from `min_batch` candidates up, 512 if 0).  The reference is fp32
        throughout; this has no counterpart there."""
        if mode not in ('f32', 'bf16x3'):
            raise ValueError("precision must be 'f32' or 'bf16x3'")
        self.precision = (mode, int(min_batch))
        if self._handle is not None:
            for d in list(self._dependents):          # launch plans bake the weight pointers in
                d.drop_samplers_of(self)
            self._apply_precision(next(self.parameters()).device)
            self.generation += 1
        return self

    def _apply_precision(self, device):
        mode, mb = self.precision
        if mode == 'bf16x3' or self._handle is not None:
            with torch.cuda.device(device):
                _lib.check(_lib.lib().trebi_net_set_precision(self._handle, 1 if mode == 'bf16x3' else 0, mb))

    def add_dependent(self, diffusion):
        self._dependents.add(diffusion)

    def _free(self):
        if getattr(self, '_handle', None) is not None:
            for d in list(getattr(self, '_dependents', ())):      # their plans / graphs point into the packed weights
                d.drop_samplers_of(self)
            _lib.lib().trebi_net_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _check_x(self, x):
        """[batch, horizon, >= transition_dim] (wider rows are read as their first transition_dim columns, which is
        how ValueGuide's x[:, :, :transition_dim] slice arrives without a copy); anything else is refused."""
        if x.dim() != 3 or x.shape[1] != self.horizon or x.shape[2] < self.transition_dim:
            raise RuntimeError(f'expected x of shape [batch, {self.horizon}, {self.transition_dim}], got {tuple(x.shape)}')
        return x.shape

    _WS_KEEP = 4                                  # workspaces kept per net (most recently used batch sizes)

    def _workspace(self, B, device, query):
        key = (B, str(device))
        ws = self._ws.pop(key, None)
        if ws is None:
            nbytes = query(self.net_handle(device), B)
            while len(self._ws) >= self._WS_KEEP:     # a sweep over batch sizes must not pin one buffer per size
                self._ws.pop(next(iter(self._ws)))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._ws[key] = ws                            # (re-)inserted last: dicts keep insertion order
        return ws

    def layers(self, B, backward=False):
        """Per-layer parity taps (trebi_net_layer_info): for every launch of the forward (or of the ValueFunction's
        backward-to-input) on a batch of B, its descriptor and a function view(ref) -> the [B, L, C] tensor it names in the
        workspace of the LAST call at that batch size (None for tensors that are the call's own arguments)."""
        dev = next(self.parameters()).device
        h = self.net_handle(dev)
        ws = self._ws[(B, str(dev))].view(torch.float32)

        def view(ref):
            if ref.where != 1:
                return None
            o = ref.offset_floats
            return ws[o:o + (B * ref.L - 1) * ref.ld + ref.C].as_strided((B, ref.L, ref.C), (ref.L * ref.ld, ref.ld, 1))
        out = []
        for i in range(_lib.lib().trebi_net_layer_count(h, 1 if backward else 0)):
            info = _lib.LayerInfo()
            _lib.check(_lib.lib().trebi_net_layer_info(h, 1 if backward else 0, i, B, C.byref(info)))
            out.append(info)
        return out, view

    def flops(self, backward=False):
        dev = next(self.parameters()).device
        return _lib.lib().trebi_net_flops(self.net_handle(dev), 1 if backward else 0)

    @staticmethod
    def _rows(x):
        """Accept [B,H,ld] views whose last dim is a leading column slice of a contiguous tensor."""
        B, H, D = x.shape
        if x.dtype != torch.float32:
            x = x.float()
        if x.stride(2) == 1 and x.stride(0) == H * x.stride(1) and x.stride(1) >= D:
            return x, x.stride(1)
        x = x.contiguous()
        return x, D

    def _time(self, time, B, device):
        """[B] int64 timesteps, range-checked on the host: the kernels index per-timestep tables built for
        t in [0, max_timesteps) (the reference evaluates its embedding for any t; here that is refused loudly).
        A Python int or a CPU tensor is checked before it is uploaded (no device read-back); a CUDA tensor costs one."""
        if not (torch.is_tensor(time) and time.is_cuda):
            t = torch.as_tensor(time)
            if t.numel():
                lo, hi = int(t.min()), int(t.max())
            t = t.to(device=device, dtype=torch.int64)
        else:
            t = time.to(torch.int64)
            if t.numel():
                lo, hi = (int(v) for v in torch.aminmax(t))          # one read-back; the fused loop never comes here
        if t.dim() == 0:
            t = t.expand(B)
        if t.numel() and (lo < 0 or hi >= self.max_timesteps):
            raise RuntimeError(f'timesteps must lie in [0, {self.max_timesteps}) (set_max_timesteps sizes the '
                               f'tables); got [{lo}, {hi}]')
        return t.contiguous()


class TemporalUnet(_HipNet):
    _kind = 0

    def __init__(self, horizon, transition_dim, cond_dim, dim=32, dim_mults=(1, 2, 4, 8), attention=False,
                 output_dim=None):
        super().__init__()
        if attention:
            raise NotImplementedError('LinearAttention is never instantiated on the planning path '
                                      '(SURVEY.md fact 3) and is not built')
        if output_dim not in (None, transition_dim):
            raise NotImplementedError('output_dim != transition_dim is only used by the decouple models')
        self._setup(spec.unet_param_shapes(horizon, transition_dim, dim, dim_mults), horizon, transition_dim, dim,
                    dim_mults)

    @torch.no_grad()
    def forward(self, x, cond, time):
        """x: [batch, horizon, transition] -> same shape (temporal.py:120-151)."""
        B, H, D = self._check_x(x)
        if B == 0:                                   # empty batch: what torch's Conv1d stack returns
            return torch.empty(0, H, self.transition_dim, device=x.device, dtype=torch.float32)
        x, ldx = self._rows(x)
        t = self._time(time, B, x.device)
        h = self.net_handle(x.device)
        ws = self._workspace(B, x.device, _lib.lib().trebi_unet_workspace_bytes)
        out = torch.empty(B, H, self.transition_dim, device=x.device, dtype=torch.float32)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().trebi_unet_forward(h, _lib.ptr(x), ldx, _lib.ptr(t), _lib.ptr(out), self.transition_dim,
                                                 B, _lib.ptr(ws), ws.numel(), C.c_void_p(st)))
        return out


    def tap(self, block, B):
        """Output of the ResidualTemporalBlock `block` ('downs.0.0', 'mid_block2', ...) left in the workspace by the last
        forward on a batch of B: a [B, L, C] view (channels-last; the reference's hooks see [B, C, L])."""
        dev = next(self.parameters()).device
        ws = self._ws[(B, str(dev))].view(torch.float32)
        off, L, Cc, ld = C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().trebi_unet_tap(self.net_handle(dev), block.encode(), B, C.byref(off), C.byref(L), C.byref(Cc),
                                             C.byref(ld)))
        return ws[off.value:off.value + B * L.value * ld.value].view(B, L.value, ld.value)[:, :, :Cc.value]


class ValueFunction(_HipNet):
    _kind = 1

    def __init__(self, horizon, transition_dim, cond_dim, dim=32, dim_mults=(1, 2, 4, 8), out_dim=1):
        super().__init__()
        if out_dim != 1:
            raise NotImplementedError('out_dim must be 1 (the guide squeezes it, guides.py:14)')
        self._setup(spec.value_param_shapes(horizon, transition_dim, dim, dim_mults, out_dim), horizon,
                    transition_dim, dim, dim_mults)

    def _run(self, x, time, want_grad):
        B, H, D = self._check_x(x)
        if B == 0:
            return (torch.empty(0, device=x.device, dtype=torch.float32),
                    torch.empty(0, H, D, device=x.device, dtype=torch.float32) if want_grad else None)
        x, ldx = self._rows(x)
        t = self._time(time, B, x.device)
        h = self.net_handle(x.device)
        ws = self._workspace(B, x.device, _lib.lib().trebi_value_workspace_bytes)
        y = torch.empty(B, device=x.device, dtype=torch.float32)
        grad = torch.empty(B, H, D, device=x.device, dtype=torch.float32) if want_grad else None
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().trebi_value_gradients(h, _lib.ptr(x), ldx, _lib.ptr(t), _lib.ptr(y), _lib.ptr(grad), D,
                                                    B, _lib.ptr(ws), ws.numel(), C.c_void_p(st)))
        return y, grad

    @torch.no_grad()
    def forward(self, x, cond, time, *args):
        """x: [batch, horizon, transition] -> [batch, 1] (temporal.py:213-239)."""
        y, _ = self._run(x, time, False)
        return y[:, None]

    @torch.no_grad()
    def value_and_input_grad(self, x, time):
        """(V(x)[:, 0], d(sum V)/dx) -- what ValueGuide.gradients needs (guides.py:16-21)."""
        return self._run(x, time, True)
