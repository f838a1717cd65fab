"""The denoiser's training step on the HIP library (BASELINE configs[4]).

Reference: Trainer.train's inner step, EMA, save/load (/root/reference/diffuser/utils/training.py:19-35,37-166) and
GaussianDiffusion.loss / p_losses / q_sample (diffuser/models/diffusion.py:229-261) with WeightedL2
(diffuser/models/helpers.py:152-167,198-201).

How it maps.  The reference builds an autograd graph in `model.loss(*batch)` and walks it in `loss.backward()`; here
one library call (trebi_trainer_accumulate) does q_sample + conditioning, the TemporalUnet forward, the weighted loss
AND the whole backward, adding `grad_scale * dloss/dparams` into a flat gradient buffer.  `optimizer.step()`,
`zero_grad()` and `step_ema()` are one more call (trebi_trainer_step: torch.optim.Adam arithmetic + the EMA rule),
after which the library re-packs its kernel-layout weights from the flat parameters.

The TemporalUnet's parameters become VIEWS of one flat CUDA tensor (state_dict order, reference layouts), so
`state_dict()` / checkpoints keep working, and the data-parallel path is a single all-reduce (RCCL over xGMI) of the
3.96 M-float flat gradient per optimiser step, after all accumulation micro-batches.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .nets import TemporalUnet


def _flatten_into(module, names, flat):
    """Copy the parameters into `flat` back to back and re-point them at views of it."""
    off = 0
    for name in names:
        p = module.get_parameter(name)
        n = p.numel()
        view = flat[off:off + n].view(p.shape)
        view.copy_(p.data.to(flat.device))
        p.data = view
        off += n
    assert off == flat.numel()


class TrainEngine:
    """Flat parameter / gradient / Adam / EMA buffers + the library's trainer object for one GaussianDiffusion."""

    def __init__(self, diffusion, max_batch, ema_model=None, cond_keys=(0,)):
        unet = diffusion.model
        if not isinstance(unet, TemporalUnet):
            raise RuntimeError('the HIP training step covers GaussianDiffusion(TemporalUnet) (SURVEY.md 8f rank 1); '
                               f'got {type(unet).__name__}')
        kind = getattr(diffusion.loss_fn, 'kind', None)
        if kind not in ('l1', 'l2'):
            raise RuntimeError(f"loss_type {kind!r}: only the weighted 'l1' / 'l2' losses of the denoiser are built")
        dev = diffusion.betas.device
        if dev.type != 'cuda':
            raise RuntimeError('trebi_amd training runs on an MI355X only (move the model to cuda); there is no CPU path')
        self.device, self.diffusion, self.unet = dev, diffusion, unet
        self.names = list(unet._shapes.keys())
        P = sum(int(np.prod(s)) for s in unet._shapes.values())
        self.max_batch = int(max_batch)
        d = _lib.TrainDesc()
        d.horizon, d.observation_dim, d.action_dim = diffusion.horizon, diffusion.observation_dim, diffusion.action_dim
        if diffusion.transition_dim != diffusion.observation_dim + diffusion.action_dim:
            raise RuntimeError('predict_reward_done models are not built')
        d.n_timesteps, d.dim, d.n_mults = diffusion.n_timesteps, unet.dim, len(unet.dim_mults)
        for i, m in enumerate(unet.dim_mults):
            d.dim_mults[i] = m
        d.predict_epsilon = int(bool(diffusion.predict_epsilon))
        d.loss_l1 = int(kind == 'l1')
        self.cond_keys = sorted(int(k) for k in cond_keys)
        d.n_cond = len(self.cond_keys)
        for i, k in enumerate(self.cond_keys):
            d.cond_index[i] = k
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().trebi_trainer_create(_lib.ctx(dev.index or 0), C.byref(d), self.max_batch, C.byref(h)))
        self.h = h
        assert _lib.lib().trebi_trainer_param_count(h) == P
        self.params = torch.empty(P, device=dev)
        self.grads = torch.zeros(P, device=dev)
        self.exp_avg = torch.zeros(P, device=dev)
        self.exp_avg_sq = torch.zeros(P, device=dev)
        _flatten_into(unet, self.names, self.params)
        self.ema = None
        if ema_model is not None:
            self.ema = torch.empty(P, device=dev)
            _flatten_into(ema_model.model, self.names, self.ema)
            self.ema_unet = ema_model.model
        self.loss_acc = torch.zeros(2, device=dev)
        tabs = [t.detach().to('cpu', torch.float32).contiguous() for t in
                (diffusion.loss_fn.weights, diffusion.sqrt_alphas_cumprod, diffusion.sqrt_one_minus_alphas_cumprod)]
        fp = lambda t: C.cast(t.data_ptr(), _lib.c_float_p)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().trebi_trainer_bind(h, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.exp_avg),
                                                     _lib.ptr(self.exp_avg_sq), _lib.ptr(self.ema), fp(tabs[0]), fp(tabs[1]),
                                                     fp(tabs[2])))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def bucket_split(self):
        """grads[bucket_split:] (ups.*, mid_block*, final_conv.*) are final before the down path's backward runs."""
        return int(_lib.lib().trebi_trainer_bucket_split(self.h))

    def set_bucket_event(self, event):
        """`event`: a torch.cuda.Event that has been recorded once (so that its handle exists), or None.  While set,
        every accumulate() records it on its stream at the point where grads[bucket_split:] are final."""
        handle = C.c_void_p(event.cuda_event) if event is not None else C.c_void_p()
        _lib.check(_lib.lib().trebi_trainer_set_bucket_event(self.h, handle))

    def flops(self, backward=False):
        return _lib.lib().trebi_trainer_flops(self.h, 1 if backward else 0)

    def set_precision(self, mode='f32', min_batch=0):
        """Operand precision of the forward and input-gradient convolutions ('f32' | 'bf16x3', see nets.set_precision);
        the weight-gradient GEMM, the loss, Adam and the EMA stay fp32."""
        if mode not in ('f32', 'bf16x3'):
            raise ValueError("precision must be 'f32' or 'bf16x3'")
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().trebi_trainer_set_precision(self.h, 1 if mode == 'bf16x3' else 0, int(min_batch)))

    def repack(self):
        """After `params` was changed from outside (load_state_dict on the module)."""
        _lib.check(_lib.lib().trebi_trainer_repack(self.h, self._stream()))

    def accumulate(self, x, cond, t, noise, grad_scale=1.0):
        """One micro-batch: grads += grad_scale * d loss/d params.  Returns a device tensor [loss, a0_loss]."""
        B = x.shape[0]
        if B == 0:
            return torch.zeros(2, device=self.device)
        if sorted(int(k) for k in cond.keys()) != self.cond_keys:
            raise RuntimeError(f'conditioned timesteps {sorted(cond.keys())} differ from the engine\'s {self.cond_keys}')
        f32 = lambda v: torch.as_tensor(v, device=self.device, dtype=torch.float32).contiguous()
        x, noise = f32(x), f32(noise)
        t = torch.as_tensor(t, device=self.device).to(torch.int64).contiguous()
        cts = [f32(cond[k]) for k in self.cond_keys]
        for c in cts:
            if c.shape != (B, self.diffusion.observation_dim):
                raise RuntimeError(f'cond has shape {tuple(c.shape)}, expected {(B, self.diffusion.observation_dim)}')
        if x.shape != (B, self.diffusion.horizon, self.diffusion.transition_dim) or noise.shape != x.shape or t.shape != (B,):
            raise RuntimeError('x / noise must be [batch, horizon, transition_dim] and t [batch]')
        out = torch.zeros(2, device=self.device)
        _lib.check(_lib.lib().trebi_trainer_accumulate(self.h, _lib.ptr(x), _lib.ptr_array(cts), _lib.ptr(t), _lib.ptr(noise),
                                                       B, float(grad_scale), _lib.ptr(out), self._stream()))
        return out

    def step(self, adam_step, lr, betas=(0.9, 0.999), eps=1e-8, ema_mode=0, ema_decay=0.995):
        _lib.check(_lib.lib().trebi_trainer_step(self.h, int(adam_step), float(lr), float(betas[0]), float(betas[1]),
                                                 float(eps), int(ema_mode), float(ema_decay), self._stream()))
        for net in (self.unet, getattr(self, 'ema_unet', None)):      # planning with these modules must re-pack
            if net is not None:
                net._packed_key = None

    def __del__(self):
        try:
            torch.cuda.synchronize(self.device)
            _lib.lib().trebi_trainer_destroy(self.h)
        except Exception:
            pass


def _clone_diffusion(model):
    """copy.deepcopy(model) of training.py:61 without copying library handles: rebuild from the constructor arguments."""
    from .diffusion import GaussianDiffusion
    u = model.model
    unet = TemporalUnet(u.horizon, u.transition_dim, 0, dim=u.dim, dim_mults=u.dim_mults)
    twin = GaussianDiffusion(unet, model.horizon, model.observation_dim, model.action_dim, n_timesteps=model.n_timesteps,
                             loss_type=model.loss_type, clip_denoised=model.clip_denoised,
                             predict_epsilon=model.predict_epsilon, action_weight=model.action_weight)
    twin.load_state_dict({k: v.detach().clone() for k, v in model.state_dict().items()})
    return twin.to(model.betas.device)


def cycle(dl):                                                     # training.py:14-17
    while True:
        for data in dl:
            yield data


class Trainer:
    """diffuser.utils.training.Trainer for the denoiser (training.py:37-166): Adam, gradient accumulation, EMA,
    save / load -- rendering, wandb and cloud sync are outside the hot path and not built.

    Data-parallel: pass `group` (or have torch.distributed initialised): every rank accumulates over its own
    micro-batches, then the flat gradient is all-reduced (sum) once per optimiser step, as two slices: the tail
    bucket (up path, mid blocks, final conv: the first gradients the backward finishes) goes out on a side stream
    while the last micro-batch's down-path backward still runs, the head bucket after it.  Each micro-batch is
    scaled by B_local / B_global so that the sum equals the gradient of the mean loss over the global batch.
    `engine_factory` exists so that the distributed logic can be exercised on CPU ranks (gloo) without a GPU."""

    def __init__(self, diffusion_model, dataset=None, renderer=None, ema_decay=0.995, train_batch_size=32, train_lr=2e-5,
                 gradient_accumulate_every=2, step_start_ema=2000, update_ema_every=10, log_freq=100, sample_freq=1000,
                 save_freq=1000, label_freq=100000, save_parallel=False, results_folder='./results', n_reference=8,
                 bucket=None, group=None, distributed=None, engine_factory=None, adam_betas=(0.9, 0.999), adam_eps=1e-8):
        self.model = diffusion_model
        self.ema_decay = ema_decay
        self.update_ema_every = update_ema_every
        self.step_start_ema = step_start_ema
        self.log_freq, self.sample_freq, self.save_freq, self.label_freq = log_freq, sample_freq, save_freq, label_freq
        self.batch_size = train_batch_size
        self.gradient_accumulate_every = gradient_accumulate_every
        self.lr, self.betas, self.eps = train_lr, adam_betas, adam_eps
        self.dataset = dataset
        self.logdir = results_folder
        self.group = group
        self.distributed = (dist.is_available() and dist.is_initialized()) if distributed is None else distributed
        if engine_factory is None:
            self.ema_model = _clone_diffusion(diffusion_model)                       # training.py:61
            self.engine = diffusion_model.train_engine(train_batch_size, ema_model=self.ema_model)
            if self.engine.ema is None:
                raise RuntimeError('the model already has a training engine without an EMA copy')
        else:
            self.ema_model = None
            self.engine = engine_factory(diffusion_model, train_batch_size)
        self.dataloader = None
        if dataset is not None:
            self.dataloader = cycle(torch.utils.data.DataLoader(dataset, batch_size=train_batch_size, num_workers=1,
                                                                shuffle=True, pin_memory=True))
        self.step = 0
        self.adam_steps = 0
        self._bucket_event = self._side = None

    def _world(self):
        return dist.get_world_size(self.group) if self.distributed else 1

    def train_step(self, batches, global_batch_sizes=None):
        """One pass of the loop body training.py:109-122.  `batches`: gradient_accumulate_every items of
        (x, cond) or (x, cond, t, noise); t / noise are drawn like diffusion.py:260,241 when absent.
        Returns the per-micro-batch [loss, a0_loss] tensors (device, no host sync)."""
        dev = self.engine.device if hasattr(self.engine, 'device') else None
        world = self._world()
        losses = []
        g = self.engine.grads
        split = int(getattr(self.engine, 'bucket_split', 0)) if world > 1 else 0
        split = split if 0 < split < g.numel() else 0
        through_host = world > 1 and g.is_cuda and dist.get_backend(self.group) == 'gloo'
        overlap = bool(split) and g.is_cuda and not through_host
        if overlap and self._bucket_event is None:
            self._bucket_event = torch.cuda.Event()
            self._bucket_event.record(torch.cuda.current_stream(dev))      # creates the handle the library records
            self._side = torch.cuda.Stream(dev)
        tail_work = None
        for i, batch in enumerate(batches):
            x, cond = batch[0], batch[1]
            B = len(x)
            if len(batch) >= 4:
                t, noise = batch[2], batch[3]
            else:
                t = torch.randint(0, self.model.n_timesteps, (B,), device=dev).long()
                noise = torch.randn(tuple(x.shape), device=dev)
            Bg = global_batch_sizes[i] if global_batch_sizes is not None else B * world
            scale = (B / Bg if Bg else 0.0) / len(batches)                              # training.py:115
            last = overlap and i == len(batches) - 1
            if last:
                if B:
                    self.engine.set_bucket_event(self._bucket_event)
                else:                                      # an empty shard enqueues nothing: the stream's tail is the point
                    self._bucket_event.record(torch.cuda.current_stream(dev))
            losses.append(self.engine.accumulate(x, cond, t, noise, grad_scale=scale))
            if last:
                self.engine.set_bucket_event(None)
                self._side.wait_event(self._bucket_event)
                with torch.cuda.stream(self._side):        # the tail bucket leaves while the down path's backward runs
                    tail_work = dist.all_reduce(g[split:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if world > 1:
            def reduce(v):
                if through_host:                           # gloo (CPU rehearsals): through the host
                    vh = v.cpu()
                    dist.all_reduce(vh, op=dist.ReduceOp.SUM, group=self.group)
                    v.copy_(vh)
                else:
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            if tail_work is not None:
                reduce(g[:split])
                tail_work.wait()                           # the current stream waits for the side stream's collective
            elif split:
                reduce(g[split:])
                reduce(g[:split])
            else:
                reduce(g)
        self.adam_steps += 1
        ema_mode = 0
        if self.step % self.update_ema_every == 0:                                      # training.py:121-122, 94-98
            ema_mode = 1 if self.step < self.step_start_ema else 2
        self.engine.step(self.adam_steps, self.lr, self.betas, self.eps, ema_mode, self.ema_decay)
        self.step += 1
        return losses

    def train(self, n_train_steps):                                                     # training.py:104-139
        if self.dataloader is None:
            raise RuntimeError('Trainer.train needs a dataset; use train_step(batches) with explicit batches')
        total = []
        for _ in range(n_train_steps):
            batches = []
            for _ in range(self.gradient_accumulate_every):
                batch = next(self.dataloader)
                x, cond = batch[0], batch[1]
                batches.append((x, {int(k): v for k, v in cond.items()}))
            step = self.step
            losses = self.train_step(batches)
            total.extend(losses)
            if step % self.save_freq == 0:                  # the reference saves BEFORE `self.step += 1` (training.py:124-138)
                self.save(step // self.label_freq * self.label_freq, step=step)
            if step % self.log_freq == 0:
                l = losses[-1].tolist()
                print(f'{step}: {l[0] / self.gradient_accumulate_every:8.4f} | a0_loss: {l[1]:8.4f}', flush=True)
        print('mean loss : ', float(torch.stack(total)[:, 0].mean()))

    # ---- checkpoints in the reference's format (training.py:141-166) ----
    def save(self, epoch, step=None):
        """`step`: the value stored under 'step' (default self.step).  Data-parallel: every rank holds the same state, rank 0
        writes the file."""
        path = os.path.join(self.logdir, f'state_{epoch}.pt')
        if self.distributed and dist.get_rank(self.group) != 0:
            return path
        os.makedirs(self.logdir, exist_ok=True)
        data = {'step': self.step if step is None else step, 'model': self.model.state_dict(), 'ema': self.ema_model.state_dict()}
        torch.save({k: ({n: v.detach().cpu().clone() for n, v in d.items()} if isinstance(d, dict) else d)
                    for k, d in data.items()}, path)
        return path

    def load(self, epoch):
        path = os.path.join(self.logdir, f'state_{epoch}.pt')
        data = torch.load(path, map_location='cpu', weights_only=True)
        self.step = data['step']
        self.model.load_state_dict(data['model'])          # copies INTO the flat views
        self.ema_model.load_state_dict(data['ema'])
        self.engine.repack()
