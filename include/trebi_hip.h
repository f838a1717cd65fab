/*
 * trebi_hip.h -- C ABI of the MI355X (gfx950) implementation of TREBI's
 * cost-guided diffusion-planning hot path.
 *
 * The reference (qianlin04/Safe-offline-RL-with-diffusion-model) is pure
 * Python/PyTorch and has no FFI of its own; each entry point below names the
 * reference interface (file:line under /root/reference) it replaces.  Plain C
 * types only: device pointers are `float*` / `int64_t*` on the context's GPU,
 * streams are passed as `void*` (a hipStream_t).  Every call returns 0 on
 * success or a negative trebi_status; trebi_last_error() gives the message of the
 * last failure on the calling thread.  Only the set-up calls (*_create, *_bind,
 * *_destroy) allocate or synchronise the device; every other call only enqueues on
 * the stream it is given (a profiled trebi_sampler_run, a diagnostic, also creates
 * timing events).  Workspaces are caller-owned (size from the *_workspace_bytes queries).
 *
 * Tensor conventions (all float32, contiguous unless a row stride is given):
 *   x, grad, trajectories : [B, horizon, ldx]   (ldx >= transition_dim; the nets read the
 *                                                first transition_dim columns, guides.py:13)
 *   t                     : [B] int64 diffusion timesteps in [0, max_timesteps).  The library cannot
 *                           look at device memory without synchronising, so the kernels clamp
 *                           an out-of-range t to the nearest table row; the host mirror
 *                           (trebi_amd/nets.py) range-checks and refuses such t loudly.
 *   cond                  : [B, observation_dim] per conditioned horizon index
 */
#ifndef TREBI_HIP_H
#define TREBI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TREBI_OK = 0,
  TREBI_BAD_ARG = -1,            /* null pointer, negative size, unknown enum value         */
  TREBI_BAD_SHAPE = -2,          /* tensor table does not match the architecture descriptor */
  TREBI_UNSUPPORTED_CONFIG = -3, /* valid in the reference but outside what the kernels tile */
  TREBI_WORKSPACE_TOO_SMALL = -4,
  TREBI_HIP_ERROR = -5
} trebi_status;

typedef struct trebi_ctx trebi_ctx;         /* one per device                          */
typedef struct trebi_net trebi_net;         /* packed TemporalUnet or ValueFunction    */
typedef struct trebi_sampler trebi_sampler; /* a captured p_sample_loop for a fixed B  */

const char* trebi_last_error(void);
int trebi_version(void);

/* ---- context -------------------------------------------------------------------- */
int trebi_ctx_create(int device, trebi_ctx** out);
void trebi_ctx_destroy(trebi_ctx* ctx);

/* ---- networks --------------------------------------------------------------------
 * kind 0: models.TemporalUnet(horizon, transition_dim, cond_dim, dim, dim_mults, attention=False)
 *         /root/reference/diffuser/models/temporal.py:49-118
 * kind 1: models.ValueFunction(horizon, transition_dim, cond_dim, dim, dim_mults, out_dim=1)
 *         /root/reference/diffuser/models/temporal.py:153-211
 * Weights are handed over as the reference's state_dict(): `names[i]` is the key
 * (e.g. "downs.0.0.blocks.0.block.0.weight"), `tensors[i]` a HOST pointer to
 * `numels[i]` contiguous float32 in the reference's own layout.  The library packs
 * them into its kernel layout on the device and precomputes, for every integer
 * timestep t < max_timesteps, the time-embedding biases the blocks add
 * (temporal.py:28-32,44,70-75; t is batch-uniform on this path, diffusion.py:199).
 */
typedef struct {
  int kind;           /* 0 = TemporalUnet, 1 = ValueFunction */
  int horizon;
  int transition_dim;
  int dim;
  int n_mults;
  int dim_mults[8];
  int max_timesteps;  /* table size for t; GaussianDiffusion.n_timesteps or larger */
} trebi_net_desc;

int trebi_net_create(trebi_ctx* ctx, const trebi_net_desc* desc, const char* const* names,
                     const float* const* tensors, const int64_t* numels, int n_tensors,
                     trebi_net** out);
void trebi_net_destroy(trebi_net* net);

/* Operand precision of the convolutions (opt-in; default 0).  No counterpart in the reference (fp32 throughout).
 *   0  f32     v_mfma_f32_16x16x4_f32, exact fp32 products
 *   1  bf16x3  every fp32 operand split into two bf16 halves (x = hi + lo), a product computed as
 *              hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: ~2^-16 relative error per
 *              product, 5.3x the fp32 matrix rate.  Engaged for launches of at least min_batch candidates
 *              (<= 0: keep the current threshold, initially 512) and for the layers that can take it (whole
 *              32-channel steps, no fused GroupNorm-backward prologue); everything else stays on the fp32 pipe.
 *              GroupNorm statistics, Mish and all accumulation stay fp32.  Stated bound: network outputs and guide
 *              gradients within 1e-3 of the reference (tests/test_gpu_bf16.py).
 * A set-up call (allocates the bf16x3 weight copies on first use, synchronises); samplers created on the net
 * before the call keep their launch plans: create them afterwards. */
int trebi_net_set_precision(trebi_net* net, int mode, int min_batch);

/* Number of reference state_dict entries the descriptor implies, and the i-th name/numel
 * (lets the caller check a checkpoint before handing it over; serialization.py:36-60). */
int trebi_net_param_count(const trebi_net_desc* desc);
int trebi_net_param_info(const trebi_net_desc* desc, int i, char* name_out, size_t name_cap,
                         int64_t* numel_out);

/* Algorithmic FLOPs per candidate of one forward (backward=0) or one backward-to-input
 * (backward=1) pass: 2 flop per multiply-add of the convolutions and linears, structurally
 * zero padding taps excluded.  No counterpart in the reference. */
double trebi_net_flops(const trebi_net* net, int backward);

/* TemporalUnet.forward(x, cond, time) -> [B, horizon, transition_dim]
 * /root/reference/diffuser/models/temporal.py:120-151 (cond is unused there). */
size_t trebi_unet_workspace_bytes(const trebi_net* net, int B);
int trebi_unet_forward(trebi_net* net, const float* x, int ldx, const int64_t* t, float* out,
                       int ldo, int B, void* workspace, size_t workspace_bytes, void* stream);

/* Parity tap (no counterpart in the reference; its tests would use forward hooks): after trebi_unet_forward on a
 * batch of B, the output of the ResidualTemporalBlock `block` ("downs.0.0", "mid_block2", "ups.1.1", ...;
 * temporal.py:18-46) is the channels-last tensor [B, L, ld] (first C columns) at workspace + offset_floats. */
int trebi_unet_tap(const trebi_net* net, const char* block, int B, size_t* offset_floats, int* L, int* C, int* ld);

/* Per-layer parity taps (no counterpart in the reference; its tests would put forward / backward hooks on the modules of
 * helpers.py:30-63 and temporal.py:18-46).  A pass is the list of launches trebi_unet_forward / trebi_value_gradients make:
 * pass 0 the forward -- one entry per Conv1dBlock (Conv1d + GroupNorm + Mish, with the block's time bias or residual folded
 * in), Downsample1d, Upsample1d, final 1x1 conv, value head --, pass 1 the ValueFunction's backward-to-input (input-gradient
 * convolutions with the GroupNorm + Mish backward as their prologue).  For a batch of B, entry `index` reads and writes the
 * channels-last tensors described by its trebi_tensor_ref's: where = 0 absent, 1 workspace + offset_floats, 2 the call's x,
 * 3 the call's output, 4 the call's gradient output; [B, L, ld] with the first C columns used (ld = 0: the caller's). */
typedef struct { int where, L, C, ld; size_t offset_floats; } trebi_tensor_ref;
typedef struct {
  char name[120];          /* state_dict prefix of the layer's weight, " (dgrad)" appended for an input-gradient conv */
  int kind;                /* 0 convolution stage, 1 value head */
  int Lin, Lout, Cin, Cin2, N, stride, pad;
  int phase_split;         /* ConvTranspose1d as two output phases: columns per phase, else 0 */
  int group_norm, time_bias, prologue;
  trebi_tensor_ref src, src2 /* 1x1 residual conv's input */, res /* identity residual */, dst;
  trebi_tensor_ref saved_xhat, saved_gain /* mish'(z) * gamma */, saved_stats /* (mean, rstd) per group */;
  trebi_tensor_ref pro_xhat, pro_gain, pro_stats;   /* what the backward prologue reads (a forward layer's saved_*) */
} trebi_layer_info;
int trebi_net_layer_count(const trebi_net* net, int pass);
int trebi_net_layer_info(const trebi_net* net, int pass, int index, int B, trebi_layer_info* out);

/* trebi_unet_forward with NO workspace argument and no allocation of any kind: the
 * activations go to a static per-GPU device array of TREBI_STATIC_SCRATCH_BYTES that the
 * library's code object carries (every GPU the library runs on holds it until the process
 * ends).  No counterpart in the reference; it exists for callers that cannot allocate or
 * size a workspace -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Returns TREBI_WORKSPACE_TOO_SMALL when
 * trebi_unet_workspace_bytes(net, B) > TREBI_STATIC_SCRATCH_BYTES (about 260
 * candidates for the halfcheetah H=32 (1,2,4,8) TemporalUnet).  Results are bit-identical to
 * trebi_unet_forward.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race; serialise them. */
#define TREBI_STATIC_SCRATCH_BYTES (32u << 20)
int trebi_unet_forward_static(trebi_net* net, const float* x, int ldx, const int64_t* t, float* out,
                              int ldo, int B, void* stream);

/* ValueFunction.forward(x, cond, time)[:, 0] -> y[B]
 * /root/reference/diffuser/models/temporal.py:213-239, guides.py:12-14.
 * With grad != NULL also ValueGuide.gradients (guides.py:16-21): grad[B,horizon,ldg] =
 * d(sum y)/dx (columns >= transition_dim are written as 0), by an explicit
 * backward-to-input chain -- no autograd tape, no weight gradients. */
size_t trebi_value_workspace_bytes(const trebi_net* net, int B);
int trebi_value_gradients(trebi_net* net, const float* x, int ldx, const int64_t* t, float* y,
                          float* grad /* nullable */, int ldg, int B, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- elementwise steps of the sample functions ------------------------------------
 * One guide-ascent iteration, functions.py:95-119 (constrained) / :21-29 (reward only):
 *   guide_type 0: g = r_grad - cost_grad * cgw*exp(clamp((c-thr)/thr,0,1)) * [c > thr]
 *   guide_type 1: g = r_grad - cost_grad * cgw*[c > thr]
 *   cost_grad == NULL: g = r_grad
 *   g *= exp(plv[t]) if scale_grad_by_std;  g[:, :, action_dim:] = 0 if state_grad_mask;
 *   g = 0 where t < t_stopgrad;  x += scale*g;  apply_conditioning (helpers.py:142-145). */
typedef struct {
  int horizon, transition_dim, action_dim, observation_dim;
  int n_timesteps;
  float scale;
  int t_stopgrad;
  int scale_grad_by_std;
  int state_grad_mask;
  float cost_grad_weight;
  int guide_type;
  int n_cond;          /* number of conditioned horizon indices (keys of the cond dict) */
  int cond_index[4];
} trebi_step_desc;

int trebi_guide_update(trebi_ctx* ctx, const trebi_step_desc* d, float* x, const float* r_grad,
                       const float* cost_grad /* nullable */, const float* cost_y,
                       const float* cost_threshold, const int64_t* t,
                       const float* posterior_log_variance /* dev [n_timesteps] */,
                       const float* const* cond /* n_cond dev ptrs */, int B, void* stream);

/* DDPM reverse step, diffusion.py:166-173 + functions.py:131-137 (predict_epsilon=False):
 *   x = coef1[t]*x0 + coef2[t]*x + exp(0.5*plv[t]) * noise * [t != 0];  apply_conditioning. */
int trebi_ddpm_update(trebi_ctx* ctx, const trebi_step_desc* d, float* x, const float* x0,
                      const float* noise, const int64_t* t, const float* posterior_log_variance,
                      const float* coef1, const float* coef2, const float* const* cond, int B,
                      void* stream);

/* Feasibility rule, functions.py:126-129 (global over the n candidates):
 *   values = -cost_y if no cost_y <= thr else where(cost_y > thr, r_y - 1e8, r_y). */
int trebi_feasibility_rule(trebi_ctx* ctx, const float* r_y, const float* cost_y,
                           const float* cost_threshold, int n, float* values_out, void* stream);

/* sort_by_values, diffusion.py:44-49: order_out[k] = index of the k-th largest value
 * (ties broken by lower index; torch.argsort leaves ties unspecified). */
int trebi_argsort_desc(trebi_ctx* ctx, const float* values, int n, int64_t* order_out,
                       void* stream);

/* max_by_values_in_batch, diffusion.py:51-62: best_out[g] = g*per + argmax(values[g*per:(g+1)*per]). */
int trebi_group_argmax(trebi_ctx* ctx, const float* values, int n_groups, int per,
                       int64_t* best_out, void* stream);

/* ---- GuidedPolicy glue on the device ----------------------------------------------
 * GuidedPolicy._format_conditions and the cost-threshold tensor
 * (/root/reference/diffuser/sampling/policies.py:28-36,70-82):
 *   cond_out[b, :] = (obs[b % n_obs, :] - mean) / std     (GaussianNormalizer.normalize,
 *                    diffuser/datasets/normalization.py:145-146; mean == std == NULL: obs is already normalised;
 *                    einops 'd w -> (repeat d) w': the repeat index is the outer one)
 *   thr_out[b]     = thr[b % n_thr]                        (torch.tensor(thr).repeat(batch_size); thr_out NULL: skipped)
 * obs: dev [n_obs, obs_dim] raw observations; B = batch_size * n_obs rows are written. */
int trebi_policy_prepare(trebi_ctx* ctx, const float* obs, int n_obs, int obs_dim, const float* mean,
                         const float* stdv, int B, float* cond_out, const float* thr, int n_thr, float* thr_out,
                         void* stream);
/* policies.py:38-47: action_out[i, :] = trajectories[index[i], 0, :action_dim] * std + mean
 * (GaussianNormalizer.unnormalize, normalization.py:148-149; NULL statistics: left normalised) -- only what the
 * caller reads leaves the device; x: dev [B, horizon, transition_dim], index: dev [n] (e.g. the head of
 * trebi_argsort_desc's order, or trebi_group_argmax's result). */
int trebi_policy_first_action(trebi_ctx* ctx, const float* x, int B, int horizon, int transition_dim, int action_dim,
                              const int64_t* index, int n, const float* mean, const float* stdv, float* action_out,
                              void* stream);

/* ---- the whole reverse-diffusion loop ---------------------------------------------
 * GaussianDiffusion.p_sample_loop (diffusion.py:187-206) driving default_sample_fn
 * (diffusion.py:19-29), n_step_guided_p_sample (functions.py:9-37) or
 * n_step_guided_and_constrained_p_sample (functions.py:75-137), captured once as a HIP
 * graph for a fixed candidate count B and replayed with no host synchronisation.  The
 * final feasibility rule / sort (diffusion.py:208-211) is NOT part of it: candidates may
 * be sharded over GPUs, so the caller all-gathers (values, costs) and then applies
 * trebi_feasibility_rule + trebi_argsort_desc to the global arrays. */
typedef struct {
  trebi_step_desc step;
  int kind;            /* 0 default_sample_fn, 1 n_step_guided_p_sample, 2 ..._and_constrained_ */
  int n_guide_steps;
  int use_graph;       /* 0: plain stream launches, 1: capture + replay a hipGraph */
  int skip_dead_evals; /* 1: do not evaluate the guides where their result cannot reach any output
                          (t < t_stopgrad and t != 0); outputs are bit-identical either way      */
} trebi_sampler_desc;

int trebi_sampler_create(trebi_ctx* ctx, trebi_net* unet, trebi_net* reward /* nullable */,
                         trebi_net* cost /* nullable */, const trebi_sampler_desc* desc,
                         const float* posterior_log_variance, const float* coef1,
                         const float* coef2 /* HOST [n_timesteps] each */, int B,
                         trebi_sampler** out);
void trebi_sampler_destroy(trebi_sampler* s);

/* noise: dev [n_timesteps+1, B, horizon, transition_dim] (noise[0] is x_T, noise[1+k] the k-th step's
 * draw, diffusion.py:192 / functions.py:134); x_out: dev [B,horizon,transition_dim]; values/costs:
 * dev [B], the guides' raw outputs at the last step (zeros where the sample function has none);
 * chain: nullable dev [n_timesteps+1, B, horizon, transition_dim].
 *
 * trebi_sampler_bind is the set-up call for a set of buffers (like *_create it allocates and
 * synchronises): it resolves every network pass to launches for these pointers and, with
 * use_graph, captures and instantiates the loop.  trebi_sampler_run then only enqueues (one
 * hipGraphLaunch, or the plain launches of the loop for use_graph = 0, which a caller may capture
 * into its own graph); it refuses pointers other than the bound ones.  The CONTENTS of the
 * buffers may change freely between runs. */
int trebi_sampler_bind(trebi_sampler* s, const float* noise, const float* const* cond,
                       const float* cost_threshold, float* x_out, float* values, float* costs, float* chain);
int trebi_sampler_run(trebi_sampler* s, const float* noise, const float* const* cond,
                      const float* cost_threshold /* dev [B], nullable unless kind 2 */,
                      float* x_out, float* values, float* costs, float* chain, void* stream);

/* Timing hook for bench.py: average duration in microseconds of the implicit-GEMM
 * kernel launches recorded (with HIP events on the launch stream) during the last
 * trebi_sampler_run with profiling enabled, and their count / algorithmic FLOPs. */
int trebi_sampler_set_profiling(trebi_sampler* s, int enabled);
int trebi_sampler_get_profile(trebi_sampler* s, double* gemm_us_total, int64_t* gemm_launches,
                              double* gemm_flops_total, double* loop_us_total);

/* ---- the denoiser's training step (BASELINE configs[4]) ----------
 * GaussianDiffusion.loss / p_losses / q_sample (diffuser/models/diffusion.py:229-261) with
 * WeightedL2 / WeightedL1 (diffuser/models/helpers.py:152-167,193-201) through a TemporalUnet
 * (temporal.py:49-151), the full backward that loss.backward() runs there (every weight, bias,
 * GroupNorm and time-MLP gradient), and Trainer.train's optimiser step: torch.optim.Adam +
 * the EMA copy (diffuser/utils/training.py:19-35,94-98,104-122).
 *
 * Parameters, gradients, Adam moments and the EMA copy are FLAT device buffers of
 * trebi_trainer_param_count() floats, the TemporalUnet's state_dict tensors back to back in
 * trebi_net_param_info order (kind 0 descriptor) and in the reference's own layouts; they are
 * caller-owned (the data-parallel all-reduce is one collective on the flat gradient).  The
 * trainer owns its workspace (sized for max_batch at create).  create and bind are set-up
 * calls (they allocate and synchronise); accumulate / step / repack only enqueue. */
typedef struct trebi_trainer trebi_trainer;
typedef struct {
  int horizon, observation_dim, action_dim, n_timesteps;
  int dim, n_mults, dim_mults[8];
  int predict_epsilon;  /* target of the loss: 0 x_start (config/locomotion.py:30), 1 the noise */
  int loss_l1;          /* 0 'l2' (WeightedL2), 1 'l1' (WeightedL1) */
  int n_cond;           /* conditioned horizon indices (apply_conditioning, helpers.py:142-145) */
  int cond_index[4];
} trebi_train_desc;

int trebi_trainer_create(trebi_ctx* ctx, const trebi_train_desc* desc, int max_batch, trebi_trainer** out);
void trebi_trainer_destroy(trebi_trainer* tr);
int64_t trebi_trainer_param_count(const trebi_trainer* tr);
/* Algorithmic FLOPs per sample: forward (backward=0) or dgrad + wgrad (backward=1). */
double trebi_trainer_flops(const trebi_trainer* tr, int backward);

/* Attach the flat device buffers (exp_avg / exp_avg_sq / ema may be NULL if trebi_trainer_step is
 * not used / no EMA is kept) and the HOST tables: loss_fn.weights [horizon*transition_dim]
 * (diffusion.py:122-149), sqrt_alphas_cumprod and sqrt_one_minus_alphas_cumprod [n_timesteps]
 * (diffusion.py:96-97).  Packs the weights from `params`. */
int trebi_trainer_bind(trebi_trainer* tr, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                       float* ema, const float* loss_weights, const float* sqrt_alphas_cumprod,
                       const float* sqrt_one_minus_alphas_cumprod);
/* trebi_net_set_precision for the trainer's convolutions (forward and input-gradient; the weight-gradient GEMM
 * stays fp32). */
int trebi_trainer_set_precision(trebi_trainer* tr, int mode, int min_batch);
/* After the caller changed `params` itself (load_state_dict): re-pack the kernels' copies. */
int trebi_trainer_repack(trebi_trainer* tr, void* stream);

/* One micro-batch of Trainer.train (training.py:110-116): x_noisy = q_sample(x_start, t, noise) +
 * conditioning, x_recon = unet(x_noisy, t) + conditioning, loss = mean(l * weights), then
 *   grads += grad_scale * d loss / d params        (training.py:115: grad_scale = 1 / gradient_accumulate_every;
 *                                                    data-parallel: times B_local / B_global before the sum all-reduce)
 *   loss_out[0] += loss,  loss_out[1] += a0_loss   (dev [2]; the caller zeroes it)
 * x_start / noise: dev [B, horizon, transition_dim]; t: dev [B] int64; cond[k]: dev [B, observation_dim]. */
int trebi_trainer_accumulate(trebi_trainer* tr, const float* x_start, const float* const* cond, const int64_t* t,
                             const float* noise, int B, float grad_scale, float* loss_out, void* stream);

/* Gradient buckets for a data-parallel caller (no counterpart in training.py, which is single-GPU; the reduction
 * itself is the caller's: torch.distributed / RCCL).  The backward finishes the gradients of the state_dict's tail --
 * ups.*, mid_block1/2.*, final_conv.* = grads[trebi_trainer_bucket_split(), param_count) -- before it starts on the
 * down path.  While an event is set (a hipEvent_t the caller owns; NULL clears it), every trebi_trainer_accumulate
 * records it on its stream at that point, so a side stream can wait for it and all-reduce the tail slice while the
 * down path's backward still runs; grads[0, split) are final when accumulate's last kernel is. */
int64_t trebi_trainer_bucket_split(const trebi_trainer* tr);
int trebi_trainer_set_bucket_event(trebi_trainer* tr, void* event);

/* optimizer.step(); optimizer.zero_grad(); step_ema() (training.py:118-122, 94-98), then re-pack:
 * torch.optim.Adam with bias correction for the adam_step-th update (counting from 1);
 * ema_mode 0: EMA untouched, 1: ema = params (step < step_start_ema), 2: ema = ema*decay + (1-decay)*params. */
int trebi_trainer_step(trebi_trainer* tr, int64_t adam_step, float lr, float beta1, float beta2, float eps,
                       int ema_mode, float ema_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TREBI_HIP_H */
