// Hand-written gfx950 (CDNA4) kernels of the TREBI guided-sampling hot path.
//
// conv_gemm_kernel  -- every Conv1d / ConvTranspose1d / input-gradient convolution of
//                      TemporalUnet and ValueFunction as an implicit GEMM on the f32 matrix
//                      cores (v_mfma_f32_16x16x4_f32: exact float32, = an fmaf chain), with the
//                      activations of a few whole candidates staged in LDS, weights streamed
//                      from L2 straight into registers, an in-workgroup split of the reduction
//                      across waves, and a fused epilogue: bias, GroupNorm(8)+Mish, time-embedding
//                      bias, 1x1-residual / identity-residual add; optionally a fused prologue:
//                      the GroupNorm+Mish backward of the producing block.
// head_kernel       -- ValueFunction's final MLP forward and the seed of its backward.
// guide/ddpm update -- the elementwise parts of the sample functions.
// feasibility / argsort / group argmax -- the selection at the end of p_sample_loop.
//
// Layout: activations are channels-last [B, L, ld]; the boundary tensors [B, horizon,
// transition] already are, so the reference's two einops transposes (temporal.py:125,150)
// disappear.  Wavefronts are 64 lanes; all block sizes are multiples of 64.
#include "device_util.h"

namespace trebi {

// ----------------------------------------------------------------------------------
// Value head: cat([x.view(B,-1), temb]) -> Linear -> Mish -> Linear   (temporal.py:237-238)
// One wave per candidate.  The temb columns of the first Linear are a per-timestep bias.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void head_kernel(HeadLaunch L) {
  const HeadP& p = L.p[blockIdx.y];
  const int b = blockIdx.x, lane = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* v = sm;             // [fc]
  float* dh = sm + p.fc;     // [hd]
  for (int i = lane; i < p.fc; i += 64) v[i] = p.v[(size_t)b * p.fc + i];
  __syncthreads();
  long long tt = p.t[b];
  tt = tt < 0 ? 0 : (tt >= p.T ? p.T - 1 : tt);
  float ysum = 0.f;
  for (int j = lane; j < p.hd; j += 64) {
    float h = p.tb[(size_t)tt * p.hd + j];
    const float* wr = p.w1 + (size_t)j * p.fc;
    for (int i = 0; i < p.fc; ++i) h = fmaf(wr[i], v[i], h);
    ysum += p.w2[j] * mish_f(h);
    dh[j] = p.w2[j] * mish_grad_f(h);
  }
  ysum = wave_sum(ysum);
  if (lane == 0) p.y[b] = ysum + p.b2;
  if (p.dv) {
    __syncthreads();
    for (int i = lane; i < p.fc; i += 64) {
      float a = 0.f;
      for (int j = 0; j < p.hd; ++j) a = fmaf(p.w1[(size_t)j * p.fc + i], dh[j], a);
      p.dv[(size_t)b * p.fc + i] = a;
    }
  }
}

hipError_t launch_head(const HeadLaunch& L, int nz, hipStream_t st) {
  const HeadP& p = L.p[0];
  size_t lds = (size_t)(p.fc + p.hd) * sizeof(float);
  hipLaunchKernelGGL(head_kernel, dim3(p.B, nz), dim3(64), lds, st, L);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Time-embedding tables (load time).  SinusoidalPosEmb -> Linear -> Mish -> Linear
// (helpers.py:21-28, temporal.py:70-75), then per block Mish -> Linear (temporal.py:28-31).
// ----------------------------------------------------------------------------------
__global__ void time_embed_kernel(const float* w1, const float* b1, const float* w2, const float* b2,
                                  int dim, int T, float* temb) {
  extern __shared__ float sm[];
  float* e = sm;             // [dim]
  float* h = sm + dim;       // [4*dim]
  const int t = blockIdx.x;
  const int half = dim / 2;
  const float sc = logf(10000.f) / (float)(half - 1);
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    float f = expf((float)i * -sc);
    float a = (float)t * f;
    e[i] = sinf(a);
    e[half + i] = cosf(a);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 4 * dim; j += blockDim.x) {
    float a = b1[j];
    for (int i = 0; i < dim; ++i) a = fmaf(w1[j * dim + i], e[i], a);
    h[j] = mish_f(a);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < dim; j += blockDim.x) {
    float a = b2[j];
    for (int i = 0; i < 4 * dim; ++i) a = fmaf(w2[j * 4 * dim + i], h[i], a);
    temb[(size_t)t * dim + j] = a;
  }
}

hipError_t launch_time_embed(const float* w1, const float* b1, const float* w2, const float* b2,
                             int dim, int T, float* temb, hipStream_t st) {
  hipLaunchKernelGGL(time_embed_kernel, dim3(T), dim3(128), (size_t)5 * dim * sizeof(float), st, w1, b1,
                     w2, b2, dim, T, temb);
  return hipGetLastError();
}

__global__ void time_linear_kernel(const float* w, int ldw, int col0, const float* b, const float* temb,
                                   int dim, int T, int N, int apply_mish, float* out) {
  const int t = blockIdx.x;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    float a = b ? b[n] : 0.f;
    for (int i = 0; i < dim; ++i) {
      float e = temb[(size_t)t * dim + i];
      if (apply_mish) e = mish_f(e);
      a = fmaf(w[(size_t)n * ldw + col0 + i], e, a);
    }
    out[(size_t)t * N + n] = a;
  }
}

hipError_t launch_time_linear(const float* w, int ldw, int col0, const float* b, const float* temb,
                              int dim, int T, int N, int apply_mish, float* out, hipStream_t st) {
  hipLaunchKernelGGL(time_linear_kernel, dim3(T), dim3(128), 0, st, w, ldw, col0, b, temb, dim, T, N,
                     apply_mish, out);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Elementwise steps of the sample functions
// ----------------------------------------------------------------------------------

// functions.py:95-119 (constrained), :21-29 (reward-only)
__global__ void guide_update_kernel(StepP s, float* x, const float* rg, const float* cg, const float* cy,
                                    const float* thr, const long long* t, const float* plv, int B) {
  const size_t per = (size_t)s.H * s.D;
  const size_t total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(i / per);
    int r = (int)(i - (size_t)b * per);
    int h = r / s.D, d = r - h * s.D;
    long long tt = t[b];
    int ti = tt < 0 ? 0 : (tt >= s.T ? s.T - 1 : (int)tt);
    float gr = rg[i];
    if (cg) {
      float c = cy[b], th = thr[b];
      float over = c > th ? 1.f : 0.f;
      if (s.guide_type == 0) {
        float cw = (c - th) / th;
        cw = cw < 0.f ? 0.f : (cw > 1.f ? 1.f : cw);      // NaN passes through, like torch.clamp
        cw = s.cgw * expf(cw);
        gr = gr - (cg[i] * cw) * over;
      } else {
        gr = gr - cg[i] * (s.cgw * over);
      }
    }
    if (s.scale_by_std) gr = expf(plv[ti]) * gr;
    if (s.state_mask && d >= s.act) gr = 0.f;
    if (tt < s.t_stopgrad) gr = 0.f;
    float v = x[i] + s.scale * gr;
    float cvv;
    if (conditioned(s, h, d, b, &cvv)) v = cvv;
    x[i] = v;
  }
}

hipError_t launch_guide_update(const StepP& s, float* x, const float* rg, const float* cg, const float* cy,
                               const float* thr, const long long* t, const float* plv, int B,
                               hipStream_t st) {
  size_t total = (size_t)B * s.H * s.D;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(guide_update_kernel, dim3(blocks), dim3(256), 0, st, s, x, rg, cg, cy, thr, t, plv, B);
  return hipGetLastError();
}

// diffusion.py:166-173, functions.py:131-137, diffusion.py:201
__global__ void ddpm_update_kernel(StepP s, float* x, const float* x0, const float* noise,
                                   const long long* t, const float* plv, const float* c1, const float* c2,
                                   int B, float* chain_out) {
  const size_t per = (size_t)s.H * s.D;
  const size_t total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(i / per);
    int r = (int)(i - (size_t)b * per);
    int h = r / s.D, d = r - h * s.D;
    long long tt = t[b];
    int ti = tt < 0 ? 0 : (tt >= s.T ? s.T - 1 : (int)tt);
    float mean = c1[ti] * x0[i] + c2[ti] * x[i];
    float nz = tt == 0 ? 0.f : noise[i];
    float v = mean + expf(0.5f * plv[ti]) * nz;
    float cvv;
    if (conditioned(s, h, d, b, &cvv)) v = cvv;
    x[i] = v;
    if (chain_out) chain_out[i] = v;
  }
}

hipError_t launch_ddpm_update(const StepP& s, float* x, const float* x0, const float* noise,
                              const long long* t, const float* plv, const float* c1, const float* c2, int B,
                              float* chain_out, hipStream_t st) {
  size_t total = (size_t)B * s.H * s.D;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ddpm_update_kernel, dim3(blocks), dim3(256), 0, st, s, x, x0, noise, t, plv, c1, c2, B,
                     chain_out);
  return hipGetLastError();
}

// diffusion.py:192-193: x = randn(shape); apply_conditioning
__global__ void init_x_kernel(StepP s, float* x, const float* noise0, int B, float* chain_out) {
  const size_t per = (size_t)s.H * s.D;
  const size_t total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(i / per);
    int r = (int)(i - (size_t)b * per);
    int h = r / s.D, d = r - h * s.D;
    float v = noise0[i];
    float cvv;
    if (conditioned(s, h, d, b, &cvv)) v = cvv;
    x[i] = v;
    if (chain_out) chain_out[i] = v;
  }
}

hipError_t launch_init_x(const StepP& s, float* x, const float* noise0, int B, float* chain_out,
                         hipStream_t st) {
  size_t total = (size_t)B * s.H * s.D;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(init_x_kernel, dim3(blocks), dim3(256), 0, st, s, x, noise0, B, chain_out);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// GuidedPolicy glue (policies.py:25-45, 70-82) on the device
// ----------------------------------------------------------------------------------
// _format_conditions + the threshold tensor: cond[b, :] = normalize(obs[b % n_obs, :]) (einops 'd w -> (repeat d) w':
// the repeat index is the OUTER one), thr_out[b] = thr[b % n_thr] (torch.tensor(thr).repeat(batch_size)).
// GaussianNormalizer.normalize (normalization.py:145-146): (x - mean) / std; mean == null: obs is already normalised.
__global__ void policy_prepare_kernel(const float* obs, int n_obs, int obs_dim, const float* mean, const float* stdv, int B,
                                      float* cond, const float* thr, int n_thr, float* thr_out) {
  const int total = B * obs_dim;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / obs_dim, d = i - b * obs_dim;
    float v = obs[(b % n_obs) * obs_dim + d];
    if (mean) v = (v - mean[d]) / stdv[d];
    cond[i] = v;
  }
  if (thr_out)
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) thr_out[b] = thr[b % n_thr];
}
hipError_t launch_policy_prepare(const float* obs, int n_obs, int obs_dim, const float* mean, const float* stdv, int B,
                                 float* cond, const float* thr, int n_thr, float* thr_out, hipStream_t st) {
  int blocks = (B * obs_dim + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(policy_prepare_kernel, dim3(blocks), dim3(256), 0, st, obs, n_obs, obs_dim, mean, stdv, B, cond, thr,
                     n_thr, thr_out);
  return hipGetLastError();
}
// action = unnormalize(trajectories[index[i], 0, :action_dim]) (policies.py:41-47; normalization.py:148-149: x*std + mean)
__global__ void policy_action_kernel(const float* x, int H, int D, int act, const long long* index, int n, int B,
                                     const float* mean, const float* stdv, float* out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * act; i += gridDim.x * blockDim.x) {
    const int k = i / act, d = i - k * act;
    long long b = index[k];
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);
    float v = x[(size_t)b * H * D + d];
    if (mean) v = v * stdv[d] + mean[d];
    out[i] = v;
  }
}
hipError_t launch_policy_action(const float* x, int H, int D, int act, const long long* index, int n, int B,
                                const float* mean, const float* stdv, float* out, hipStream_t st) {
  hipLaunchKernelGGL(policy_action_kernel, dim3((n * act + 63) / 64), dim3(64), 0, st, x, H, D, act, index, n, B, mean, stdv,
                     out);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// bf16x3 copy of a packed weight block: [(K/16)][4][Npad][4] fp32 -> per 32-channel step k two 16-byte-per-(lane group,
// column) planes, unit 2k = the eight hi halves, unit 2k+1 the eight lo halves, of channels 32k + 8g .. +7 (the B operand of
// v_mfma_f32_16x16x32_bf16: lane l holds k = 8(l>>4)+j of column l&15); same addressing as the fp32 layout.
// ----------------------------------------------------------------------------------
__global__ void pack_bf16x3_kernel(const float* w32, unsigned short* w16, int steps16, int Npad) {
  const long total = (long)(steps16 / 2) * 4 * Npad;          // one thread per (32-step, lane group, column)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Npad);
    const long r = i / Npad;
    const int g = (int)(r & 3);
    const long k32 = r >> 2;
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int within = 8 * g + e;                               // channel inside the 32-step
      const long k16 = 2 * k32 + (within >> 4);
      const int g16 = (within & 15) >> 2, j = within & 3;
      unsigned h, l;
      split_bf16(w32[((k16 * 4 + g16) * Npad + n) * 4 + j], &h, &l);
      hi[e] = (unsigned short)h;
      lo[e] = (unsigned short)l;
    }
    unsigned short* ph = w16 + (((2 * k32) * 4 + g) * Npad + n) * 8;
    unsigned short* pl = w16 + (((2 * k32 + 1) * 4 + g) * Npad + n) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ph[e] = hi[e]; pl[e] = lo[e]; }
  }
}
hipError_t launch_pack_bf16x3(const float* w32, float* w16, int steps16, int Npad, hipStream_t st) {
  const long total = (long)(steps16 / 2) * 4 * Npad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(blocks), dim3(256), 0, st, w32, (unsigned short*)w16, steps16, Npad);
  return hipGetLastError();
}

__global__ void fill_t_kernel(long long* t, int B, int value) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) t[i] = value;
}
__global__ void fill_f_kernel(float* p, int n, float value) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = value;
}
hipError_t launch_fill_f(float* p, int n, float value, hipStream_t st) {
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_f_kernel, dim3(blocks), dim3(256), 0, st, p, n, value);
  return hipGetLastError();
}
hipError_t launch_fill_t(long long* t, int B, int value, hipStream_t st) {
  int blocks = (B + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_t_kernel, dim3(blocks), dim3(256), 0, st, t, B, value);
  return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Selection (functions.py:126-129, diffusion.py:44-62)
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void feasibility_kernel(const float* ry, const float* cy,
                                                           const float* thr, int n, float* out) {
  __shared__ int any_feasible;
  if (threadIdx.x == 0) any_feasible = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mine |= (cy[i] <= thr[i]) ? 1 : 0;
  if (mine) any_feasible = 1;      // benign race: every writer stores 1
  __syncthreads();
  const int feas = any_feasible;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float c = cy[i];
    out[i] = feas ? (c > thr[i] ? ry[i] - 1e8f : ry[i]) : -c;
  }
}
hipError_t launch_feasibility(const float* ry, const float* cy, const float* thr, int n, float* out,
                              hipStream_t st) {
  hipLaunchKernelGGL(feasibility_kernel, dim3(1), dim3(1024), 0, st, ry, cy, thr, n, out);
  return hipGetLastError();
}

// Descending rank by counting: rank(i) = #{j : v[j] sorts before v[i]}, ties by index.  NaN sorts before every number,
// as in torch.argsort(descending=True) (the reference's sort_by_values, diffusion.py:44-49: NaN counts as the largest
// value), NaNs among themselves by index.  Every value becomes ONE 64-bit key -- the order-preserving integer image of
// the float (-0 folded onto +0, every NaN onto the top) above the inverted index -- so "j sorts before i" is a single
// unsigned compare.  O(n^2) compares: a workgroup ranks 256 values, its four quarters each sweep a quarter of the tiles
// (the all-gathered 65536 scores of configs[3]: 8.2 ms with the float compares of round 2, see DESIGN).
__device__ __forceinline__ unsigned long long sort_key(float v, int idx) {
  v += 0.f;                                           // -0 -> +0: equal to +0 like the float compare
  unsigned u = __float_as_uint(v);
  u = (v != v) ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
  return ((unsigned long long)u << 32) | (unsigned)(~idx);
}
__global__ __launch_bounds__(1024) void argsort_desc_kernel(const float* v, int n, long long* order) {
  __shared__ __attribute__((aligned(16))) unsigned long long tile[4][256];
  __shared__ int part[4][256];
  const int e = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int i = blockIdx.x * 256 + e;
  const unsigned long long ki = i < n ? sort_key(v[i], i) : 0ull;
  int rank = 0;
  const int ntiles = (n + 255) / 256;
  for (int t0 = 0; t0 < ntiles; t0 += 4) {
    const int j = (t0 + q) * 256 + e;
    tile[q][e] = j < n ? sort_key(v[j], j) : 0ull;    // key 0 sorts after everything (never counted)
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 256; k += 2) {
      const ulonglong2 kk = *(const ulonglong2*)&tile[q][k];
      rank += (kk.x > ki) + (kk.y > ki);
    }
    __syncthreads();
  }
  part[q][e] = rank;
  __syncthreads();
  if (q == 0 && i < n) order[part[0][e] + part[1][e] + part[2][e] + part[3][e]] = i;
}
hipError_t launch_argsort_desc(const float* v, int n, long long* order, hipStream_t st) {
  hipLaunchKernelGGL(argsort_desc_kernel, dim3((n + 255) / 256), dim3(1024), 0, st, v, n, order);
  return hipGetLastError();
}

// torch.argmax per group (max_by_values_in_batch, diffusion.py:51-62): the first maximum; NaN counts as the largest value
// (the first NaN of a group wins), as torch.argmax does.
__global__ __launch_bounds__(64) void group_argmax_kernel(const float* v, int per, long long* best) {
  const int gidx = blockIdx.x, lane = threadIdx.x;
  float bv = 0.f;
  int bi = 0x7fffffff;
  // a sorts before b: NaN before numbers, larger before smaller, lower index on ties
  auto better = [](float xv, int xi, float yv, int yi) {
    if (yi == 0x7fffffff) return xi != 0x7fffffff;
    if (xi == 0x7fffffff) return false;
    const bool xn = xv != xv, yn = yv != yv;
    if (xn || yn) return xn && (!yn || xi < yi);
    return xv > yv || (xv == yv && xi < yi);
  };
  for (int i = lane; i < per; i += 64) {
    const float x = v[(size_t)gidx * per + i];
    if (better(x, i, bv, bi)) { bv = x; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) best[gidx] = (long long)gidx * per + bi;
}
hipError_t launch_group_argmax(const float* v, int ng, int per, long long* best, hipStream_t st) {
  hipLaunchKernelGGL(group_argmax_kernel, dim3(ng), dim3(64), 0, st, v, per, best);
  return hipGetLastError();
}

}  // namespace trebi
