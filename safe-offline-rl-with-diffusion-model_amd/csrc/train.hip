// train.hip -- gfx950 kernels of the TemporalUnet TRAINING step (BASELINE configs[4]).
//
// Reference: GaussianDiffusion.q_sample / p_losses / loss (diffuser/models/diffusion.py:229-261), WeightedL2 / L1
// (diffuser/models/helpers.py:152-167,193-201), what torch.autograd computes for Conv1d / ConvTranspose1d /
// GroupNorm / Mish / Linear on the way back through TemporalUnet (temporal.py:18-46,120-151), torch.optim.Adam and
// the trainer's EMA rule (diffuser/utils/training.py:19-35,94-98,104-122).
//
// The forward and the input-gradient (dgrad) convolutions are conv_gemm_kernel (conv.hip) -- the same kernels the
// planning path runs.  What is new here:
//   wgrad_kernel      weight gradient as a GEMM with K = (candidate, position) on v_mfma_f32_16x16x4_f32:
//                     dW[cb][ca][tap] = sum A[b, l*stride+tap-pad, ca] * Bm[b, l, cb].  The four k-slots of one
//                     MFMA are the SAME position of four candidates, so a (position, tap) pair that falls on zero
//                     padding is skipped as a whole and no halo is staged.  K is split over workgroups; every
//                     workgroup stores its partial tile to a slab and wreduce_kernel sums the slabs into the
//                     gradient in the reference's [cb][ca][tap] layout (plain stores + a sum pass instead of float
//                     atomics: bitwise reproducible and, per MI355X_MICROARCH.md "Global float atomics", faster).
//   gnb_kernel        GroupNorm+Mish backward of a Conv1dBlock: dY, and per-workgroup partial sums of the gamma /
//                     beta / conv-bias gradients and of the time-bias gradient scattered by timestep.
//   the elementwise ends: q_sample + conditioning, weighted loss + its gradient, Adam + EMA, the weight re-pack
//   (gather through an index map), the small time-MLP backward.
#include "device_util.h"

#include <algorithm>

namespace trebi {

// =====================================================================================================
// q_sample + apply_conditioning   (diffusion.py:229-238, 243-244)
// =====================================================================================================
__global__ void qsample_kernel(StepP s, const float* x0, const float* noise, const long long* t, const float* sa,
                               const float* sb, float* out, int B) {
  const size_t per = (size_t)s.H * s.D, total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per), r = (int)(i - (size_t)b * per), h = r / s.D, d = r - h * s.D;
    long long tt = t[b];
    const int ti = tt < 0 ? 0 : (tt >= s.T ? s.T - 1 : (int)tt);
    float v = sa[ti] * x0[i] + sb[ti] * noise[i];
    float cv;
    if (conditioned(s, h, d, b, &cv)) v = cv;
    out[i] = v;
  }
}
hipError_t launch_qsample(const StepP& s, const float* x0, const float* noise, const long long* t, const float* sa,
                          const float* sb, float* out, int B, hipStream_t st) {
  const size_t total = (size_t)B * s.H * s.D;
  int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(qsample_kernel, dim3(blocks), dim3(256), 0, st, s, x0, noise, t, sa, sb, out, B);
  return hipGetLastError();
}

// =====================================================================================================
// weighted loss and its gradient   (diffusion.py:246-254, helpers.py:159-167)
//   x_recon = apply_conditioning(net output);  l = (x_recon - targ)^2 (or |.|);  loss = mean(l * w)
//   a0_loss = mean(l[:, 0, :act] / w[0, :act]);  d loss / d(net output) = 0 where the output was overwritten
// =====================================================================================================
__global__ __launch_bounds__(256) void loss_kernel(StepP s, const float* pred, const float* targ, const float* w, int l1,
                                                   float inv_count, float inv_a0, float gscale, float* dout,
                                                   float* partial, int B) {
  const size_t per = (size_t)s.H * s.D, total = (size_t)B * per;
  float ls = 0.f, as = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per), r = (int)(i - (size_t)b * per), h = r / s.D, d = r - h * s.D;
    float p = pred[i], cv;
    const bool over = conditioned(s, h, d, b, &cv);
    if (over) p = cv;
    const float df = p - targ[i], wt = w[r];
    const float l = l1 ? fabsf(df) : df * df;
    ls += l * wt;
    if (h == 0 && d < s.act) as += l / wt;
    const float g = l1 ? (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) : 2.f * df;
    dout[i] = over ? 0.f : g * wt * gscale;
  }
  __shared__ float red[2][4];
  ls = wave_sum(ls);
  as = wave_sum(as);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ls; red[1][threadIdx.x >> 6] = as; }
  __syncthreads();
  if (threadIdx.x == 0) {                      // per-workgroup partials, summed in a fixed order below (no float atomics)
    partial[2 * blockIdx.x] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv_count;
    partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv_a0;
  }
}
__global__ __launch_bounds__(64) void loss_finish_kernel(const float* partial, int n, float* loss_out) {
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) {
    a += partial[2 * i];
    b += partial[2 * i + 1];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (threadIdx.x == 0) {
    loss_out[0] += a;
    loss_out[1] += b;
  }
}
hipError_t launch_loss(const StepP& s, const float* pred, const float* targ, const float* w, int l1, float inv_count,
                       float inv_a0, float gscale, float* dout, float* partial /* [2 * LOSS_BLOCKS] */, float* loss_out, int B,
                       hipStream_t st) {
  const size_t total = (size_t)B * s.H * s.D;
  int blocks = (int)std::min<size_t>((total + 255) / 256, LOSS_BLOCKS);
  hipLaunchKernelGGL(loss_kernel, dim3(blocks), dim3(256), 0, st, s, pred, targ, w, l1, inv_count, inv_a0, gscale, dout,
                     partial, B);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, partial, blocks, loss_out);
  return hipGetLastError();
}

// =====================================================================================================
// GroupNorm + Mish backward of a Conv1dBlock (helpers.py:46-63)
//   z = gamma*xhat + beta;  dz = dA * mish'(z);  d gamma = sum dz*xhat;  d beta = sum dz
//   dxhat = dz*gamma;  dY = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat*xhat));  d bias = sum dY
//   time bias (temporal.py:44): d tb[b, c] = sum_l dA[b, l, c], scattered into row t[b] of a [T, C] table
// A team of 16 lanes owns one (candidate, group): both passes stay inside the team (DPP reductions).
// =====================================================================================================
template <int R>
__device__ __forceinline__ float ror_add(float v) {      // v + (v rotated right by R inside its row of 16 lanes)
  return v + dpp_f<0x120 + R>(v);
}
// sum over the lanes of a 16-lane row that share (lane % g4); g4 a power of two <= 16
__device__ __forceinline__ float stride_sum(float v, int g4) {
  if (g4 <= 8) v = ror_add<8>(v);
  if (g4 <= 4) v = ror_add<4>(v);
  if (g4 <= 2) v = ror_add<2>(v);
  if (g4 <= 1) v = ror_add<1>(v);
  return v;
}

__global__ __launch_bounds__(1024) void gnb_kernel(GnbP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = p.C, T = p.T;
  // No float atomics: sums are formed in a FIXED order, so the step is bit-reproducible.  A team always works on the same
  // group (its instances are 64 x gridDim.x apart, a multiple of 8), so the gamma / beta / bias column sums accumulate in
  // the team's registers over its sweeps and the eight teams of a group are combined once, in team order, through LDS; the
  // time-bias rows, which depend on each candidate's timestep, are staged per sweep and added to the [T, C] table by one
  // thread per column in team order (two barriers per sweep; a workgroup makes ~2 sweeps at a 4096 batch).
  const int gch0 = p.gch;
  float* ttab = sm;                         // [T][C]  (only with ptime)
  float* stage = sm + (p.ptime ? T * C : 0);  // [64 teams][3][gch] column-sum parts, or [64][gch] time rows per sweep
  int* tts = (int*)(stage + 64 * 3 * gch0);   // [64] timestep of each team's candidate
  if (p.ptime)
    for (int i = threadIdx.x; i < T * C; i += 1024) ttab[i] = 0.f;
  const int tl = threadIdx.x & 15, team = threadIdx.x >> 4;      // 64 teams of 16 lanes
  const int gch = p.gch, g4 = gch >> 2, L = p.L, cnt4 = L * g4;
  const float icnt = 1.f / (float)(cnt4 * 4);
  const int ninst = p.B * GN_GROUPS;
  const int c4 = tl & (g4 - 1);             // the float4 column of this lane inside the group (16 % g4 == 0)
  const int per_sweep = gridDim.x * 64;
  const int first = blockIdx.x * 64;
  const int nsweeps = first < ninst ? (ninst - first + per_sweep - 1) / per_sweep : 0;     // the same for every team
  f32x4 acc_ga = {0.f, 0.f, 0.f, 0.f}, acc_be = acc_ga, acc_bi = acc_ga;
  __syncthreads();
  for (int it = 0; it < nsweeps; ++it) {
    const int inst = first + team + it * per_sweep;
    const bool valid = inst < ninst;
    const int b = valid ? inst >> 3 : 0, gi = team & 7;
    const int c = gi * gch + 4 * c4;
    f32x4 dga = {0.f, 0.f, 0.f, 0.f}, dbe = dga, dtb = dga, dbi = dga;
    if (valid) {
    const float rstd = gld(p.stats + 2 * inst + 1);
    const f32x4 ga = gld4(p.gamma + c), be = gld4(p.beta + c);
    const size_t row0 = (size_t)b * L;
    float s1 = 0.f, s2 = 0.f;
    if (cnt4 <= 32) {
      // at most two float4 per lane (every H = 32 shape): the gradient and xhat stay in registers for the second pass
      // instead of being read from HBM again (5 -> 3 streams of B x L x C floats; the kernel is HBM-bound)
      f32x4 dxr[2], xhr[2];
      size_t off[2];
      bool ev[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tl + 16 * u;
        ev[u] = e < cnt4;
        const int li = (ev[u] ? e : tl) / g4;
        off[u] = row0 + li;
        dxr[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xhr[u] = dxr[u];
        if (ev[u]) {
          const f32x4 da = gld4(p.dA + off[u] * (size_t)p.ldA + c);
          xhr[u] = gld4(p.xhat + off[u] * (size_t)C + c);
          f32x4 dz;
#pragma unroll
          for (int q = 0; q < 4; ++q) dz[q] = da[q] * mish_grad_f(xhr[u][q] * ga[q] + be[q]);
          dxr[u] = dz * ga;
          s1 += hsum(dxr[u]);
          s2 += hsum(dxr[u] * xhr[u]);
          dga += dz * xhr[u];
          dbe += dz;
          dtb += da;
        }
      }
      s1 += dpp_f<0xB1>(s1); s1 += dpp_f<0x4E>(s1); s1 += dpp_f<0x141>(s1); s1 += dpp_f<0x140>(s1);
      s2 += dpp_f<0xB1>(s2); s2 += dpp_f<0x4E>(s2); s2 += dpp_f<0x141>(s2); s2 += dpp_f<0x140>(s2);
      const float m1 = s1 * icnt, m2 = s2 * icnt;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (ev[u]) {
          const f32x4 dy = (dxr[u] - m1 - xhr[u] * m2) * rstd;
          dbi += dy;
          gst4(p.dY + off[u] * (size_t)C + c, dy);
        }
    } else {
    for (int e = tl; e < cnt4; e += 16) {
      const int li = e / g4;
      const f32x4 da = gld4(p.dA + (row0 + li) * (size_t)p.ldA + c);
      const f32x4 xh = gld4(p.xhat + (row0 + li) * (size_t)C + c);
      f32x4 dz;
#pragma unroll
      for (int q = 0; q < 4; ++q) dz[q] = da[q] * mish_grad_f(xh[q] * ga[q] + be[q]);
      const f32x4 dx = dz * ga;
      s1 += hsum(dx);
      s2 += hsum(dx * xh);
      dga += dz * xh;
      dbe += dz;
      dtb += da;
    }
    // team sums (16 lanes = one DPP row)
    s1 += dpp_f<0xB1>(s1); s1 += dpp_f<0x4E>(s1); s1 += dpp_f<0x141>(s1); s1 += dpp_f<0x140>(s1);
    s2 += dpp_f<0xB1>(s2); s2 += dpp_f<0x4E>(s2); s2 += dpp_f<0x141>(s2); s2 += dpp_f<0x140>(s2);
    const float m1 = s1 * icnt, m2 = s2 * icnt;
    for (int e = tl; e < cnt4; e += 16) {
      const int li = e / g4;
      const f32x4 da = gld4(p.dA + (row0 + li) * (size_t)p.ldA + c);
      const f32x4 xh = gld4(p.xhat + (row0 + li) * (size_t)C + c);
      f32x4 dy;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dx = da[q] * mish_grad_f(xh[q] * ga[q] + be[q]) * ga[q];
        dy[q] = (dx - m1 - xh[q] * m2) * rstd;
      }
      dbi += dy;
      gst4(p.dY + (row0 + li) * (size_t)C + c, dy);
    }
    }
    // column sums over the positions of this candidate: lanes with the same c4 hold different positions
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dga[q] = stride_sum(dga[q], g4);
      dbe[q] = stride_sum(dbe[q], g4);
      dbi[q] = stride_sum(dbi[q], g4);
      dtb[q] = stride_sum(dtb[q], g4);
    }
    }   // valid
    acc_ga += dga; acc_be += dbe; acc_bi += dbi;                  // sweep order: fixed
    if (p.ptime) {
      if (tl < g4) *(f32x4*)&stage[team * gch + 4 * c4] = dtb;
      if (tl == 0) {
        long long tt = valid ? gldll(p.t + b) : 0;
        tts[team] = (int)(tt < 0 ? 0 : (tt >= T ? T - 1 : tt));
      }
      __syncthreads();
      if ((int)threadIdx.x < 8 * gch) {                           // one thread per column, the group's eight teams in order
        const int g = threadIdx.x / gch, cc = threadIdx.x - g * gch;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int tm = g + 8 * k;
          ttab[tts[tm] * C + g * gch + cc] += stage[tm * gch + cc];
        }
      }
      __syncthreads();
    }
  }
  if (tl < g4) {
    *(f32x4*)&stage[(team * 3 + 0) * gch + 4 * c4] = acc_ga;
    *(f32x4*)&stage[(team * 3 + 1) * gch + 4 * c4] = acc_be;
    *(f32x4*)&stage[(team * 3 + 2) * gch + 4 * c4] = acc_bi;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 1024) {
    const int kind = i / C, col = i - kind * C, g = col / gch, cc = col - g * gch;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += stage[((g + 8 * k) * 3 + kind) * gch + cc];
    gst(p.pcol + (size_t)blockIdx.x * 3 * C + i, a);
  }
  if (p.ptime)
    for (int i = threadIdx.x; i < T * C; i += 1024) gst(p.ptime + (size_t)blockIdx.x * T * C + i, ttab[i]);
}
hipError_t launch_gnb(const GnbP& p, int nblk, hipStream_t st) {
  const size_t lds = ((size_t)(p.ptime ? p.T * p.C : 0) + (size_t)64 * 3 * p.gch + 64) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;          // trebi_trainer_create refuses such shapes
  hipLaunchKernelGGL(gnb_kernel, dim3(nblk), dim3(1024), lds, st, p);
  return hipGetLastError();
}

// =====================================================================================================
// column sums of a [rows, ld] tensor (bias gradients of the convolutions that have no GroupNorm behind them)
// =====================================================================================================
__global__ __launch_bounds__(256) void colsum_kernel(const float* src, int ld, long rows, int C, float* partial) {
  __shared__ float red[256];
  int CW = 1;
  while (CW < C && CW < 256) CW <<= 1;
  const int c = threadIdx.x % CW, rl = threadIdx.x / CW, nrl = 256 / CW;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  for (int cb = 0; cb < C; cb += CW) {           // (C <= 256 everywhere on this path: one trip); uniform trip count
    const int cc = cb + c;
    float a = 0.f;
    if (cc < C) {
      // four independent loads in flight per lane (one per trip left the loop waiting a memory latency per row)
      float a1 = 0.f, a2 = 0.f, a3 = 0.f;
      long r = r0 + rl;
      for (; r + 3 * nrl < r1; r += 4 * nrl) {
        a += gld(src + r * ld + cc);
        a1 += gld(src + (r + nrl) * ld + cc);
        a2 += gld(src + (r + 2 * nrl) * ld + cc);
        a3 += gld(src + (r + 3 * nrl) * ld + cc);
      }
      for (; r < r1; r += nrl) a += gld(src + r * ld + cc);
      a = (a + a1) + (a2 + a3);
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (rl == 0 && cc < C) {
      for (int k = 1; k < nrl; ++k) a += red[k * CW + c];
      gst(partial + (size_t)blockIdx.x * C + cc, a);
    }
    __syncthreads();
  }
}
hipError_t launch_colsum(const float* src, int ld, long rows, int C, float* partial, int nblk, hipStream_t st) {
  hipLaunchKernelGGL(colsum_kernel, dim3(nblk), dim3(256), 0, st, src, ld, rows, C, partial);
  return hipGetLastError();
}

// dst[i] (+)= sum_blk src[blk*stride + i]: all per-workgroup partial tables of one backward pass in ONE launch
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const RJob* jobs) {
  const RJob j = jobs[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < j.n; i += gridDim.x * 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 3 < j.nblk; b += 4) {
      a0 += gld(j.src + (size_t)b * j.stride + i);
      a1 += gld(j.src + (size_t)(b + 1) * j.stride + i);
      a2 += gld(j.src + (size_t)(b + 2) * j.stride + i);
      a3 += gld(j.src + (size_t)(b + 3) * j.stride + i);
    }
    for (; b < j.nblk; ++b) a0 += gld(j.src + (size_t)b * j.stride + i);
    const float a = (a0 + a1) + (a2 + a3);
    j.dst[i] = j.assign ? a : j.dst[i] + a;
  }
}
hipError_t launch_reduce_jobs(const RJob* dev_jobs, int njobs, int max_n, hipStream_t st) {
  int bx = (max_n + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3(bx, njobs), dim3(256), 0, st, dev_jobs);
  return hipGetLastError();
}

// dst[r, c] += src[r, c] (skip-connection gradients meeting the main path, temporal.py:142)
__global__ void add_cols_kernel(float* dst, int ldd, const float* src, int lds, long rows, int C4) {
  const long total = rows * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C4;
    const int c = 4 * (int)(i - r * C4);
    gst4(dst + r * ldd + c, gld4(dst + r * ldd + c) + gld4(src + r * lds + c));
  }
}
hipError_t launch_add_cols(float* dst, int ldd, const float* src, int lds, long rows, int C, hipStream_t st) {
  const long total = rows * (C / 4);
  int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(add_cols_kernel, dim3(blocks), dim3(256), 0, st, dst, ldd, src, lds, rows, C / 4);
  return hipGetLastError();
}

// =====================================================================================================
// weight gradient GEMM
// =====================================================================================================
#ifdef TREBI_STAMPS          // diagnostic build only: phase cycles of workgroup (0,0), thread 0 of the last wgrad launch
__device__ unsigned long long g_wstamps[16];
extern "C" int trebi_debug_read_wstamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(g_wstamps));
}
#endif
template <int NB>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgP p) {      // two workgroups per CU: at most 256 VGPRs
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lr = lane & 15;
  const int nthr = 64 * p.nwa, TA = 16 * p.nwa, TB = 16 * NB;
  const int nta = p.CA16 / TA;                           // A tiles
  const int ta = blockIdx.x % nta, tb = blockIdx.x / nta;
  const int ca0 = ta * TA, cb0 = tb * TB;
  const int S = p.S, LA = p.LA, LB = p.LB, ldAs = p.ldAs, ldBs = p.ldBs, SA = p.SA, SB = p.SB;
  float* As = sm;
  float* Bs = sm + S * SA;
  const int s_begin = blockIdx.y * p.spw;
  const int s_end = s_begin + p.spw < p.B ? s_begin + p.spw : p.B;
  f32x4 acc[5][NB];
#pragma unroll
  for (int tp = 0; tp < 5; ++tp)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[tp][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ta4 = TA / 4, tb4 = TB / 4;
  const int stride = p.stride, pad = p.pad, k = p.k;
  // Staging is software-pipelined: the global loads of the NEXT chunk of S candidates are issued (into registers)
  // before the MFMA loop of the current one and written to LDS after it, so their latency hides behind the matrix
  // work.  At most 8 float4 per thread and operand (the host sizes S for that).
  constexpr int U = 8;
  f32x4 ra[U], rb[U];
  const int nA = S * LA * ta4, nB = S * LB * tb4;
  // a thread's float4 column is fixed and its rows advance by a constant (nthr is a multiple of ta4 and tb4)
  const int c4a = tid % ta4, rowa0 = tid / ta4, rstepA = nthr / ta4;
  const int c4b = tid % tb4, rowb0 = tid / tb4, rstepB = nthr / tb4;
  const bool chA = ca0 + 4 * c4a < p.CA, chB = cb0 + 4 * c4b < p.CB;
  const FastDiv dLA(LA, p.mLA), dLB(LB, p.mLB);
  const int spadA = SA - LA * ldAs, spadB = SB - LB * ldBs;
  auto fetch = [&](int s0) {                  // (vectorisable operands only: the host checks)
    int r0 = rowa0, r1 = rowb0;
    asm volatile("" : "+v"(r0), "+v"(r1));    // keep the per-item index math inside the chunk loop (registers)
    const int ns = s_end - s0 < S ? s_end - s0 : S;
    const float* ba = p.A + (size_t)s0 * LA * p.ldA;
    const float* bb = p.Bm + (size_t)s0 * LB * p.ldB;
    const unsigned oa = (unsigned)(ca0 + 4 * c4a), ob = (unsigned)(cb0 + 4 * c4b);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r0 + u * rstepA;
      ra[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (chA && row < ns * LA) ra[u] = gld4(ba + ((unsigned)row * (unsigned)p.ldA + oa));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r1 + u * rstepB;
      rb[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (chB && row < ns * LB) rb[u] = gld4(bb + ((unsigned)row * (unsigned)p.ldB + ob));
    }
  };
  auto commit = [&]() {
    int r0 = rowa0, r1 = rowb0;
    asm volatile("" : "+v"(r0), "+v"(r1));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r0 + u * rstepA;
      if (row < S * LA) *(f32x4*)&As[row * ldAs + (int)dLA.div(row) * spadA + 4 * c4a] = ra[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = r1 + u * rstepB;
      if (row < S * LB) *(f32x4*)&Bs[row * ldBs + (int)dLB.div(row) * spadB + 4 * c4b] = rb[u];
    }
  };
  // narrow layers (few waves, long rows) exceed the register budget: they stage synchronously
  auto stage_sync = [&](int s0) {
    for (int i = tid; i < nA; i += nthr) {
      const int row = i / ta4, c4 = i - row * ta4;
      const int s = row / LA, pos = row - s * LA;
      const int ch = ca0 + 4 * c4;
      const int left = (s0 + s < s_end) ? p.CA - ch : 0;
      const float* sp = p.A + ((size_t)s0 * LA + row) * p.ldA + ch;
      *(f32x4*)&As[s * SA + pos * ldAs + 4 * c4] = p.vecA ? load4<true>(sp, left) : load4<false>(sp, left);
    }
    for (int i = tid; i < nB; i += nthr) {
      const int row = i / tb4, c4 = i - row * tb4;
      const int s = row / LB, l = row - s * LB;
      const int ch = cb0 + 4 * c4;
      const int left = (s0 + s < s_end) ? p.CB - ch : 0;
      const float* sp = p.Bm + ((size_t)s0 * LB + row) * p.ldB + ch;
      *(f32x4*)&Bs[s * SB + l * ldBs + 4 * c4] = p.vecB ? load4<true>(sp, left) : load4<false>(sp, left);
    }
  };
  const bool pf = p.pf != 0;
#ifdef TREBI_STAMPS
  unsigned long long tq0 = __builtin_amdgcn_s_memtime(), tq_commit = 0, tq_bar = 0, tq_fetch = 0, tq_mma = 0, tq1, tq2;
#define WQ(acc) { tq2 = __builtin_amdgcn_s_memtime(); acc += tq2 - tq1; tq1 = tq2; }
#else
#define WQ(acc)
#endif
  if (pf && s_begin < s_end) fetch(s_begin);
#ifdef TREBI_STAMPS
  tq1 = __builtin_amdgcn_s_memtime();
#endif
  for (int s0 = s_begin; s0 < s_end; s0 += S) {
    if (pf) commit(); else stage_sync(s0);
    WQ(tq_commit)
    __syncthreads();
    WQ(tq_bar)
    if (pf && s0 + S < s_end) fetch(s0 + S);
    WQ(tq_fetch)
    // ---- MFMA: k-slot g of an instruction = candidate 4*sq + g, all four at the same position ----
    const float* aw = As + g * SA + wave * 16 + lr;
    const float* bw = Bs + g * SB + lr;
    for (int sq = 0; sq < S / 4; ++sq) {
      const float* a_s = aw + sq * 4 * SA;
      const float* b_s = bw + sq * 4 * SB;
      for (int l = 0; l < LB; ++l) {
        float bf[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bf[nb] = b_s[l * ldBs + nb * 16];
        const int pos0 = l * stride - pad;
#pragma unroll
        for (int tp = 0; tp < 5; ++tp) {
          const int pos = pos0 + tp;
          if (tp < k && (unsigned)pos < (unsigned)LA) {           // wave-uniform: zero padding is skipped whole
            const float a = a_s[pos * ldAs];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[tp][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[nb], acc[tp][nb], 0, 0, 0);
          }
        }
      }
    }
    WQ(tq_mma)
    __syncthreads();
    WQ(tq_bar)
  }
#ifdef TREBI_STAMPS
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    g_wstamps[0] = tq_commit; g_wstamps[1] = tq_bar; g_wstamps[2] = tq_fetch; g_wstamps[3] = tq_mma;
    g_wstamps[4] = __builtin_amdgcn_s_memtime() - tq0; g_wstamps[5] = (unsigned long long)((s_end - s_begin + S - 1) / S);
    g_wstamps[6] = (unsigned long long)S; g_wstamps[7] = (unsigned long long)LB; g_wstamps[8] = (unsigned long long)(NB * 100 + p.nwa * 10 + k);
    g_wstamps[9] = (unsigned long long)p.CA16 * 10000 + p.CB16;
  }
#endif
  // ---- partial tile -> slab[ks][tap][cb][ca]: D[i][j]: lane holds rows i = 4g..4g+3 (ca), column j = lr (cb) ----
  float* out = p.slab + (size_t)blockIdx.y * p.slab_stride;
#pragma unroll
  for (int tp = 0; tp < 5; ++tp)
    if (tp < k) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int cb = cb0 + nb * 16 + lr, ca = ca0 + wave * 16 + 4 * g;
        gst4(out + ((size_t)tp * p.CB16 + cb) * p.CA16 + ca, acc[tp][nb]);
      }
    }
}

__global__ void time_mlp_bwd_kernel(const float* DT, int njobs, const float* e, const float* a1, const float* temb,
                                    const float* W3, int dim, int T, float* gW1, float* gb1, float* gW3, float* gb3);
hipError_t train_init_device() {
  hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)wgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)time_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)gnb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

hipError_t launch_wgrad(const WgP& p, int ksplit, hipStream_t st) {
  if (p.k > 5 || p.nwa < 1 || p.nwa > 4 || p.S % 4 || p.CA16 % (16 * p.nwa) || p.CB16 % (16 * p.NB) ||
      (p.pf && (p.S * p.LA * 4 * p.nwa > 8 * 64 * p.nwa || p.S * p.LB * 4 * p.NB > 8 * 64 * p.nwa)))   // 8 float4 per thread
    return hipErrorInvalidValue;
  const size_t lds = (size_t)p.S * (p.SA + p.SB) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  dim3 grid((p.CA16 / (16 * p.nwa)) * (p.CB16 / (16 * p.NB)), ksplit), block(64 * p.nwa);
  switch (p.NB) {
    case 1: hipLaunchKernelGGL(wgrad_kernel<1>, grid, block, lds, st, p); break;
    case 2: hipLaunchKernelGGL(wgrad_kernel<2>, grid, block, lds, st, p); break;
    case 4: hipLaunchKernelGGL(wgrad_kernel<4>, grid, block, lds, st, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// grad[(cb*CA + ca)*k + tap] += sum_ks slab[ks][tap][cb][ca]
// A workgroup owns 64 float4 columns of the slab; its 16 waves each sum every 16th slab (four loads in flight per
// lane), the partial sums meet in LDS in a fixed order (bitwise reproducible), then one scattered add per element.
__global__ __launch_bounds__(1024) void wreduce_kernel(const float* slab, int ksplit, long slab_stride, int k, int CA,
                                                       int CB, int CA16, int CB16, float* grad) {
  __shared__ f32x4 part[16][64];
  const int e4 = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const long n4 = (long)k * CB16 * (CA16 / 4);
  const long i4 = (long)blockIdx.x * 64 + e4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (i4 < n4) {
    const float* p = slab + 4 * i4;
    int s = ks;
    for (; s + 48 < ksplit; s += 64) {
      a0 += gld4(p + (size_t)s * slab_stride);
      a1 += gld4(p + (size_t)(s + 16) * slab_stride);
      a2 += gld4(p + (size_t)(s + 32) * slab_stride);
      a3 += gld4(p + (size_t)(s + 48) * slab_stride);
    }
    for (; s < ksplit; s += 16) a0 += gld4(p + (size_t)s * slab_stride);
  }
  part[ks][e4] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ks == 0 && i4 < n4) {
    f32x4 a = part[0][e4];
#pragma unroll
    for (int j = 1; j < 16; ++j) a += part[j][e4];
    const int ca = (int)(i4 % (CA16 / 4)) * 4;
    const long r = i4 / (CA16 / 4);
    const int cb = (int)(r % CB16), tp = (int)(r / CB16);
    if (cb < CB) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ca + q < CA) grad[((size_t)cb * CA + ca + q) * k + tp] += a[q];
    }
  }
}
hipError_t launch_wreduce(const float* slab, int ksplit, long slab_stride, int k, int CA, int CB, int CA16, int CB16,
                          float* grad, hipStream_t st) {
  const long n4 = (long)k * CB16 * (CA16 / 4);
  hipLaunchKernelGGL(wreduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, st, slab, ksplit, slab_stride, k, CA,
                     CB, CA16, CB16, grad);
  return hipGetLastError();
}

// =====================================================================================================
// weight re-pack: the kernels' packed layouts are pure permutations (+ zero fill) of the reference tensors, so a
// map from packed position to flat parameter index (built once, see trebi.hip) turns the re-pack after every
// optimiser step into one gather
// =====================================================================================================
__global__ void gather_kernel(float* dst, const int* map, long n, const float* params) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int m = map[i];
    if (m >= 0) dst[i] = params[m];
  }
}
hipError_t launch_gather(float* dst, const int* map, long n, const float* params, hipStream_t st) {
  int blocks = (int)std::min<long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, st, dst, map, n, params);
  return hipGetLastError();
}
// coded[i] = 1 + flat parameter index as a float (exact below 2^24), 0 = structural zero
__global__ void make_map_kernel(const float* coded, int* map, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    map[i] = (int)coded[i] - 1;
}
hipError_t launch_make_map(const float* coded, int* map, long n, hipStream_t st) {
  int blocks = (int)std::min<long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(make_map_kernel, dim3(blocks), dim3(256), 0, st, coded, map, n);
  return hipGetLastError();
}

// =====================================================================================================
// torch.optim.Adam (bias-corrected, eps added after the corrected sqrt) + the trainer's EMA copy + zero_grad
//   ema_mode 0: leave, 1: ema = p (training.py:95-97), 2: ema = ema*decay + (1-decay)*p (training.py:32-35)
// =====================================================================================================
__global__ void adam_ema_kernel(float* p, float* g, float* m, float* v, float* ema, long n, float b1, float b2,
                                float step_size, float inv_bc2_sqrt, float eps, int ema_mode, float decay) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    const float pi = p[i] - step_size * (mi / denom);
    p[i] = pi;
    g[i] = 0.f;
    if (ema_mode == 1) ema[i] = pi;
    else if (ema_mode == 2) ema[i] = ema[i] * decay + (1.f - decay) * pi;
  }
}
hipError_t launch_adam_ema(float* p, float* g, float* m, float* v, float* ema, long n, float b1, float b2, float step_size,
                           float inv_bc2_sqrt, float eps, int ema_mode, float ema_decay, hipStream_t st) {
  int blocks = (int)std::min<long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(adam_ema_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, ema, n, b1, b2, step_size, inv_bc2_sqrt,
                     eps, ema_mode, ema_decay);
  return hipGetLastError();
}

// =====================================================================================================
// time embedding: forward tables that keep what the backward needs, and the backward
//   e = SinusoidalPosEmb(t);  a1 = W1 e + b1;  temb = W3 mish(a1) + b3      (temporal.py:70-75, helpers.py:21-28)
//   block i: tb_i = Wt_i mish(temb) + bt_i                                    (temporal.py:28-32)
// t only takes the T integer values, so every gradient of this path is a [T, .] table problem.
// =====================================================================================================
__global__ void time_embed_train_kernel(const float* w1, const float* b1, const float* w2, const float* b2, int dim,
                                        int T, float* e_out, float* a1_out, float* temb) {
  extern __shared__ float sm[];
  float* e = sm;             // [dim]
  float* h = sm + dim;       // [4*dim]
  const int t = blockIdx.x, half = dim / 2;
  const float sc = logf(10000.f) / (float)(half - 1);
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    float f = expf((float)i * -sc), a = (float)t * f;
    e[i] = sinf(a);
    e[half + i] = cosf(a);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < dim; i += blockDim.x) e_out[(size_t)t * dim + i] = e[i];
  for (int j = threadIdx.x; j < 4 * dim; j += blockDim.x) {
    float a = b1[j];
    for (int i = 0; i < dim; ++i) a = fmaf(w1[j * dim + i], e[i], a);
    a1_out[(size_t)t * 4 * dim + j] = a;
    h[j] = mish_f(a);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < dim; j += blockDim.x) {
    float a = b2[j];
    for (int i = 0; i < 4 * dim; ++i) a = fmaf(w2[j * 4 * dim + i], h[i], a);
    temb[(size_t)t * dim + j] = a;
  }
}
hipError_t launch_time_embed_train(const float* w1, const float* b1, const float* w2, const float* b2, int dim, int T,
                                   float* e, float* a1, float* temb, hipStream_t st) {
  hipLaunchKernelGGL(time_embed_train_kernel, dim3(T), dim3(128), (size_t)5 * dim * sizeof(float), st, w1, b1, w2, b2, dim,
                     T, e, a1, temb);
  return hipGetLastError();
}

// per block: gW[c][k] += sum_t G[t][c] * mish(temb[t][k]);  gb[c] += sum_t G[t][c];
//            DT[t][k] = sum_c G[t][c] * Wt[c][k]      (the block's contribution to d mish(temb))
__global__ __launch_bounds__(256) void time_block_bwd_kernel(const TBJob* jobs, const float* temb, int dim, int T) {
  const TBJob j = jobs[blockIdx.x];
  const int C = j.C, tid = blockIdx.y * 256 + threadIdx.x, nthr = gridDim.y * 256;
  for (int i = tid; i < C * dim; i += nthr) {
    const int c = i / dim, k = i - c * dim;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a = fmaf(gld(j.G + t * C + c), mish_f(gld(temb + t * dim + k)), a);
    j.gW[i] += a;
  }
  for (int c = tid; c < C; c += nthr) {
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += gld(j.G + t * C + c);
    j.gb[c] += a;
  }
  for (int i = tid; i < T * dim; i += nthr) {
    const int t = i / dim, k = i - t * dim;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < C; c += 4) {               // C is a multiple of 32
      a0 = fmaf(gld(j.G + t * C + c), gld(j.Wt + c * dim + k), a0);
      a1 = fmaf(gld(j.G + t * C + c + 1), gld(j.Wt + (c + 1) * dim + k), a1);
      a2 = fmaf(gld(j.G + t * C + c + 2), gld(j.Wt + (c + 2) * dim + k), a2);
      a3 = fmaf(gld(j.G + t * C + c + 3), gld(j.Wt + (c + 3) * dim + k), a3);
    }
    j.DT[i] = (a0 + a1) + (a2 + a3);
  }
}
hipError_t launch_time_block_bwd(const TBJob* dev_jobs, int njobs, const float* temb, int dim, int T, hipStream_t st) {
  hipLaunchKernelGGL(time_block_bwd_kernel, dim3(njobs, 32), dim3(256), 0, st, dev_jobs, temb, dim, T);
  return hipGetLastError();
}

// d temb = mish'(temb) * sum_blocks DT;  then the two Linears of time_mlp backwards (one workgroup: it is tiny).
// Everything that is read more than once sits in LDS first (Mish(a1), W3, e): the first version evaluated Mish(a1) once per
// OUTPUT row of W3 (dim x too often) and read its operands from global memory inside the reductions: 111 us for a 20 x 32 problem.
constexpr int TMB_THREADS = 1024;
__global__ __launch_bounds__(TMB_THREADS) void time_mlp_bwd_kernel(const float* DT, int njobs, const float* e, const float* a1,
                                                                   const float* temb, const float* W3, int dim, int T,
                                                                   float* gW1, float* gb1, float* gW3, float* gb3) {
  extern __shared__ float sm[];
  const int d4 = 4 * dim, NT = TMB_THREADS;
  float* dte = sm;                     // [T][dim]
  float* da1 = dte + T * dim;          // [T][4dim]
  float* ma1 = da1 + T * d4;           // [T][4dim]   Mish(a1)
  float* ga1 = ma1 + T * d4;           // [T][4dim]   Mish'(a1)
  float* w3s = ga1 + T * d4;           // [dim][4dim]
  float* es = w3s + dim * d4;          // [T][dim]
  for (int i = threadIdx.x; i < T * dim; i += NT) {
    float a = 0.f;
    for (int j = 0; j < njobs; ++j) a += gld(DT + (size_t)j * T * dim + i);
    dte[i] = a * mish_grad_f(gld(temb + i));
    es[i] = gld(e + i);
  }
  for (int i = threadIdx.x; i < T * d4; i += NT) {
    float m, dm;
    mish_both(gld(a1 + i), &m, &dm);
    ma1[i] = m;
    ga1[i] = dm;
  }
  for (int i = threadIdx.x; i < dim * d4; i += NT) w3s[i] = gld(W3 + i);
  __syncthreads();
  for (int i = threadIdx.x; i < dim * d4; i += NT) {         // gW3[j][i2] = sum_t dte[t][j] * mish(a1[t][i2])
    const int jj = i / d4, i2 = i - jj * d4;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a = fmaf(dte[t * dim + jj], ma1[t * d4 + i2], a);
    gW3[i] += a;
  }
  for (int jj = threadIdx.x; jj < dim; jj += NT) {
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += dte[t * dim + jj];
    gb3[jj] += a;
  }
  for (int i = threadIdx.x; i < T * d4; i += NT) {           // da1[t][i2] = mish'(a1) * sum_j dte[t][j] W3[j][i2]
    const int t = i / d4, i2 = i - t * d4;
    float a = 0.f;
    for (int jj = 0; jj < dim; ++jj) a = fmaf(dte[t * dim + jj], w3s[jj * d4 + i2], a);
    da1[i] = a * ga1[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < d4 * dim; i += NT) {         // gW1[i2][k] = sum_t da1[t][i2] * e[t][k]
    const int i2 = i / dim, k = i - i2 * dim;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a = fmaf(da1[t * d4 + i2], es[t * dim + k], a);
    gW1[i] += a;
  }
  for (int i2 = threadIdx.x; i2 < d4; i2 += NT) {
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += da1[t * d4 + i2];
    gb1[i2] += a;
  }
}
hipError_t launch_time_mlp_bwd(const float* DT, int njobs, const float* e, const float* a1, const float* temb,
                               const float* W3, int dim, int T, float* gW1, float* gb1, float* gW3, float* gb3,
                               hipStream_t st) {
  const size_t lds = ((size_t)T * (2 * dim + 12 * dim) + (size_t)dim * 4 * dim) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(time_mlp_bwd_kernel, dim3(1), dim3(TMB_THREADS), lds, st, DT, njobs, e, a1, temb, W3, dim, T, gW1, gb1,
                     gW3, gb3);
  return hipGetLastError();
}

}  // namespace trebi
