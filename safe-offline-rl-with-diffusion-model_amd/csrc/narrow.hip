// MI355X (gfx950) -- the 32- and 64-channel Conv1dBlocks at LARGE batches (helpers.py:46-63 inside temporal.py:37-46):
// Conv1d(k=5, pad=2) -> GroupNorm(8) -> Mish (+ time bias | + 1x1 residual conv | + identity residual), stride 1, L = 32 / 16.
//
// Why a kernel of its own.  conv_gemm_kernel gives such a layer workgroups of 2-4 candidates that each stream the layer's
// whole weight matrix from L2, stage, multiply for ~1 us and run a GroupNorm epilogue through LDS: at 8192 candidates the
// narrow layers reach 30-70 TFLOP/s and are ~29 % of a planning call (profiles/r03_cfg3_layers.json), bound by the fixed
// costs of a workgroup's lifetime, not by the matrix pipe or HBM.  Here the workgroups are PERSISTENT: a wave owns one
// 16-column tile of the output and keeps that tile's weights for the WHOLE reduction in registers (K/4 VGPRs: 40-96),
// loaded once per launch; the workgroup then walks candidate tiles: stage (registers -> LDS, the next tile's rows are
// requested before the current tile is multiplied), one ds_read_b128 per 4 MFMAs, accumulators final in the wave, so the
// GroupNorm statistics are lane-local sums + six cross-lane adds (no LDS exchange, no barrier), Mish / time bias /
// residual and the stores straight from the accumulators.  3-4 workgroups per CU overlap each other's phases.
//
// Same arithmetic as the other conv paths (exact fp32 products on v_mfma_f32_16x16x4_f32, two-pass GroupNorm, Mish from one
// exponential); the summation order differs, as it does between any two tiles of conv_gemm_kernel.
#include "device_util.h"

#include <cstdint>
#include <cstdlib>

namespace trebi {

namespace {

// NT: 16-column tiles of the output (N = 16 NT);  RT: 16-row tiles of a candidate (L = 16 RT);
// K1: 16-channel reduction steps of the 5-tap source (5 Cin / 16);  K2: steps of the 1x1 second source (Cin2 / 16)
// GN: GroupNorm + Mish epilogue (the forward Conv1dBlocks); without it the epilogue is bias + second source + residual (the
//     input-gradient layers);  PRO: GroupNorm + Mish BACKWARD applied to the staged source first (ConvP::pro, the value nets'
//     backward-to-input): d = dY * saved mish'(z) gamma, then rstd (d - mean(d) - xhat mean(d xhat)) per (candidate, group)
template <int NT, int RT, int K1, int K2, bool GN, bool PRO>
__global__ __launch_bounds__(NT == 4 ? 512 : 256) void narrow_conv_kernel(const ConvLaunch LA, const int ntiles) {
  constexpr int NTHR = NT == 4 ? 512 : 256;    // 4 waves for two column tiles, 8 for four: two candidates per tile either way
  constexpr int S = (NTHR / 64) / NT;          // candidates per tile (one wave per candidate and column tile)
  constexpr int L = 16 * RT, N = 16 * NT, CB = K1 / 5, Cin = 16 * CB, Cin2 = 16 * K2;
  constexpr int GCH = N / GN_GROUPS;           // channels per GroupNorm group: 4 or 8
  constexpr int ldA = Cin + 8, ld2 = Cin2 + 8, RB = L + 4;     // row strides == 8 (mod 16) floats: conflict-free b128 reads
  constexpr int C4 = Cin / 4, C24 = K2 ? Cin2 / 4 : 1;
  constexpr int nA4 = S * L * C4, n24 = K2 ? S * L * C24 : 0;
  constexpr int IA = (nA4 + NTHR - 1) / NTHR, I2 = (n24 + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) float slabA[S * RB * ldA];
  __shared__ __attribute__((aligned(16))) float slab2[K2 ? S * L * ld2 : 4];

  const int z = blockIdx.z;
#define NP(f) (z ? LA.p[1].f : LA.p[0].f)
  const float* px = NP(x);
  const float* px2 = NP(x2);
  const float* pw = NP(w);
  const float* pres = NP(res);
  const float* ptb = NP(tbias);
  const long long* pt = NP(t);
  float* pout = NP(out);
  float* sy = NP(save_y);
  float* sg = NP(save_g);
  float* sst = NP(save_stats);
  const int ldx = NP(ldx), ldx2 = NP(ldx2), ldo = NP(ldo), ldres = NP(ldres), B = NP(B), T = NP(T), t_uniform = NP(t_uniform);

  const int cin1 = NP(Cin), cin2 = NP(Cin2);
  const bool vec1 = cin1 == Cin && (ldx & 3) == 0 && (((uintptr_t)px) & 15) == 0;
  const bool vec2 = K2 && cin2 == Cin2 && (ldx2 & 3) == 0 && (((uintptr_t)px2) & 15) == 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave % NT, slot = wave / NT, g = lane >> 4, lr = lane & 15;
  const int col = ct * 16 + lr;

  // ---- this wave's weights: the whole reduction of its column tile, once per launch ----
  f32x4 w1[K1], w2[K2 ? K2 : 1];
#pragma unroll
  for (int kk = 0; kk < K1; ++kk) w1[kk] = gld4(pw + ((size_t)(kk * 4 + g) * N + col) * 4);
#pragma unroll
  for (int kk = 0; kk < K2; ++kk) w2[kk] = gld4(pw + ((size_t)((K1 + kk) * 4 + g) * N + col) * 4);
  const float bias = (NP(bias) && col < NP(N)) ? gld(NP(bias) + col) : 0.f;
  const float bias2 = (K2 && NP(bias2) && col < NP(N)) ? gld(NP(bias2) + col) : 0.f;
  const float gam = GN ? gld(NP(gamma) + col) : 1.f, bet = GN ? gld(NP(beta) + col) : 0.f;
  const int nreal = NP(N);                     // an input-gradient layer may have fewer real columns than the padded tile (23 of 32)
  const float* ppy = PRO ? NP(pro_y) : nullptr;
  const float* ppg = PRO ? NP(pro_g) : nullptr;
  const float* pps = PRO ? NP(pro_stats) : nullptr;
  // PRO: partial sums of the (candidate, group) means, one row per 64 consecutive float4 of the tile
  constexpr int NCH = (nA4 + 63) / 64, CHS = NCH / S;           // chunks per tile / per candidate
  constexpr int GCP = Cin / GN_GROUPS;                           // channels per group of the SOURCE (4 or 8)
  __shared__ float part[PRO ? NCH * GN_GROUPS * 2 : 2];

  // halo rows are zero for the whole launch: staging only writes rows 2 .. L+1 of a candidate's slab
  for (int i = tid; i < S * 4 * ldA; i += NTHR) {
    const int s = i / (4 * ldA), r = (i / ldA) & 3, c = i % ldA;
    slabA[(s * RB + (r < 2 ? r : L + r)) * ldA + c] = 0.f;
  }

  f32x4 ra[IA], r2[I2 ? I2 : 1], rg[PRO ? IA : 1], rx[PRO ? IA : 1];
  float rrs[PRO ? IA : 1];
  auto request = [&](int tile) {               // this thread's share of a candidate tile, into registers
    const int b0 = tile * S;
    if constexpr (PRO) {
#pragma unroll
      for (int j = 0; j < IA; ++j) {
        const int i = tid + j * NTHR;
        const int s = i / (L * C4), r = (i / C4) % L, c = i % C4;
        rg[j] = rx[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rrs[j] = 0.f;
        if (i < nA4 && b0 + s < B) {
          const size_t o = ((size_t)(b0 + s) * L + r) * Cin + c * 4;
          rg[j] = gld4(ppg + o);
          rx[j] = gld4(ppy + o);
          rrs[j] = gld(pps + ((size_t)(b0 + s) * GN_GROUPS + (c * 4) / GCP) * 2 + 1);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < IA; ++j) {
      const int i = tid + j * NTHR;
      const int s = i / (L * C4), r = (i / C4) % L, c = i % C4;
      ra[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < nA4 && b0 + s < B) {
        const float* src = px + ((size_t)(b0 + s) * L + r) * ldx + c * 4;
        if (vec1) ra[j] = gld4(src);
        else                                   // a caller tensor with an odd row stride (x itself: 23 columns)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c * 4 + e < cin1) ra[j][e] = gld(src + e);
      }
    }
#pragma unroll
    for (int j = 0; j < I2; ++j) {
      const int i = tid + j * NTHR;
      const int s = i / (L * C24), r = (i / C24) % L, c = i % C24;
      r2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < n24 && b0 + s < B) {
        const float* src = px2 + ((size_t)(b0 + s) * L + r) * ldx2 + c * 4;
        if (vec2) r2[j] = gld4(src);
        else
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c * 4 + e < cin2) r2[j][e] = gld(src + e);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < IA; ++j) {
      const int i = tid + j * NTHR;
      const int s = i / (L * C4), r = (i / C4) % L, c = i % C4;
      if (i < nA4) *(f32x4*)&slabA[(s * RB + 2 + r) * ldA + c * 4] = ra[j];
    }
#pragma unroll
    for (int j = 0; j < I2; ++j) {
      const int i = tid + j * NTHR;
      const int s = i / (L * C24), r = (i / C24) % L, c = i % C24;
      if (i < n24) *(f32x4*)&slab2[(s * L + r) * ld2 + c * 4] = r2[j];
    }
  };

  int tile = blockIdx.x;
  if (!PRO && tile < ntiles) request(tile);
  const float* arow = slabA + (slot * RB + lr) * ldA + g * 4;        // row lr of the candidate's slab (+ tap), this lane's 4 channels
  const float* a2row = slab2 + (slot * L + lr) * ld2 + g * 4;
  const float inv_cnt = 1.f / (float)(L * GCH);

  for (; tile < ntiles; tile += gridDim.x) {
    if constexpr (PRO) {
      // requested here, not a tile ahead: three tensors per element would sit in registers under the K loop; the other
      // workgroups of the CU cover the wait
      request(tile);
      float p1[IA], p2[IA];
#pragma unroll
      for (int j = 0; j < IA; ++j) {
        ra[j] = ra[j] * rg[j];                 // d = dY * mish'(z) * gamma
        p1[j] = hsum(ra[j]);
        p2[j] = hsum(ra[j] * rx[j]);
        // sum over the rows of this wave's 64 float4 (lanes with the same float4 column), then over the columns of a group
#pragma unroll
        for (int o = C4; o < 64; o <<= 1) {
          p1[j] += __shfl_xor(p1[j], o, 64);
          p2[j] += __shfl_xor(p2[j], o, 64);
        }
        if (GCP == 8) {
          p1[j] += __shfl_xor(p1[j], 1, 64);
          p2[j] += __shfl_xor(p2[j], 1, 64);
        }
      }
      __syncthreads();                         // the previous tile's operand reads (and its use of `part`) are done
#pragma unroll
      for (int j = 0; j < IA; ++j) {
        const int i = tid + j * NTHR, q = i >> 6;
        if (i < nA4 && lane < C4 && (GCP == 4 || (lane & 1) == 0)) {
          float* o = part + (q * GN_GROUPS + (lane * 4) / GCP) * 2;
          o[0] = p1[j];
          o[1] = p2[j];
        }
      }
      __syncthreads();
      const float inv_pro = 1.f / (float)(L * GCP);
#pragma unroll
      for (int j = 0; j < IA; ++j) {
        const int i = tid + j * NTHR;
        const int s = i / (L * C4), c = i % C4, gi = (c * 4) / GCP;
        float m1 = 0.f, m2 = 0.f;
        if (i < nA4) {
#pragma unroll
          for (int k = 0; k < CHS; ++k) {      // fixed order: bit-reproducible
            m1 += part[((s * CHS + k) * GN_GROUPS + gi) * 2];
            m2 += part[((s * CHS + k) * GN_GROUPS + gi) * 2 + 1];
          }
        }
        m1 *= inv_pro;
        m2 *= inv_pro;
        ra[j] = (ra[j] - m1 - rx[j] * m2) * rrs[j];
      }
      commit();
      __syncthreads();
    } else {
      __syncthreads();                         // the previous tile's operand reads are done
      commit();
      __syncthreads();
      if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);   // in flight under this tile's MFMAs
    }

    const int b = tile * S + slot;
    const bool live = b < B;
    // what the epilogue adds, requested now: identity residual, time-bias row
    float rres[RT][4];
    float tb = 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rres[r][q] = 0.f;
        if (pres && live && col < nreal) rres[r][q] = gld(pres + ((size_t)b * L + r * 16 + 4 * g + q) * ldres + col);
      }
    if (ptb && live && col < nreal) {
      long long tt = t_uniform >= 0 ? (long long)t_uniform : gldll(pt + b);
      tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
      tb = gld(ptb + (size_t)tt * N + col);
    }

    f32x4 acc[RT], acc2[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ---- the reduction: step kk = (tap, 16-channel block); one 16-byte operand read per row tile feeds 4 MFMAs ----
#pragma unroll
    for (int kk = 0; kk < K1; ++kk) {
      const int tap = kk / CB, cb = kk % CB;
      f32x4 a[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) a[r] = *(const f32x4*)(arow + (r * 16 + tap) * ldA + cb * 16);
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, w1[kk].x, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, w1[kk].y, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, w1[kk].z, acc[r], 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, w1[kk].w, acc[r], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kk = 0; kk < K2; ++kk) {
      f32x4 a[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) a[r] = *(const f32x4*)(a2row + (r * 16) * ld2 + kk * 16);
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        acc2[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, w2[kk].x, acc2[r], 0, 0, 0);
        acc2[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, w2[kk].y, acc2[r], 0, 0, 0);
        acc2[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, w2[kk].z, acc2[r], 0, 0, 0);
        acc2[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, w2[kk].w, acc2[r], 0, 0, 0);
      }
    }

    // ---- epilogue in registers.  C layout: lane (g, lr) holds rows 4g + q of column lr of each row tile; a (candidate,
    // group) is GCH adjacent lanes x 4 lane groups x RT tiles x 4 values of ONE wave. ----
    if constexpr (GN) {
      auto group_sum = [&](float u) {
        u += __shfl_xor(u, 1, 64);
        u += __shfl_xor(u, 2, 64);
        if (GCH == 8) u += __shfl_xor(u, 4, 64);
        u += __shfl_xor(u, 16, 64);
        u += __shfl_xor(u, 32, 64);
        return u;
      };
      float v[RT][4];
      float s1 = 0.f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[r][q] = acc[r][q] + bias;
          s1 += v[r][q];
        }
      const float mean = group_sum(s1) * inv_cnt;
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d = v[r][q] - mean;
          s2 = fmaf(d, d, s2);
        }
      const float rstd = __frsqrt_rn(group_sum(s2) * inv_cnt + GN_EPS);
      if (live) {
        if (sst && g == 0 && (lr & (GCH - 1)) == 0) {
          float* o = sst + ((size_t)b * GN_GROUPS + col / GCH) * 2;
          gst(o, mean);
          gst(o + 1, rstd);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const size_t row = (size_t)b * L + r * 16 + 4 * g + q;
            const float xh = (v[r][q] - mean) * rstd;
            float m, dm = 0.f;
            if (sg) mish_both(xh * gam + bet, &m, &dm);
            else m = mish_f(xh * gam + bet);
            if (sy) gst(sy + row * N + col, xh);
            if (sg) gst(sg + row * N + col, dm * gam);
            float val = m + tb;
            if (K2) val += acc2[r][q] + bias2;
            val += rres[r][q];
            gst(pout + row * ldo + col, val);
          }
      }
    } else {
      if (live && col < nreal) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const size_t row = (size_t)b * L + r * 16 + 4 * g + q;
            float val = acc[r][q] + bias + tb;
            if (K2) val += acc2[r][q] + bias2;
            val += rres[r][q];
            gst(pout + row * ldo + col, val);
          }
      }
    }
  }
#undef NP
}

// 0: not a shape of this kernel; otherwise an id of the instantiation
int narrow_variant(const ConvP& p) {
  if (p.osplit || p.stride != 1 || p.ntaps != 5 || p.tap_lo != 0 || p.pad != 2 || p.Lin != p.Lout) return 0;
  if (!p.x || !p.out || (p.x2 && p.res)) return 0;
  const int c1 = p.Cin_p, c2 = p.x2 ? p.Cin2_p : 0;        // sources narrower than their padded width are staged by scalar loads
  if (p.gn) {                                              // forward Conv1dBlocks
    if (p.pro || !p.gamma || !p.beta || p.N != p.Npad || p.gch * GN_GROUPS != p.N) return 0;
    if (p.N == 32 && p.Lout == 32) {
      if (c1 == 32 && c2 == 0) return 1;
      if (c1 == 64 && c2 == 0) return 2;
      if (c1 == 32 && c2 == 64) return 3;
      if (c1 == 32 && c2 == 32) return 7;
    } else if (p.N == 64 && p.Lout == 16) {
      if (c1 == 32 && c2 == 0) return 4;
      if (c1 == 64 && c2 == 0) return 5;
      if (c1 == 64 && c2 == 32) return 6;
    } else if (p.N == 32 && p.Lout == 16) {                // the TemporalUnet's last up level
      if (c1 == 32 && c2 == 0) return 8;
      if (c1 == 32 && c2 == 128) return 9;                 // its first block's second conv: the 1x1 residual reads the concat
      if (c1 == 32 && c2 == 256) return 10;                // (dim_mults (1,4,8))
    }
    return 0;
  }
  if (p.save_y || p.save_g || p.save_stats || p.tbias) return 0;
  if (!p.pro) {                                            // plain input-gradient layers (the training step's backward)
    if (p.Npad == 32 && p.Lout == 32 && c1 == 32 && c2 == 0) return 21;
    if (p.Npad == 64 && p.Lout == 16 && c1 == 64 && c2 == 0) return 22;
    if (p.Npad == 32 && p.Lout == 16 && c1 == 64 && c2 == 64) return 23;
    if (p.Npad == 64 && p.Lout == 32 && c1 == 32 && c2 == 32) return 24;
    return 0;
  }
  // input-gradient layers behind the GroupNorm + Mish backward (the value nets' backward-to-input)
  if (!p.pro_y || !p.pro_g || !p.pro_stats || p.Cin != p.Cin_p || p.pro_gch * GN_GROUPS != p.Cin) return 0;
  if ((p.ldx & 3) || (((uintptr_t)p.x) & 15)) return 0;
  if (p.Npad == 32 && p.Lout == 32) {
    if (c1 == 32 && c2 == 0) return 11;
    if (c1 == 32 && c2 == 32) return 12;
  } else if (p.Npad == 64 && p.Lout == 16) {
    if (c1 == 64 && c2 == 0) return 13;
  } else if (p.Npad == 32 && p.Lout == 16) {
    if (c1 == 64 && c2 == 64) return 14;
  }
  return 0;
}

template <int NT, int RT, int K1, int K2, bool GN, bool PRO>
hipError_t launch_t(const ConvLaunch& L, int nz, hipStream_t st) {
  constexpr int NTHR = NT == 4 ? 512 : 256, S = (NTHR / 64) / NT;
  const int ntiles = (L.p[0].B + S - 1) / S;
  static const int per_cu = getenv("TREBI_NARROW_WGS") ? atoi(getenv("TREBI_NARROW_WGS")) : 4;     // resident workgroups per CU (diagnostics)
  const int cap = 256 * per_cu / nz / (NTHR / 256);
  const int gx = ntiles < cap ? ntiles : cap;
  hipLaunchKernelGGL((narrow_conv_kernel<NT, RT, K1, K2, GN, PRO>), dim3(gx, 1, nz), dim3(NTHR), 0, st, L, ntiles);
  return hipGetLastError();
}

}  // namespace

// The layers this kernel takes: from 512 candidates up (below, a launch is latency-bound and the resident chains /
// conv_gemm_kernel's small tiles are the better tools).  Measured, whole planning calls with / without: 512 candidates (the layers
// the chains leave) 41.4 / 42.4 ms, 1024: 69.2 / 71.5, 2048: 110.6 / 116.5, 8192: 356 / 374.  TREBI_NARROW=0 switches it off,
// TREBI_NARROW_MINB moves the threshold.
bool narrow_ok(const ConvLaunch& L, int nz) {
  static const bool on = !(getenv("TREBI_NARROW") && atoi(getenv("TREBI_NARROW")) == 0);
  static const int min_b = getenv("TREBI_NARROW_MINB") ? atoi(getenv("TREBI_NARROW_MINB")) : 512;
  if (!on || L.p[0].B < min_b) return false;
  const int v = narrow_variant(L.p[0]);
  if (!v) return false;
  if (nz > 1 && (narrow_variant(L.p[1]) != v || L.p[1].B != L.p[0].B || (L.p[0].res == nullptr) != (L.p[1].res == nullptr) ||
                 (L.p[0].tbias == nullptr) != (L.p[1].tbias == nullptr)))
    return false;
  return true;
}

hipError_t launch_narrow(const ConvLaunch& L, int nz, hipStream_t st) {
  switch (narrow_variant(L.p[0])) {
    case 1: return launch_t<2, 2, 10, 0, true, false>(L, nz, st);
    case 2: return launch_t<2, 2, 20, 0, true, false>(L, nz, st);
    case 3: return launch_t<2, 2, 10, 4, true, false>(L, nz, st);
    case 4: return launch_t<4, 1, 10, 0, true, false>(L, nz, st);
    case 5: return launch_t<4, 1, 20, 0, true, false>(L, nz, st);
    case 6: return launch_t<4, 1, 20, 2, true, false>(L, nz, st);
    case 7: return launch_t<2, 2, 10, 2, true, false>(L, nz, st);
    case 8: return launch_t<2, 1, 10, 0, true, false>(L, nz, st);
    case 9: return launch_t<2, 1, 10, 8, true, false>(L, nz, st);
    case 10: return launch_t<2, 1, 10, 16, true, false>(L, nz, st);
    case 11: return launch_t<2, 2, 10, 0, false, true>(L, nz, st);
    case 12: return launch_t<2, 2, 10, 2, false, true>(L, nz, st);
    case 13: return launch_t<4, 1, 20, 0, false, true>(L, nz, st);
    case 14: return launch_t<2, 1, 20, 4, false, true>(L, nz, st);
    case 21: return launch_t<2, 2, 10, 0, false, false>(L, nz, st);
    case 22: return launch_t<4, 1, 20, 0, false, false>(L, nz, st);
    case 23: return launch_t<2, 1, 20, 4, false, false>(L, nz, st);
    case 24: return launch_t<4, 2, 10, 2, false, false>(L, nz, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace trebi
