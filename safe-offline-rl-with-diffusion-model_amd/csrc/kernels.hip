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
#include "common.h"

namespace trebi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef TREBI_STAMPS          // diagnostic build only: per-phase cycle stamps of workgroup (0,0,0), wave 0
__device__ unsigned long long g_stamps[32];
#define STAMP(i)                                                                                  \
  do {                                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {              \
      g_stamps[i] = __builtin_amdgcn_s_memtime();                                                 \
      g_stamps[16 + i] = __builtin_amdgcn_s_memrealtime();                                        \
    }                                                                                             \
  } while (0)
extern "C" int trebi_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#else
#define STAMP(i)
#endif

// Mish and its derivative from ONE exponential: with e = exp(x), n = e*(e+2):
//   tanh(softplus(x)) = n/(n+2),   sigmoid(x) = e/(1+e).
// torch's softplus switches to the identity above 20 (tanh(20) == 1 in f32), so x > 20 returns x / 1.
__device__ __forceinline__ float mish_f(float x) {
  if (x > 20.f) return x;
  float e = __expf(x);
  float n = e * (e + 2.f);
  return x * __fdividef(n, n + 2.f);
}
__device__ __forceinline__ float mish_grad_f(float x) {
  if (x > 20.f) return 1.f;
  float e = __expf(x);
  float n = e * (e + 2.f);
  float th = __fdividef(n, n + 2.f);
  float sg = __fdividef(e, 1.f + e);
  return th + x * sg * (1.f - th * th);
}
__device__ __forceinline__ void mish_both(float x, float* m, float* dm) {
  if (x > 20.f) { *m = x; *dm = 1.f; return; }
  float e = __expf(x);
  float n = e * (e + 2.f);
  float th = __fdividef(n, n + 2.f);
  float sg = __fdividef(e, 1.f + e);
  *m = x * th;
  *dm = th + x * sg * (1.f - th * th);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// n / d with a host-computed multiplier (ConvGeo): one mul-hi.
struct FastDiv {
  unsigned d, m;
  __device__ FastDiv(unsigned dd, unsigned mm) : d(dd), m(mm) {}
  __device__ __forceinline__ unsigned div(unsigned n) const { return m == 0 ? n : __umulhi(n, m); }
};
static inline unsigned magic_of(unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }

static ConvGeo make_geo(const ConvP& p, const Tile& tl) {
  ConvGeo g{};
  g.S = tl.S; g.NG = tl.NG; g.KS = tl.KS;
  g.TM = 16 * tl.MI; g.TN = 16 * tl.NI * tl.NG; g.TNp = g.TN + 4;
  g.R = (p.Lout - 1) * p.stride + p.ntaps;
  g.ldA = p.Cin_p + 4;
  g.ld2 = p.x2 ? p.Cin2_p + 4 : 0;
  g.li_min = p.tap_lo - p.pad;
  g.cbpt = p.Cin_p >> 4;
  g.KK1 = p.ntaps * g.cbpt;
  g.KK = g.KK1 + (p.x2 ? (p.Cin2_p >> 4) : 0);
  g.chunk = (g.KK + g.KS - 1) / g.KS;
  int nA = g.S * g.R * g.ldA, n2 = p.x2 ? g.S * p.Lout * g.ld2 : 0, nY = p.pro ? g.S * p.Lin * g.ldA : 0;
  g.oSlab2 = nA;
  g.oY = nA + n2;
  g.red1 = g.KS * g.TM * g.TNp;
  int red = g.red1 * (p.x2 ? 2 : 1), stage = nA + n2 + nY;
  int body = ((stage > red ? stage : red) + 3) & ~3;
  g.oRes = body;
  g.oColp = g.oRes + (p.res ? g.TM * g.TNp : 0);
  g.oTb = g.oColp + 4 * g.TN;
  g.oPstat = g.oTb + (p.tbias ? g.S * g.TN : 0);
  g.total = g.oPstat + (p.pro ? 2 * g.S * GN_GROUPS : 0);
  g.c4n = p.Cin_p >> 2;
  g.c4n2 = p.x2 ? (p.Cin2_p >> 2) : 1;
  g.nSlab4 = g.S * g.R * g.c4n;
  g.nX24 = p.x2 ? g.S * p.Lout * g.c4n2 : 0;
  g.nY4 = p.pro ? g.S * p.Lin * g.c4n : 0;
  const int nteams16 = 4 * tl.NG * tl.KS;
  g.gpt = p.gn ? g.TN / p.gch : 1;
  g.cnt = p.gn ? p.Lout * p.gch : 1;
  g.pcnt = p.pro ? p.Lin * p.pro_gch : 1;
  g.ts_epi = (g.S * g.gpt * 4 <= nteams16) ? 64 : 16;
  g.ts_pro = (g.S * GN_GROUPS * 4 <= nteams16) ? 64 : 16;
  g.mLout = magic_of(p.Lout); g.mTN = magic_of(g.TN); g.mR = magic_of(g.R); g.mLin = magic_of(p.Lin);
  g.mGch = magic_of(p.gn ? p.gch : 1); g.mPGch = magic_of(p.pro ? p.pro_gch : 1); g.mGpt = magic_of(g.gpt);
  g.mOs = magic_of(p.osplit > 0 ? p.osplit : 1); g.mC4 = magic_of(g.c4n); g.mC4b = magic_of(g.c4n2);
  g.mNG = magic_of(g.NG); g.mCbpt = magic_of(g.cbpt);
  return g;
}

size_t conv_lds_bytes(const ConvP& p, const Tile& tl) { return (size_t)make_geo(p, tl).total * sizeof(float); }

// Keep a wave-uniform value in scalar registers: without this the compiler re-reads kernel-argument
// fields with an s_load + s_waitcnt in front of every use (hundreds of stall cycles per use).
#define PIN(x) asm volatile("" : "+s"(x))

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Sum over a team of 16 lanes (one DPP row) or the whole wave, result in every lane; VALU only.
__device__ __forceinline__ float team_sum(float v, int ts) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror
  if (ts == 64) {           // wave-uniform: the whole wave is one team, every lane is active
    const int iv = __float_as_int(v);
    v = __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16)) +
        __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  }
  return v;
}

// float4 (VEC) or 4 guarded scalars from a row of C valid floats
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* sp, int c0, int C) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (c0 < C) v = *(const f32x4*)sp;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < C) v[e] = sp[e];
  }
  return v;
}

// conv_gemm_kernel<MI, NI, VEC>: VEC = all staged sources are float4-aligned with channel counts % 4 == 0
// (everything except layers touching the [B, horizon, transition_dim] boundary tensors).
// WB = reduction steps whose weights a wave preloads into registers before anything else, so the weight
// fetch overlaps the activation staging (one memory latency per kernel instead of one per step).
template <int MI, int NI, bool VEC>
__global__ __launch_bounds__(512) void conv_gemm_kernel(ConvLaunch L) {
  constexpr int WB = (NI == 1) ? (MI <= 2 ? 24 : 16) : 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ConvP& p = L.p[blockIdx.z];
  const ConvGeo& G = L.g;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- every kernel-argument field used below, read once and pinned in scalar registers ---------
  const float* px = p.x;  const float* pw = p.w;  float* pout = p.out;
  int ldx = p.ldx, Cin = p.Cin, Lin = p.Lin, Lout = p.Lout, stride = p.stride, B = p.B, N = p.N, Npad = p.Npad;
  int S = G.S, NG = G.NG, KS = G.KS, TM = G.TM, TN = G.TN, TNp = G.TNp, R = G.R, ldA = G.ldA, ld2 = G.ld2;
  int li_min = G.li_min, cbpt = G.cbpt, KK1 = G.KK1, KK = G.KK, chunk = G.chunk;
  PIN(px); PIN(pw); PIN(pout); PIN(ldx); PIN(Cin); PIN(Lin); PIN(Lout); PIN(stride); PIN(B); PIN(N); PIN(Npad);
  PIN(S); PIN(NG); PIN(KS); PIN(TM); PIN(TN); PIN(TNp); PIN(R); PIN(ldA); PIN(ld2);
  PIN(li_min); PIN(cbpt); PIN(KK1); PIN(KK); PIN(chunk);
  const bool has2 = p.x2 != nullptr, haspro = p.pro != 0, hasres = p.res != nullptr, hastb = p.tbias != nullptr,
             hasgn = p.gn != 0;
  float* slab = smem;
  float* slab2 = smem + G.oSlab2;
  float* yslab = smem + G.oY;
  float* restile = smem + G.oRes;
  float* colp = smem + G.oColp;          // [4][TN]: bias, gamma, beta, bias2
  float* tbrow = smem + G.oTb;           // [S][TN]
  float* pstat = smem + G.oPstat;        // [S*8][2] mean, rstd of the producing block
  const int b0 = blockIdx.x * S;
  const int n0 = blockIdx.y * TN;
  const FastDiv dLout(Lout, G.mLout), dTN(TN, G.mTN);
  STAMP(0);

  // ---- wave geometry + weight preload (registers) -------------------------------------------
  // v_mfma_f32_16x16x4_f32: lane l feeds A[row l&15][k l>>4] and B[k l>>4][col l&15].  The
  // reduction index is permuted so that lane-group g = l>>4 owns channels 16*kk+4g .. +3 of
  // step kk and the four MFMAs of the step consume one float4 of A (LDS) and one of B (global).
  const int g = lane >> 4, lr = lane & 15;
  const int ks = FastDiv(NG, G.mNG).div(wave), ng = wave - ks * NG;
  const int k_begin = ks * chunk;
  const int k_end = (k_begin + chunk < KK) ? k_begin + chunk : KK;
  // uniform base (scalar registers) + one 32-bit per-lane offset: the loads need no per-lane 64-bit math
  const unsigned wlane = (unsigned)((g * Npad + n0 + ng * NI * 16 + lr) * 4);
  const unsigned wstep = (unsigned)Npad * 16u;     // floats per kk step
  f32x4 wreg[WB][NI];
  {
    const float* wu = pw + (size_t)k_begin * wstep;
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      if (k_begin + u < k_end) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wreg[u][ni] = *(const f32x4*)(wu + wlane + ni * 64);
      }
      wu += wstep;
    }
  }
  STAMP(1);

  // ---- stage everything this workgroup reads from global: all loads first, then the LDS writes ----
  {
    float c0 = 0.f, c1 = 1.f, c2 = 0.f, c3 = 0.f;
    const bool colt = tid < TN;
    if (colt) {
      int n = n0 + tid;
      bool ok = n < N;
      if (ok && p.bias) c0 = p.bias[n];
      if (ok && hasgn) { c1 = p.gamma[n]; c2 = p.beta[n]; }
      if (ok && p.bias2) c3 = p.bias2[n];
    }
    auto tb_load = [&](int e) -> float {
      int s = dTN.div(e), c = e - s * TN;
      int b = b0 + s, n = n0 + c;
      if (b >= B || n >= N) return 0.f;
      long long tt = p.t_uniform >= 0 ? p.t_uniform : p.t[b];
      tt = tt < 0 ? 0 : (tt >= p.T ? p.T - 1 : tt);
      return p.tbias[(unsigned)tt * (unsigned)N + n];
    };
    auto ps_load = [&](int i) -> float {
      return (b0 + i / (GN_GROUPS * 2) < B) ? p.pro_stats[(unsigned)b0 * GN_GROUPS * 2 + i] : 0.f;
    };
    auto res_load = [&](int e) -> float {
      int row = dTN.div(e), col = e - row * TN;
      int s = dLout.div(row), n = n0 + col;
      return (s < S && b0 + s < B && n < N) ? p.res[((unsigned)b0 * Lout + row) * (unsigned)p.ldres + n] : 0.f;
    };
    float tbv[2] = {0.f, 0.f}, psv = 0.f, rv[4] = {0.f, 0.f, 0.f, 0.f};
    if (hastb) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < S * TN) tbv[u] = tb_load(tid + u * nthr);
    }
    if (haspro && tid < S * GN_GROUPS * 2) psv = ps_load(tid);
    if (hasres) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tid + u * nthr < TM * TN) rv[u] = res_load(tid + u * nthr);
    }
    // primary source rows (with halo) -> slab; with a prologue the source is dL/d(Mish output) and is
    // multiplied by the saved mish'(z)*gamma on the way in, so the slab holds dL/dxhat
    {
      const FastDiv dC4(G.c4n, G.mC4), dR(R, G.mR);
      const int c4n = G.c4n, total = G.nSlab4;
      const float* pg = haspro ? p.pro_g : nullptr;
      for (int base = tid; base < total; base += 4 * nthr) {
        f32x4 v[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * nthr;
          v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          m[u] = (f32x4){1.f, 1.f, 1.f, 1.f};
          if (idx < total) {
            int row = dC4.div(idx), c4 = idx - row * c4n;
            int s = dR.div(row), li = row - s * R + li_min, b = b0 + s;
            if (b < B && (unsigned)li < (unsigned)Lin) {
              unsigned grow = (unsigned)b * Lin + li;
              v[u] = load4<VEC>(px + grow * (unsigned)ldx + 4 * c4, 4 * c4, Cin);
              if (pg) m[u] = load4<true>(pg + grow * (unsigned)Cin + 4 * c4, 4 * c4, Cin);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * nthr;
          if (idx < total) {
            int row = dC4.div(idx);
            *(f32x4*)&slab[row * ldA + 4 * (idx - row * c4n)] = v[u] * m[u];
          }
        }
      }
    }
    if (has2) {           // second source rows at the output positions -> slab2
      const FastDiv dC4(G.c4n2, G.mC4b);
      const int c4n = G.c4n2, total = G.nX24, C2 = p.Cin2;
      const float* p2 = p.x2;
      const unsigned ldx2 = p.ldx2;
      for (int base = tid; base < total; base += 2 * nthr) {
        f32x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int idx = base + u * nthr;
          v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (idx < total) {
            int row = dC4.div(idx), c4 = idx - row * c4n;
            if (b0 + (int)dLout.div(row) < B)
              v[u] = load4<VEC>(p2 + ((unsigned)b0 * Lout + row) * ldx2 + 4 * c4, 4 * c4, C2);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int idx = base + u * nthr;
          if (idx < total) {
            int row = dC4.div(idx);
            *(f32x4*)&slab2[row * ld2 + 4 * (idx - row * c4n)] = v[u];
          }
        }
      }
    }
    if (haspro) {         // xhat saved by the forward -> yslab
      const FastDiv dC4(G.c4n, G.mC4), dLin(Lin, G.mLin);
      const int c4n = G.c4n, total = G.nY4;
      const float* py = p.pro_y;
      for (int base = tid; base < total; base += 4 * nthr) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * nthr;
          v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (idx < total) {
            int row = dC4.div(idx), c4 = idx - row * c4n;
            if (b0 + (int)dLin.div(row) < B)
              v[u] = load4<true>(py + ((unsigned)b0 * Lin + row) * (unsigned)Cin + 4 * c4, 4 * c4, Cin);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * nthr;
          if (idx < total) {
            int row = dC4.div(idx);
            *(f32x4*)&yslab[row * ldA + 4 * (idx - row * c4n)] = v[u];
          }
        }
      }
      if (tid < S * GN_GROUPS * 2) pstat[tid] = psv;
      for (int i = tid + nthr; i < S * GN_GROUPS * 2; i += nthr) pstat[i] = ps_load(i);
    }
    if (colt) {
      colp[tid] = c0;
      colp[TN + tid] = c1;
      colp[2 * TN + tid] = c2;
      colp[3 * TN + tid] = c3;
    }
    if (hastb) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < S * TN) tbrow[tid + u * nthr] = tbv[u];
      for (int e = tid + 2 * nthr; e < S * TN; e += nthr) tbrow[e] = tb_load(e);
    }
    if (hasres) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int e = tid + u * nthr;
        if (e < TM * TN) {
          int row = dTN.div(e);
          restile[row * TNp + (e - row * TN)] = rv[u];
        }
      }
      for (int e = tid + 4 * nthr; e < TM * TN; e += nthr) {
        int row = dTN.div(e);
        restile[row * TNp + (e - row * TN)] = res_load(e);
      }
    }
  }
  STAMP(2);
  __syncthreads();
  STAMP(3);

  // Teams of 16 (or 64) lanes each own one (candidate, GroupNorm group): all passes over a group stay
  // inside the team (DPP reductions), so neither the prologue nor the epilogue needs a barrier between
  // its passes.

  // ---- prologue: GroupNorm backward of the producing Conv1dBlock, in LDS ----------------------
  //   slab = dxhat = dA * mish'(z) * gamma (multiplied while staging), yslab = xhat (saved by the forward)
  //   dY = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat))
  if (haspro) {
    const int gch = p.pro_gch, cnt = G.pcnt, ninst = S * GN_GROUPS;
    const int ts = G.ts_pro;
    const int team = ts == 64 ? (tid >> 6) : (tid >> 4), tl = tid & (ts - 1), nteams = ts == 64 ? (nthr >> 6) : (nthr >> 4);
    const FastDiv dG(gch, G.mPGch);
    const float icnt = 1.f / (float)cnt;
    for (int inst = team; inst < ninst; inst += nteams) {
      int s = inst >> 3, gi = inst & 7;
      if (b0 + s >= B) continue;
      const float rstd = pstat[2 * inst + 1];
      const int ybase = s * Lin * ldA + gi * gch, dbase = (s * R - li_min) * ldA + gi * gch;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int e = tl; e < cnt; e += ts) {
        int li = dG.div(e), o = li * ldA + (e - li * gch);
        float d = slab[dbase + o];
        s1 += d;
        s2 += d * yslab[ybase + o];
      }
      const float m1 = team_sum(s1, ts) * icnt, m2 = team_sum(s2, ts) * icnt;
#pragma unroll 4
      for (int e = tl; e < cnt; e += ts) {
        int li = dG.div(e), o = li * ldA + (e - li * gch);
        slab[dbase + o] = rstd * (slab[dbase + o] - m1 - yslab[ybase + o] * m2);
      }
    }
    __syncthreads();
  }
  STAMP(4);

  // ---- implicit GEMM on the f32 matrix cores --------------------------------------------
  int rowA[MI], row2[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int r = mi * 16 + lr;
    int s = dLout.div(r), lo = r - s * Lout;
    bool ok = s < S;
    rowA[mi] = ok ? (s * R + lo * stride) * ldA + 4 * g : 4 * g;
    row2[mi] = ok ? (s * Lout + lo) * ld2 + 4 * g : 4 * g;
  }
  f32x4 acc[MI][NI], acc2[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const FastDiv dCb(cbpt, G.mCbpt);
  for (int base = k_begin; base < k_end; base += WB) {
    if (base != k_begin) {                      // later blocks (large reductions only): reload the registers
      const float* wu = pw + (size_t)base * wstep;
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        if (base + u < k_end) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wreg[u][ni] = *(const f32x4*)(wu + wlane + ni * 64);
        }
        wu += wstep;
      }
    }
    int tau = dCb.div(base), cb = base - tau * cbpt;
    int aoff = tau * ldA + cb * 16;
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int kk = base + u;
      if (kk < k_end) {
        f32x4 a[MI];
        if (kk < KK1) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)&slab[rowA[mi] + aoff];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], wreg[u][ni][j], acc[mi][ni], 0, 0, 0);
          aoff += 16;
          if (++cb == cbpt) { cb = 0; aoff += ldA - cbpt * 16; }
        } else {
          const int a2 = (kk - KK1) * 16;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)&slab2[row2[mi] + a2];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc2[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], wreg[u][ni][j], acc2[mi][ni], 0, 0, 0);
        }
      }
    }
  }
  STAMP(5);
  __syncthreads();     // every wave is done with the staged activations; LDS becomes the reduce buffer

  // ---- split-K partials -> LDS.  C/D layout: col = lane&15, row = 4*(lane>>4) + reg -------
  float* red = smem;
  float* red2 = smem + G.red1;
  const int kstride = TM * TNp;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int o = ks * kstride + (mi * 16 + g * 4 + q) * TNp + (ng * NI + ni) * 16 + lr;
        red[o] = acc[mi][ni][q];
        if (has2) red2[o] = acc2[mi][ni][q];
      }
  __syncthreads();
  STAMP(6);

  const FastDiv dOs(p.osplit > 0 ? p.osplit : 1, G.mOs);
  const int osplit = p.osplit;
  const unsigned ldo = p.ldo;
  // finish one output element: second-source sum, time bias, residual, store
  auto finish = [&](float v, int s, int row, int col) {
    int n = n0 + col;
    if (hastb) v += tbrow[s * TN + col];
    if (has2) {
      float v2 = colp[3 * TN + col];
      for (int k = 0; k < KS; ++k) v2 += red2[k * kstride + row * TNp + col];
      v += v2;
    }
    if (hasres) v += restile[row * TNp + col];
    unsigned orow = (unsigned)b0 * Lout + row;
    if (osplit > 0) {
      int ph = dOs.div(n), co = n - ph * osplit;
      pout[(orow * 2 + ph) * ldo + co] = v;
    } else {
      pout[orow * ldo + n] = v;
    }
  };

  if (hasgn) {
    // ---- split-K reduce + bias, GroupNorm statistics (two passes), normalise, Mish: per team --------
    const int gch = p.gch, cnt = G.cnt, gpt = G.gpt, ninst = S * gpt;
    const int ts = G.ts_epi;
    const int team = ts == 64 ? (tid >> 6) : (tid >> 4), tl = tid & (ts - 1), nteams = ts == 64 ? (nthr >> 6) : (nthr >> 4);
    const FastDiv dG(gch, G.mGch), dGpt(gpt, G.mGpt);
    const float icnt = 1.f / (float)cnt;
    float* sy = p.save_y;
    float* sg = p.save_g;
    for (int inst = team; inst < ninst; inst += nteams) {
      int s = dGpt.div(inst), gi = inst - s * gpt;
      const int b = b0 + s;
      if (b >= B) continue;
      const int tbase = s * Lout * TNp + gi * gch;
      float sum = 0.f;
#pragma unroll 4
      for (int e = tl; e < cnt; e += ts) {
        int lo = dG.div(e), c = e - lo * gch, o = tbase + lo * TNp + c;
        float v = colp[gi * gch + c];
        for (int k = 0; k < KS; ++k) v += red[k * kstride + o];
        red[o] = v;
        sum += v;
      }
      const float mean = team_sum(sum, ts) * icnt;
      float sq = 0.f;
#pragma unroll 4
      for (int e = tl; e < cnt; e += ts) {
        int lo = dG.div(e), o = tbase + lo * TNp + (e - lo * gch);
        float d = red[o] - mean;
        sq += d * d;
      }
      const float rstd = 1.f / sqrtf(team_sum(sq, ts) * icnt + GN_EPS);
      if (tl == 0 && p.save_stats) {
        float* o = p.save_stats + ((unsigned)b * GN_GROUPS + (n0 / gch + gi)) * 2;
        o[0] = mean;
        o[1] = rstd;
      }
#pragma unroll 2
      for (int e = tl; e < cnt; e += ts) {
        int lo = dG.div(e), col = gi * gch + (e - lo * gch), row = s * Lout + lo;
        float xh = (red[row * TNp + col] - mean) * rstd;
        float ga = colp[TN + col];
        float m, dm;
        mish_both(xh * ga + colp[2 * TN + col], &m, &dm);
        if (sy) {
          unsigned so = ((unsigned)b0 * Lout + row) * (unsigned)N + n0 + col;
          sy[so] = xh;
          sg[so] = dm * ga;
        }
        finish(m, s, row, col);
      }
    }
  } else {
    const int tile_elems = TM * TN;
    for (int e = tid; e < tile_elems; e += nthr) {
      int row = dTN.div(e), col = e - row * TN;
      int s = dLout.div(row);
      if (s >= S || b0 + s >= B || n0 + col >= N) continue;
      float v = colp[col];
      for (int k = 0; k < KS; ++k) v += red[k * kstride + row * TNp + col];
      finish(v, s, row, col);
    }
  }
  STAMP(9);
}

template <int MI, int NI>
static hipError_t launch_conv_t(const ConvLaunch& L, int nz, const Tile& tl, hipStream_t st) {
  const ConvP& p = L.p[0];
  ConvLaunch LL = L;
  LL.g = make_geo(p, tl);
  size_t lds = (size_t)LL.g.total * sizeof(float);
  dim3 grid((p.B + tl.S - 1) / tl.S, p.Npad / (16 * tl.NI * tl.NG), nz);
  dim3 block(64 * tl.NG * tl.KS);
  auto vec_ok = [](const ConvP& q) {
    bool v = ((q.ldx & 3) == 0) && ((q.Cin & 3) == 0) && ((((uintptr_t)q.x) & 15) == 0);
    if (q.x2) v = v && ((q.ldx2 & 3) == 0) && ((q.Cin2 & 3) == 0) && ((((uintptr_t)q.x2) & 15) == 0);
    if (q.pro) v = v && ((((uintptr_t)q.pro_y) & 15) == 0) && ((((uintptr_t)q.pro_g) & 15) == 0);
    return v;
  };
  bool vec = vec_ok(L.p[0]) && (nz < 2 || vec_ok(L.p[1]));
  if (p.pro && !vec) return hipErrorInvalidValue;       // the saved tensors are always dense and aligned
  if (vec)
    hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, true>), grid, block, lds, st, LL);
  else
    hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, false>), grid, block, lds, st, LL);
  return hipGetLastError();
}

// Dynamic LDS above 64 KiB has to be opted into per kernel (once per device, at context creation).
hipError_t conv_init_device() {
  hipError_t e = hipSuccess;
#define TREBI_ATTR(mi, ni)                                                                   \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni, true>,                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni, false>,                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  TREBI_ATTR(1, 1) TREBI_ATTR(2, 1) TREBI_ATTR(4, 1) TREBI_ATTR(8, 1)
  TREBI_ATTR(1, 2) TREBI_ATTR(2, 2) TREBI_ATTR(4, 2)
#undef TREBI_ATTR
  return e;
}

hipError_t launch_conv(const ConvLaunch& L, int nz, const Tile& tl, hipStream_t st) {
#define TREBI_CASE(mi, ni) \
  if (tl.MI == mi && tl.NI == ni) return launch_conv_t<mi, ni>(L, nz, tl, st);
  TREBI_CASE(1, 1) TREBI_CASE(2, 1) TREBI_CASE(4, 1) TREBI_CASE(8, 1)
  TREBI_CASE(1, 2) TREBI_CASE(2, 2) TREBI_CASE(4, 2)
#undef TREBI_CASE
  return hipErrorInvalidValue;
}

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
__device__ __forceinline__ bool conditioned(const StepP& s, int h, int d, int b, float* val) {
  if (d < s.act || d >= s.act + s.obs) return false;
  for (int k = 0; k < s.n_cond; ++k)
    if (s.cond_index[k] == h) {
      *val = s.cond[k][(size_t)b * s.obs + (d - s.act)];
      return true;
    }
  return false;
}

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

__global__ void fill_t_kernel(long long* t, int B, int value) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) t[i] = value;
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

// Descending rank by counting: rank(i) = #{j : v[j] > v[i] or (v[j] == v[i] and j < i)}.
// NaNs compare false everywhere, so they are ordered among themselves by index after every number.
__global__ __launch_bounds__(256) void argsort_desc_kernel(const float* v, int n, long long* order) {
  __shared__ float tile[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float vi = i < n ? v[i] : 0.f;
  const bool nan_i = vi != vi;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 256) {
    int j = j0 + threadIdx.x;
    tile[threadIdx.x] = j < n ? v[j] : 0.f;
    __syncthreads();
    int m = n - j0 < 256 ? n - j0 : 256;
    for (int k = 0; k < m; ++k) {
      float vj = tile[k];
      int jj = j0 + k;
      bool nan_j = vj != vj;
      bool before = nan_i ? (!nan_j || jj < i) : (!nan_j && (vj > vi || (vj == vi && jj < i)));
      rank += before ? 1 : 0;
    }
    __syncthreads();
  }
  if (i < n) order[rank] = i;
}
hipError_t launch_argsort_desc(const float* v, int n, long long* order, hipStream_t st) {
  hipLaunchKernelGGL(argsort_desc_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, n, order);
  return hipGetLastError();
}

__global__ __launch_bounds__(64) void group_argmax_kernel(const float* v, int per, long long* best) {
  const int gidx = blockIdx.x, lane = threadIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < per; i += 64) {
    float x = v[(size_t)gidx * per + i];
    if (x > bv || (x == bv && i < bi) || bi == 0x7fffffff) { bv = x; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
  }
  if (lane == 0) best[gidx] = (long long)gidx * per + bi;
}
hipError_t launch_group_argmax(const float* v, int ng, int per, long long* best, hipStream_t st) {
  hipLaunchKernelGGL(group_argmax_kernel, dim3(ng), dim3(64), 0, st, v, per, best);
  return hipGetLastError();
}

}  // namespace trebi
