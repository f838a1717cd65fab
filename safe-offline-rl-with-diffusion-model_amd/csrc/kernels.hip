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

// n / d for the small index ranges used here (host guarantees n * d < 2^32): one mul-hi with a
// host-computed multiplier m = ceil(2^32 / d) (m == 0 encodes d == 1).
struct FastDiv {
  unsigned d, m;
  __device__ FastDiv(unsigned dd, unsigned mm) : d(dd), m(mm) {}
  __device__ __forceinline__ unsigned div(unsigned n) const { return m == 0 ? n : __umulhi(n, m); }
};
static inline unsigned magic_of(unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }

// ----------------------------------------------------------------------------------
// LDS carve (floats), shared between host sizing and the kernel
//   [ stage: slab | slab2 | yslab ]  (dead after the K loop; aliased by the split-K partials)
//   [ restile | colp | tbrow | pstat ]
// ----------------------------------------------------------------------------------
struct Carve {
  int R, ldA, nA, ld2, n2, nY, TM, TN, TNp, red1, body, nres, ncolp, ntb, npstat, total;
};
__host__ __device__ inline Carve make_carve(const ConvP& p, int MI, int NI, int NG, int KS, int S) {
  Carve c;
  c.R = (p.Lout - 1) * p.stride + p.ntaps;
  c.ldA = p.Cin_p + 4;
  c.nA = S * c.R * c.ldA;
  c.ld2 = p.x2 ? p.Cin2_p + 4 : 0;
  c.n2 = p.x2 ? S * p.Lout * c.ld2 : 0;
  c.nY = p.pro ? S * p.Lin * c.ldA : 0;
  c.TM = 16 * MI;
  c.TN = 16 * NI * NG;
  c.TNp = c.TN + 4;
  c.red1 = KS * c.TM * c.TNp;
  int red = c.red1 * (p.x2 ? 2 : 1);
  int stage = c.nA + c.n2 + c.nY;
  c.body = ((stage > red ? stage : red) + 3) & ~3;
  c.nres = p.res ? c.TM * c.TNp : 0;
  c.ncolp = 4 * c.TN;
  c.ntb = p.tbias ? S * c.TN : 0;
  c.npstat = p.pro ? 2 * S * GN_GROUPS : 0;
  c.total = c.body + c.nres + c.ncolp + c.ntb + c.npstat;
  return c;
}

size_t conv_lds_bytes(const ConvP& p, const Tile& tl) {
  Carve c = make_carve(p, tl.MI, tl.NI, tl.NG, tl.KS, tl.S);
  return (size_t)c.total * sizeof(float);
}

// A [rows, C] global tensor -> LDS rows of stride ld (zero filled past C and for invalid rows);
// optionally multiplied elementwise by a second [rows, C] tensor on the way.
struct RowStream {
  const float* src;
  const float* mul;          // dense [.., C] or null
  float* dst;
  int ld, C, c4n, total;
  bool vec, vecm;
  FastDiv dv;
  __device__ RowStream(float* dst_, int ld_, int nrows, int C_, int C_p, const float* src_, int ldsrc_,
                       const float* mul_, unsigned magic)
      : src(src_), mul(mul_), dst(dst_), ld(ld_), C(C_), c4n(C_p >> 2), total(nrows * (C_p >> 2)),
        vec(((ldsrc_ & 3) == 0) && ((C_ & 3) == 0) && ((((uintptr_t)src_) & 15) == 0)),
        vecm(((C_ & 3) == 0) && ((((uintptr_t)mul_) & 15) == 0)), dv(C_p >> 2, magic) {}
  __device__ __forceinline__ f32x4 load4(const float* base, bool v4, int c4, long off) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* sp = base + off + 4 * c4;
    if (v4) {
      if (4 * c4 < C) v = *(const f32x4*)sp;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * c4 + e < C) v[e] = sp[e];
    }
    return v;
  }
};

// Issue up to U loads per thread before the first LDS write, so one memory round trip covers them.
// rowfn(row, &moff) returns the source row offset in floats (< 0: zero row) and the multiplier's.
template <int U, typename RowFn>
__device__ __forceinline__ void stage_stream(const RowStream& rs, int tid, int nthr, RowFn rowfn) {
  for (int base = tid; base < rs.total; base += U * nthr) {
    f32x4 v[U], m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int idx = base + u * nthr;
      v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      m[u] = (f32x4){1.f, 1.f, 1.f, 1.f};
      if (idx < rs.total) {
        int row = rs.dv.div(idx), c4 = idx - row * rs.c4n;
        long moff = 0;
        long off = rowfn(row, &moff);
        if (off >= 0) {
          v[u] = rs.load4(rs.src, rs.vec, c4, off);
          if (rs.mul) m[u] = rs.load4(rs.mul, rs.vecm, c4, moff);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int idx = base + u * nthr;
      if (idx < rs.total) {
        int row = rs.dv.div(idx);
        *(f32x4*)&rs.dst[row * rs.ld + 4 * (idx - row * rs.c4n)] = v[u] * m[u];
      }
    }
  }
}

__device__ __forceinline__ float team_sum(float v, int ts) {
  for (int o = ts >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// WB = reduction steps whose weights a wave preloads into registers before anything else, so the
// weight fetch overlaps the activation staging (one memory latency per kernel instead of one per step).
template <int MI, int NI>
__global__ __launch_bounds__(512) void conv_gemm_kernel(ConvLaunch L, int S, int NG, int KS) {
  constexpr int WB = (NI == 1) ? (MI <= 2 ? 24 : 16) : 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ConvP& p = L.p[blockIdx.z];
  const ConvAux& ax = L.aux;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Carve cv = make_carve(p, MI, NI, NG, KS, S);
  const int R = cv.R, ldA = cv.ldA, ld2 = cv.ld2, TM = cv.TM, TN = cv.TN, TNp = cv.TNp;
  float* slab = smem;
  float* slab2 = slab + cv.nA;
  float* yslab = slab2 + cv.n2;
  float* restile = smem + cv.body;
  float* colp = restile + cv.nres;          // [4][TN]: bias, gamma, beta, bias2
  float* tbrow = colp + cv.ncolp;           // [S][TN]
  float* pstat = tbrow + cv.ntb;            // [S*8][2] mean, rstd of the producing block
  const int b0 = blockIdx.x * S;
  const int n0 = blockIdx.y * TN;
  const int B = p.B, Lin = p.Lin, Lout = p.Lout;
  const int li_min = p.tap_lo - p.pad;
  const FastDiv dLout(Lout, ax.mLout), dTN(TN, ax.mTN);
  STAMP(0);

  // ---- wave geometry + weight preload (registers) -------------------------------------------
  // v_mfma_f32_16x16x4_f32: lane l feeds A[row l&15][k l>>4] and B[k l>>4][col l&15].  The
  // reduction index is permuted so that lane-group g = l>>4 owns channels 16*kk+4g .. +3 of
  // step kk and the four MFMAs of the step consume one float4 of A (LDS) and one of B (global).
  const int g = lane >> 4, lr = lane & 15;
  const int ng = wave % NG, ks = wave / NG;
  const int cbpt = p.Cin_p >> 4;
  const int KK1 = p.ntaps * cbpt;
  const int KK2 = p.x2 ? (p.Cin2_p >> 4) : 0;
  const int KK = KK1 + KK2;
  const int chunk = (KK + KS - 1) / KS;
  const int k_begin = ks * chunk;
  const int k_end = (k_begin + chunk < KK) ? k_begin + chunk : KK;
  // uniform base (scalar registers) + one 32-bit per-lane offset: the loads need no per-lane 64-bit math
  const unsigned wlane = (unsigned)((g * p.Npad + n0 + ng * NI * 16 + lr) * 4);
  const unsigned wstep = (unsigned)p.Npad * 16u;     // floats per kk step
  f32x4 wreg[WB][NI];
#pragma unroll
  for (int u = 0; u < WB; ++u)
    if (k_begin + u < k_end) {
      const float* wu = p.w + (size_t)(k_begin + u) * wstep;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wreg[u][ni] = *(const f32x4*)(wu + wlane + ni * 64);
    }
  STAMP(1);

  // ---- stage everything this workgroup reads from global -----------------------------------------
  // Small per-column / per-candidate parameters first (their loads are issued before the big streams
  // and complete under them), then the activation streams with 4 loads in flight per thread.
  {
    float c0 = 0.f, c1 = 1.f, c2 = 0.f, c3 = 0.f;
    const bool colt = tid < TN;
    if (colt) {
      int n = n0 + tid;
      bool ok = n < p.N;
      if (ok && p.bias) c0 = p.bias[n];
      if (ok && p.gn) { c1 = p.gamma[n]; c2 = p.beta[n]; }
      if (ok && p.bias2) c3 = p.bias2[n];
    }
    // first round of the small per-candidate streams in registers (later rounds, rare, loop below)
    auto tb_load = [&](int e) -> float {
      int s = dTN.div(e), c = e - s * TN;
      int b = b0 + s, n = n0 + c;
      if (b >= B || n >= p.N) return 0.f;
      long long tt = p.t_uniform >= 0 ? p.t_uniform : p.t[b];
      tt = tt < 0 ? 0 : (tt >= p.T ? p.T - 1 : tt);
      return p.tbias[(size_t)tt * p.N + n];
    };
    auto ps_load = [&](int i) -> float {
      return (b0 + i / (GN_GROUPS * 2) < B) ? p.pro_stats[(size_t)b0 * GN_GROUPS * 2 + i] : 0.f;
    };
    auto res_load = [&](int e) -> float {
      int row = dTN.div(e), col = e - row * TN;
      int s = dLout.div(row), n = n0 + col;
      return (s < S && b0 + s < B && n < p.N) ? p.res[((size_t)b0 * Lout + row) * p.ldres + n] : 0.f;
    };
    float tbv[2] = {0.f, 0.f}, psv = 0.f, rv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.tbias) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < S * TN) tbv[u] = tb_load(tid + u * nthr);
    }
    if (p.pro && tid < S * GN_GROUPS * 2) psv = ps_load(tid);
    if (p.res) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tid + u * nthr < TM * TN) rv[u] = res_load(tid + u * nthr);
    }
    {
      // with a prologue the source is dL/d(Mish output) and is multiplied by the saved mish'(z)*gamma
      // on the way in, so the slab holds dL/dxhat
      const FastDiv dR(R, ax.mR);
      RowStream rs(slab, ldA, S * R, p.Cin, p.Cin_p, p.x, p.ldx, p.pro ? p.pro_g : nullptr, ax.mC4);
      stage_stream<4>(rs, tid, nthr, [&](int row, long* moff) -> long {
        int s = dR.div(row), j = row - s * R;
        int li = j + li_min, b = b0 + s;
        if (!(b < B && li >= 0 && li < Lin)) return -1;
        *moff = ((long)b * Lin + li) * p.Cin;
        return ((long)b * Lin + li) * p.ldx;
      });
    }
    if (p.x2) {
      RowStream rs(slab2, ld2, S * Lout, p.Cin2, p.Cin2_p, p.x2, p.ldx2, nullptr, ax.mC4b);
      stage_stream<2>(rs, tid, nthr, [&](int row, long*) -> long {
        int s = dLout.div(row);
        return (b0 + s < B) ? ((long)b0 * Lout + row) * p.ldx2 : -1;
      });
    }
    if (p.pro) {
      const FastDiv dLin(Lin, ax.mLin);
      RowStream rs(yslab, ldA, S * Lin, p.Cin, p.Cin_p, p.pro_y, p.Cin, nullptr, ax.mC4);
      stage_stream<4>(rs, tid, nthr, [&](int row, long*) -> long {
        int s = dLin.div(row);
        return (b0 + s < B) ? ((long)b0 * Lin + row) * p.Cin : -1;
      });
      if (tid < S * GN_GROUPS * 2) pstat[tid] = psv;
      for (int i = tid + nthr; i < S * GN_GROUPS * 2; i += nthr) pstat[i] = ps_load(i);
    }
    if (colt) {
      colp[tid] = c0;
      colp[TN + tid] = c1;
      colp[2 * TN + tid] = c2;
      colp[3 * TN + tid] = c3;
    }
    if (p.tbias) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < S * TN) tbrow[tid + u * nthr] = tbv[u];
      for (int e = tid + 2 * nthr; e < S * TN; e += nthr) tbrow[e] = tb_load(e);
    }
    if (p.res) {
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
  // inside the team (shuffle reductions), so neither the prologue nor the epilogue needs a barrier
  // between its passes.
  const int nteams16 = nthr >> 4;

  // ---- prologue: GroupNorm backward of the producing Conv1dBlock, in LDS ----------------------
  //   slab = dxhat = dA * mish'(z) * gamma (multiplied while staging), yslab = xhat (saved by the forward)
  //   dY = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat))
  if (p.pro) {
    const int gch = p.pro_gch, cnt = Lin * gch, ninst = S * GN_GROUPS;
    const int ts = (ninst * 4 <= nteams16) ? 64 : 16;
    const int team = tid / ts, tl = tid & (ts - 1), nteams = nthr / ts;
    const FastDiv dG(gch, ax.mPGch);
    for (int inst = team; inst < ninst; inst += nteams) {
      int s = inst / GN_GROUPS, gi = inst - s * GN_GROUPS;
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
      const float m1 = team_sum(s1, ts) / (float)cnt, m2 = team_sum(s2, ts) / (float)cnt;
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
    rowA[mi] = ok ? (s * R + lo * p.stride) * ldA + 4 * g : 4 * g;
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
  for (int base = k_begin; base < k_end; base += WB) {
    if (base != k_begin) {                      // later blocks (large reductions only): reload the registers
#pragma unroll
      for (int u = 0; u < WB; ++u)
        if (base + u < k_end) {
          const float* wu = p.w + (size_t)(base + u) * wstep;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) wreg[u][ni] = *(const f32x4*)(wu + wlane + ni * 64);
        }
    }
    int tau = base / cbpt, cb = base - tau * cbpt;
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int kk = base + u;
      if (kk < k_end) {
        f32x4 a[MI];
        if (kk < KK1) {
          const int aoff = tau * ldA + cb * 16;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)&slab[rowA[mi] + aoff];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], wreg[u][ni][j], acc[mi][ni], 0, 0, 0);
          if (++cb == cbpt) { cb = 0; ++tau; }
        } else {
          const int aoff = (kk - KK1) * 16;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)&slab2[row2[mi] + aoff];
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
  float* red2 = smem + cv.red1;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int o = (ks * TM + mi * 16 + g * 4 + q) * TNp + (ng * NI + ni) * 16 + lr;
        red[o] = acc[mi][ni][q];
        if (p.x2) red2[o] = acc2[mi][ni][q];
      }
  __syncthreads();
  STAMP(6);

  const FastDiv dOs(p.osplit > 0 ? p.osplit : 1, ax.mOs);
  // finish one output element: second-source sum, time bias, residual, store
  auto finish = [&](float v, int s, int row, int col) {
    int n = n0 + col;
    if (p.tbias) v += tbrow[s * TN + col];
    if (p.x2) {
      float v2 = colp[3 * TN + col];
      for (int k = 0; k < KS; ++k) v2 += red2[(k * TM + row) * TNp + col];
      v += v2;
    }
    if (p.res) v += restile[row * TNp + col];
    size_t orow = (size_t)b0 * Lout + row;
    if (p.osplit > 0) {
      int ph = dOs.div(n), co = n - ph * p.osplit;
      p.out[(orow * 2 + ph) * p.ldo + co] = v;
    } else {
      p.out[orow * p.ldo + n] = v;
    }
  };

  if (p.gn) {
    // ---- split-K reduce + bias, GroupNorm statistics (two passes), normalise, Mish: per team --------
    const int gch = p.gch, cnt = Lout * gch, gpt = TN / gch, ninst = S * gpt;
    const int ts = (ninst * 4 <= nteams16) ? 64 : 16;
    const int team = tid / ts, tl = tid & (ts - 1), nteams = nthr / ts;
    const FastDiv dG(gch, ax.mGch), dGpt(gpt, ax.mGpt);
    for (int inst = team; inst < ninst; inst += nteams) {
      int s = dGpt.div(inst), gi = inst - s * gpt;
      const int b = b0 + s;
      if (b >= B) continue;
      const int tbase = s * Lout * TNp + gi * gch;
      float sum = 0.f;
#pragma unroll 4
      for (int e = tl; e < cnt; e += ts) {
        int lo = dG.div(e), o = tbase + lo * TNp + (e - lo * gch);
        float v = colp[gi * gch + (e - lo * gch)];
        for (int k = 0; k < KS; ++k) v += red[k * TM * TNp + o];
        red[o] = v;
        sum += v;
      }
      const float mean = team_sum(sum, ts) / (float)cnt;
      float sq = 0.f;
#pragma unroll 4
      for (int e = tl; e < cnt; e += ts) {
        int lo = dG.div(e), o = tbase + lo * TNp + (e - lo * gch);
        float d = red[o] - mean;
        sq += d * d;
      }
      const float rstd = 1.f / sqrtf(team_sum(sq, ts) / (float)cnt + GN_EPS);
      if (tl == 0 && p.save_stats) {
        float* o = p.save_stats + ((size_t)b * GN_GROUPS + (n0 / gch + gi)) * 2;
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
        if (p.save_y) {
          size_t so = ((size_t)b0 * Lout + row) * p.N + n0 + col;
          p.save_y[so] = xh;
          p.save_g[so] = dm * ga;
        }
        finish(m, s, row, col);
      }
    }
  } else {
    const int tile_elems = TM * TN;
    for (int e = tid; e < tile_elems; e += nthr) {
      int row = dTN.div(e), col = e - row * TN;
      int s = dLout.div(row);
      if (s >= S || b0 + s >= B || n0 + col >= p.N) continue;
      float v = colp[col];
      for (int k = 0; k < KS; ++k) v += red[(k * TM + row) * TNp + col];
      finish(v, s, row, col);
    }
  }
  STAMP(9);
}

template <int MI, int NI>
static hipError_t launch_conv_t(const ConvLaunch& L, int nz, const Tile& tl, hipStream_t st) {
  const ConvP& p = L.p[0];
  size_t lds = conv_lds_bytes(p, tl);
  if (nz > 1) {
    size_t l2 = conv_lds_bytes(L.p[1], tl);
    if (l2 > lds) lds = l2;
  }
  dim3 grid((p.B + tl.S - 1) / tl.S, p.Npad / (16 * tl.NI * tl.NG), nz);
  dim3 block(64 * tl.NG * tl.KS);
  ConvLaunch LL = L;
  const int TN = 16 * tl.NI * tl.NG;
  const int R = (p.Lout - 1) * p.stride + p.ntaps;
  ConvAux& ax = LL.aux;
  ax.mLout = magic_of(p.Lout);
  ax.mTN = magic_of(TN);
  ax.mR = magic_of(R);
  ax.mLin = magic_of(p.Lin);
  ax.mGch = magic_of(p.gn ? p.gch : 1);
  ax.mPGch = magic_of(p.pro ? p.pro_gch : 1);
  ax.mGpt = magic_of(p.gn ? TN / p.gch : 1);
  ax.mOs = magic_of(p.osplit > 0 ? p.osplit : 1);
  ax.mC4 = magic_of(p.Cin_p >> 2);
  ax.mC4b = magic_of(p.x2 ? (p.Cin2_p >> 2) : 1);
  ax.mNth16 = 0;
  hipLaunchKernelGGL((conv_gemm_kernel<MI, NI>), grid, block, lds, st, LL, tl.S, tl.NG, tl.KS);
  return hipGetLastError();
}

// Dynamic LDS above 64 KiB has to be opted into per kernel (once per device, at context creation).
hipError_t conv_init_device() {
  hipError_t e = hipSuccess;
#define TREBI_ATTR(mi, ni)                                                                   \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni>,                           \
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
