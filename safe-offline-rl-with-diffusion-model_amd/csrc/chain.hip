// chain.hip -- resident chains: consecutive layers of a network in ONE launch, activations kept in LDS.
//
// At planning batch sizes a layer's matrix work takes ~1 us while a kernel that reads its input from
// global memory, stages it, runs, and stores the result takes 7-9 us of mostly latency.  A workgroup that
// owns S whole candidates and ALL output columns of a layer depends only on its own earlier outputs, so a
// run of such layers executes here as stages of one kernel (rc_stage):
//   * a stage's output is written straight into the LDS layout its consumers read (canonical layout, see
//     RStage in common.h) -- no global round trip, no store drain between layers; tensors that something
//     outside the chain needs (skip connections, saved GroupNorm operands, the chain's result) are also
//     stored to global memory, fire and forget;
//   * stage descriptors are held in VGPRs (DV, device_util.h), loaded with vector loads a stage ahead;
//   * packed weights sit in three 8-step register slots; the first three blocks of the NEXT stage (a whole
//     narrow layer) are requested before this stage's epilogue, long reductions refill a slot as soon as
//     its block is done (branch-free steady state, so the compiler's vmcnt waits are exact);
//   * stages are separated by an LDS-only barrier (s_waitcnt lgkmcnt(0) + s_barrier); outstanding global
//     stores are drained only in front of a stage that reads a global tensor written earlier in the chain.
// Two stage forms share the residency code, each with its own kernel instantiation (one kernel per form
// keeps the register footprint within two waves per SIMD without spilling):
//   DIRECT  a wave owns one 16-row x 16-column MFMA tile over the whole reduction, so the tile is final in
//           registers: GroupNorm statistics by DPP + ds_bpermute (Chan-combined with ONE partner wave
//           through LDS when a candidate spans two row tiles or a group two column tiles), Mish, time bias,
//           residual and stores straight from the accumulators.  Also launched as a single stage with a
//           column-tile grid (blockIdx.y) for wide layers (TREBI_DIRECT_SINGLE, diagnostic).
//   split-K the form of conv_body (conv.hip): K split across waves, partials through LDS, per (candidate,
//           group) teams -- here with the team's values held in registers between the passes.
// The arithmetic is that of conv_body: v_mfma_f32_16x16x4_f32 implicit GEMM, GroupNorm(8) + Mish (+ time
// bias, residual), GroupNorm backward as a prologue of the backward layers.
#include "device_util.h"

namespace trebi {

#ifdef TREBI_STAMPS          // diagnostic build: cycle stamps of workgroup (0,0,0) per (launch shape, stage, phase)
__device__ unsigned long long g_cstamps[96 * 48 * 16];
extern "C" int trebi_debug_read_cstamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cstamps), sizeof(g_cstamps));
}
#define CSTAMP(ph)                                                                                 \
  do {                                                                                             \
    if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0)                                    \
      g_cstamps[((size_t)skey * 48 + sidx) * 16 + (ph)] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#else
#define CSTAMP(ph)
#endif

namespace {

constexpr int WREG = 8;            // float4 weight registers per slot (8 reduction steps of one 16-column tile)
constexpr int TABV = 4;            // float4 registers per thread that carry a chain's parameter table into LDS

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void drain_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Request the weights of reduction steps [kb, kb + WREG/NI) of this wave's K range into w (flat [u][ni]).
__device__ __forceinline__ void wload(f32x4 (&w)[WREG], const float* pw, unsigned wstep, unsigned wlane, int NI,
                                      int kb, int ke) {
  const float* wu = pw + (size_t)kb * wstep + wlane;
  if (NI == 1) {
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
      if (kb + u < ke) w[u] = gld4(wu);
      wu += wstep;
    }
  } else {
#pragma unroll
    for (int u = 0; u < WREG / 2; ++u) {
      if (kb + u < ke) {
        w[2 * u] = gld4(wu);
        w[2 * u + 1] = gld4(wu + 64);
      }
      wu += wstep;
    }
  }
}

// What a wave needs to know about a stage to request its first weights.
struct WInfo {
  const float* w;
  int Npad, NG, KS, NI, chunk, KK, KK1, direct;
  unsigned mNG;
  int n0;                          // first column of this workgroup's tile (single-layer launches)
};
// ---- stage descriptors live in VGPRs ---------------------------------------------------------------
// Reading descriptor fields with scalar loads costs a scalar-cache round trip (hundreds of cycles) at
// every point the compiler decides to (re)load one -- dozens of serialized waits per stage.  Instead each
// wave keeps the stage's descriptor view (its net's ConvP followed by the shared part of RStage) spread
// over the lanes of 4 VGPRs, loaded with ordinary vector loads one stage ahead; a field is then one
// v_readlane_b32 with a constant lane: no memory access, no wait.
constexpr int PW = (int)(sizeof(ConvP) / 4);
constexpr int TAIL0 = (int)offsetof(RStage, g);
constexpr int VIEW_DW = PW + (int)((sizeof(RStage) - TAIL0) / 4);
static_assert(VIEW_DW <= 256, "descriptor view must fit 4 VGPRs");
static_assert(sizeof(ConvP) % 4 == 0 && TAIL0 % 4 == 0 && sizeof(RStage) % 4 == 0, "dword layout");
__device__ __forceinline__ void dload(DV& d, const RStage* st, int z, int lane) {
  const unsigned* base = (const unsigned*)st;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane + 64 * k;
    const int src = j < PW ? z * PW + j : TAIL0 / 4 + (j - PW);
    d.v[k] = j < VIEW_DW ? *(const TREBI_GLOBAL unsigned*)(base + src) : 0u;
  }
}
#define DPF(d, f) dfield<decltype(ConvP::f), (int)(offsetof(ConvP, f) / 4)>(d)
#define DGF(d, f) dfield<decltype(ConvGeo::f), PW + (int)((offsetof(RStage, g) + offsetof(ConvGeo, f) - TAIL0) / 4)>(d)
#define DSF(d, f) dfield<decltype(RStage::f), PW + (int)((offsetof(RStage, f) - TAIL0) / 4)>(d)
#define DP(f) DPF(dv, f)
#define DG(f) DGF(dv, f)
#define DS(f) DSF(dv, f)

__device__ __forceinline__ WInfo winfo_of(const DV& d) {
  return WInfo{DPF(d, w), DPF(d, Npad), DGF(d, NG), DGF(d, KS), DSF(d, NI), DGF(d, chunk), DGF(d, KK), DGF(d, KK1),
               DSF(d, direct), DGF(d, mNG), DSF(d, direct) ? (int)blockIdx.y * DGF(d, TN) : 0};
}

// Direct stages walk the reduction in blocks of up to WH = WREG/NI steps that never straddle the two
// sources (steps [0, KK1) read the primary source, [KK1, KK) the 1x1 source); block i covers [kb, kb + n).
struct Blk { int kb, n; };
__device__ __forceinline__ Blk blk_of(int i, int sh, int KK1, int KK) {
  const int WH = 1 << sh, nb1 = (KK1 + WH - 1) >> sh;
  const int kb = i < nb1 ? (i << sh) : KK1 + ((i - nb1) << sh);
  const int ke = i < nb1 ? KK1 : KK;
  int n = ke - kb;
  n = n < 0 ? 0 : (n > WH ? WH : n);
  return Blk{kb < KK ? kb : KK, n};                   // past the end: the slack steps behind the layer's weights
}
// Weights of a block into the first n slots of w.  Whole WH-step blocks are requested unconditionally (no
// branches around the loads): the packed weights of every layer are followed by 8 steps of slack.
template <int NI>
__device__ __forceinline__ void wblk(f32x4 (&w)[WREG], const float* pw, unsigned wstep, unsigned wlane, Blk b) {
  constexpr int WH = WREG / NI;
  const float* wu = pw + (size_t)b.kb * wstep + wlane;
#pragma unroll
  for (int u = 0; u < WH; ++u) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) w[u * NI + ni] = gld4(wu + ni * 64);
    wu += wstep;
  }
}

// Request the first weights of a stage: direct stages the first three blocks (a whole narrow layer), split-K
// stages the first block of the wave's K range.  Issued a stage ahead, before the epilogue.
// (DIRECT is the kernel's compile-time stage form; direct stages always have NI == 1 -- chain_tile, plan_direct_single:
// with the form and NI decided at run time both request sequences were compiled into every kernel and the registers they
// share were merged with copies behind vmcnt(0) waits.)
template <bool DIRECT>
__device__ __forceinline__ void wprefetch(const WInfo& q, int wave, int lane, f32x4 (&w)[3][WREG]) {
  if (wave >= q.NG * q.KS) return;
  const int ks = FastDiv(q.NG, q.mNG).div(wave), ng = wave - ks * q.NG;
  const unsigned wlane = (unsigned)(((lane >> 4) * q.Npad + q.n0 + ng * q.NI * 16 + (lane & 15)) * 4);
  const unsigned wstep = (unsigned)q.Npad * 16u;
  if constexpr (DIRECT) {
#pragma unroll
    for (int i = 0; i < 3; ++i) wblk<1>(w[i], q.w, wstep, wlane, blk_of(i, 3, q.KK1, q.KK));
    return;
  }
  const int kb = ks * q.chunk;
  const int ke = kb + q.chunk < q.KK ? kb + q.chunk : q.KK;
  wload(w[0], q.w, wstep, wlane, q.NI, kb, ke);
}

// Sum over aligned groups of `lanes` (4, 8 or 16) lanes of each 16-lane row; VALU only.
__device__ __forceinline__ float row_sum(float v, int lanes) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  if (lanes >= 8) v += dpp_f<0x141>(v);      // row_half_mirror
  if (lanes >= 16) v += dpp_f<0x140>(v);     // row_mirror
  return v;
}

__device__ __forceinline__ f32x4 ld4(const float* sp, int left, bool vec) {
  return vec ? load4<true>(sp, left) : load4<false>(sp, left);
}

// The value head as a chain stage (RStage::head): Linear(fc + dim -> hd) with the time columns folded into the table row,
// Mish, Linear(hd -> 1), and the seed of the backward dL/dv = W1^T (w2 * mish'(h)).  One wave per candidate; W1 is staged
// in LDS once per workgroup (row stride fc + 4 floats: rows for the forward, columns for the backward, both conflict-free).
// Arithmetic and summation order are head_kernel's (kernels.hip), so the two routes agree bit for bit.
__device__ __forceinline__ void head_stage(const DV& dv, const DV& dvn, const bool has_next, float* smem, const int tid,
                                        const int b0, f32x4 (&wreg)[3][WREG], const float* ptab, const int parN) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fc = DP(Cin), hd = DP(N), S = DG(S), B = DP(B), ldw = fc + 4;
  const int nvs = (B - b0 < S) ? B - b0 : S;
  float* w1s = smem + DS(oHead);
  float* dhs = w1s + hd * ldw;
  float* vs = dhs + S * hd;
  const float* w1 = DP(w);
  const int fc4 = fc >> 2;
#pragma clang loop unroll(disable)
  for (int e = tid; e < hd * fc4; e += 512) {
    const int row = e / fc4, c4 = e - row * fc4;
    *(f32x4*)&w1s[row * ldw + 4 * c4] = gld4(w1 + (size_t)row * fc + 4 * c4);
  }
  const bool srcRes = DS(srcRes) != 0;
  if (!srcRes)
    for (int e = tid; e < nvs * fc; e += 512) vs[e] = gld(DP(x) + (size_t)b0 * fc + e);
  const bool dstRes = DS(dstRes) != 0;
  float* dreg = smem + DS(oD);
  const int RbD = DS(RbD), ldD = DS(ldD);
  if (dstRes) {                                          // zero halo rows of the resident destination
    const int ld4n = ldD >> 2, tot = S * 4 * ld4n;
    for (int idx = tid; idx < tot; idx += 512) {
      const int h = idx / ld4n, c4 = idx - h * ld4n;
      const int s = h >> 2, k = h & 3;
      *(f32x4*)&dreg[(s * RbD + (k < 2 ? k : DS(Ld) + k)) * ldD + 4 * c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  lds_barrier();
  if (wave < nvs) {
    const int s = wave;
    const float* v = srcRes ? smem + DS(oSrc) + (s * DS(Rb) + 2) * DG(ldA) : vs + s * fc;
    float ysum = 0.f;
    for (int j = lane; j < hd; j += 64) {
      float h = ptab[3 * parN + j];                      // W1[:, fc:] @ temb(t) + b1
      const float* wr = w1s + j * ldw;
#pragma clang loop unroll(disable)
      for (int i = 0; i < fc; ++i) h = fmaf(wr[i], v[i], h);
      const float w2 = ptab[parN + j];
      ysum += w2 * mish_f(h);
      dhs[s * hd + j] = w2 * mish_grad_f(h);
    }
    ysum = wave_sum(ysum);
    if (lane == 0) gst(DP(out) + b0 + s, ysum + ptab[4 * parN]);
    float* dvg = DP(save_y);
    if (dstRes || dvg) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's dh is in LDS (one wave per candidate)
      for (int i = lane; i < fc; i += 64) {
        float a = 0.f;
#pragma clang loop unroll(disable)
        for (int j = 0; j < hd; ++j) a = fmaf(w1s[j * ldw + i], dhs[s * hd + j], a);
        if (dstRes) dreg[(s * RbD + 2) * ldD + i] = a;
        if (dvg) gst(dvg + (size_t)(b0 + s) * fc + i, a);
      }
    }
  }
  // (behind the head's own arithmetic: with 96 weight registers live across it the kernel spilled)
  if (has_next) wprefetch<true>(winfo_of(dvn), wave, lane, wreg);
}

// FEAT: what the chain's stages may need, decided per launch on the host so that a chain's kernel carries no code for the
// rest (bit 0: a GroupNorm-backward prologue -- value-net backward chains; bit 1: saved GroupNorm operands -- value-net
// forward chains; TemporalUnet chains need neither).  Measured: the less code the kernel carries, the faster the SAME stage
// code runs (register allocation, waits).
template <int MI, int NI, bool DIRECT, int FEAT = 3>
__device__ __forceinline__ void rc_stage(const DV& dv, const DV& dvn, const bool has_next, float* smem, const int tid,
                                         const int b0, int t_uniform, const long long* tptr,
                                         f32x4 (&wreg)[3][WREG], int skey, int sidx, const float* ptab, const int parN,
                                         const f32x4 (&tabv)[TABV], float* tabdst, const int tabn4) {
  constexpr int WH = WREG / NI;    // reduction steps per register slot
  CSTAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const float* px = DP(x);  const float* pw = DP(w);  float* pout = DP(out);
  int ldx = DP(ldx), Cin = DP(Cin), Lin = DP(Lin), Lout = DP(Lout), stride = DP(stride), B = DP(B), N = DP(N), Npad = DP(Npad);
  int S = DG(S), NG = DG(NG), KS = DG(KS), nthr = DG(nthr), TM = DG(TM), TN = DG(TN), TNp = DG(TNp), R = DG(R), ldA = DG(ldA),
      ld2 = DG(ld2);
  int cbpt = DG(cbpt), KK1 = DG(KK1), KK = DG(KK), chunk = DG(chunk), nv = DG(nv), fh = DG(fh);
  int Rb = DS(Rb), a0 = DS(a0), Rb2 = DS(Rb2), h2 = DS(h2);
  PIN(px); PIN(pw); PIN(pout); PIN(ldx); PIN(Cin); PIN(Lin); PIN(Lout); PIN(stride); PIN(B); PIN(N); PIN(Npad);
  PIN(S); PIN(NG); PIN(KS); PIN(nthr); PIN(TM); PIN(TN); PIN(TNp); PIN(R); PIN(ldA); PIN(ld2);
  PIN(cbpt); PIN(KK1); PIN(KK); PIN(chunk); PIN(nv); PIN(fh); PIN(Rb); PIN(a0); PIN(Rb2); PIN(h2);
  CSTAMP(7);
  const bool has2 = DP(Cin2_p) != 0, haspro = (FEAT & 1) && DP(pro) != 0, hasres = DP(res) != nullptr || DS(resRes),
             hastb = DP(tbias) != nullptr, hasgn = DP(gn) != 0;
  const bool vec = DS(vec) != 0;
  const bool active = tid < nthr;
  float* slab = smem + DS(oSrc);
  float* slab2 = smem + DS(o2);
  float* resb = smem + DS(oR);
  float* dreg = smem + DS(oD);
  float* red = smem + DS(oRed);
  float* red2 = smem + DS(oRed2);
  float* colp = smem + DS(oColp);         // [4][TN]: bias, gamma, beta, bias2
  float* tbrow = smem + DS(oTb);          // [S][TN]
  const FastDiv dLout(Lout, DG(mLout)), dTN4(DG(tn4), DG(mTN4)), dTN(TN, DG(mTN));
  const int nvs = (B - b0 < S) ? B - b0 : S;       // candidates of this tile that exist

  const int g = lane >> 4, lr = lane & 15;
  const int ks = FastDiv(NG, DG(mNG)).div(wave), ng = wave - ks * NG;
  // split-K stages: wave = (K split ks, column group ng); direct stages: wave = (row tile ks, column group ng)
  const int k_begin = (active && !DIRECT) ? ks * chunk : 0;
  const int k_end = active ? (DIRECT ? KK : ((k_begin + chunk < KK) ? k_begin + chunk : KK)) : 0;
  const int rowbase = DIRECT ? ks * 16 : 0;
  const int n0 = DIRECT ? (int)blockIdx.y * TN : 0;          // column tile (single-layer launches; chains: 0)
  const unsigned wlane = (unsigned)((g * Npad + n0 + ng * NI * 16 + lr) * 4);
  const unsigned wstep = (unsigned)Npad * 16u;

  // ---- phase A: everything this stage reads from global memory --------------------------------
  // GroupNorm-backward operands of this thread's team (registers): saved mish'(z)*gamma and xhat
  const int ts_pro = DG(ts_pro);
  const int pteam = ts_pro == 64 ? (tid >> 6) : (tid >> 4), ptl = tid & (ts_pro - 1);
  const int pnteams = ts_pro == 64 ? (nthr >> 6) : (nthr >> 4);
  // operands of GroupNorm-backward instance `inst` (candidate, group) for this lane: up to 2 float4 each
  auto pro_fetch = [&](int inst, f32x4 (&gg)[2], f32x4 (&xx)[2], int (&po)[2], int (&pi)[2], float& rs) -> bool {
    po[0] = po[1] = -1;
    pi[0] = pi[1] = 0;
    const int s = inst >> 3, gi = inst & 7;
    if (inst >= S * GN_GROUPS || s >= nvs) return false;
    const int gch = DP(pro_gch), pg4 = DG(pgch4), cnt4 = DG(pcnt4);
    const FastDiv dG(pg4, DG(mPGch4));
    rs = gld(DP(pro_stats) + ((unsigned)(b0 + s) * GN_GROUPS + gi) * 2 + 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = ptl + j * ts_pro;
      if (e < cnt4) {
        const int li = dG.div(e), c = gi * gch + 4 * (e - li * pg4);
        po[j] = (s * Rb + DS(pr0) + li) * ldA + c;
        pi[j] = DS(proRes) ? (s * (Lin + 4) + 2 + li) * ldA + c : po[j];
        const unsigned go = ((unsigned)(b0 + s) * Lin + li) * (unsigned)Cin + c;
        gg[j] = gld4(DP(pro_g) + go);
        xx[j] = gld4(DP(pro_y) + go);
      }
    }
    return true;
  };
  f32x4 pg_[2], px_[2];
  int po_[2] = {-1, -1}, pi_[2] = {0, 0};
  float prstd = 0.f;
  bool pinst = false;
  if (haspro && active) pinst = pro_fetch(pteam, pg_, px_, po_, pi_, prstd);   // first instance: latency overlaps staging
  CSTAMP(8);
  // per-column parameters and time-bias rows: requested now, parked in registers, written to LDS after the K
  // loop (consuming a load here would also wait for the previous stage's stores: vmcnt retires in order).
  // Direct stages: a lane's accumulators all sit in one column per sub-tile, so the parameters simply stay
  // in that lane's registers.
  float c0 = 0.f, c1 = 1.f, c2 = 0.f, c3 = 0.f;
  float tbv[2] = {0.f, 0.f};
  // direct: rows 4*g + q of the wave's 16-row tile -> (candidate, position)
  int sq[4], loq[4];
  bool vq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = rowbase + 4 * g + q;
    sq[q] = dLout.div(r);
    loq[q] = r - sq[q] * Lout;
    vq[q] = DIRECT && active && sq[q] < nvs;
  }
  const bool colt = !DIRECT && active && tid < TN;
  auto tb_load = [&](int e) -> float {
    int s = dTN.div(e), c = e - s * TN;
    if (s >= nvs || c >= N) return 0.f;
    long long tt = t_uniform >= 0 ? t_uniform : gldll(tptr + b0 + s);
    tt = tt < 0 ? 0 : (tt >= DP(T) ? DP(T) - 1 : tt);
    return gld(DP(tbias) + (unsigned)tt * (unsigned)N + c);
  };
  if (active) {
    if (colt) {
      const int n = tid;
      const bool ok = n < N;
      if (ok && DP(bias)) c0 = gld(DP(bias) + n);
      if (ok && hasgn) { c1 = gld(DP(gamma) + n); c2 = gld(DP(beta) + n); }
      if (ok && DP(bias2)) c3 = gld(DP(bias2) + n);
    }
    if (hastb && !DIRECT) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < S * TN) tbv[u] = tb_load(tid + u * nthr);
    }
    CSTAMP(9);
    // identity residual from global memory -> tile [TM][TNp]
    if (hasres && !DS(resRes)) {
      const int nRes4 = DG(nRes4);
      for (int e = tid; e < nRes4; e += nthr) {
        int row = dTN4.div(e), c4 = e - row * DG(tn4);
        int left = N - n0 - 4 * c4;
        if (row >= nvs * Lout) left = 0;
        const float* sp = DP(res) + ((unsigned)b0 * Lout + row) * (unsigned)DP(ldres) + n0 + 4 * c4;
        *(f32x4*)&resb[row * TNp + 4 * c4] = DG(vecr) ? load4<true>(sp, left) : load4<false>(sp, left);
      }
    }
    // primary source from global memory -> slab [S][R][ldA] with a zero halo (with a resident gradient
    // only the halo: the prologue writes the rows)
    if (!DS(srcRes)) {
      const FastDiv dC4(DG(c4n), DG(mC4)), dNv(nv, DG(mNv));
      const int c4n = DG(c4n), total = DS(proRes) ? 0 : DG(nValid4), vlo = DG(vlo);
      for (int base = tid; base < total; base += 4 * nthr) {
        f32x4 v[4];
        int so[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * nthr;
          so[u] = -1;
          if (idx < total) {
            int rowv = dC4.div(idx), c4 = idx - rowv * c4n;
            int s = dNv.div(rowv), i = rowv - s * nv;
            so[u] = (s * R + fh + i) * ldA + 4 * c4;
            unsigned grow = (unsigned)(b0 + s) * Lin + vlo + i;
            int left = s < nvs ? Cin - 4 * c4 : 0;
            v[u] = ld4(px + grow * (unsigned)ldx + 4 * c4, left, vec);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (so[u] >= 0) *(f32x4*)&slab[so[u]] = v[u];
      }
      const FastDiv dHr(DG(hr), DG(mHr));
      const int nh = DG(nHalo4), hr = DG(hr);
      for (int idx = tid; idx < nh; idx += nthr) {
        int h = dC4.div(idx), c4 = idx - h * c4n;
        int s = dHr.div(h), k = h - s * hr;
        int j = k < fh ? k : k + nv;
        *(f32x4*)&slab[(s * R + j) * ldA + 4 * c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (has2 && !DS(x2Res)) {            // second source rows at the output positions -> slab2 [S*Lout][ld2]
      const FastDiv dC4(DG(c4n2), DG(mC4b));
      const int c4n = DG(c4n2), total = DG(nX24), C2 = DP(Cin2);
      const float* p2 = DP(x2);
      const unsigned ldx2 = DP(ldx2);
      for (int base = tid; base < total; base += 2 * nthr) {
        f32x4 v[2];
        int so[2] = {-1, -1};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int idx = base + u * nthr;
          if (idx < total) {
            int row = dC4.div(idx), c4 = idx - row * c4n;
            so[u] = row * ld2 + 4 * c4;
            int left = row < nvs * Lout ? C2 - 4 * c4 : 0;
            v[u] = ld4(p2 + ((unsigned)b0 * Lout + row) * ldx2 + 4 * c4, left, vec);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (so[u] >= 0) *(f32x4*)&slab2[so[u]] = v[u];
      }
    }
    CSTAMP(10);
    // zero halo rows of the resident destination (2 in front, 2 behind every candidate)
    if (DS(dstRes)) {
      const int ld4n = DS(ldD) >> 2, tot = S * 4 * ld4n;
      const FastDiv dL4(ld4n, DS(mLd4));
      for (int idx = tid; idx < tot; idx += nthr) {
        int h = dL4.div(idx), c4 = idx - h * ld4n;
        int s = h >> 2, k = h & 3;
        int row = s * DS(RbD) + (k < 2 ? k : DS(Ld) + k);
        *(f32x4*)&dreg[row * DS(ldD) + 4 * c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  CSTAMP(1);
  if (tabn4 > 0) {                 // first stage of a direct chain: the parameter table requested at kernel start
#pragma unroll
    for (int k = 0; k < TABV; ++k)
      if (tid + 512 * k < tabn4) *(f32x4*)&tabdst[4 * (tid + 512 * k)] = tabv[k];
  }
  if (tabn4 > 0 || (!DS(srcRes) && !DS(proRes)) || (has2 && !DS(x2Res))) lds_barrier();     // staged operands visible to every wave

  // ---- prologue: GroupNorm backward of the producing Conv1dBlock ----------------------------------
  //   d = dA * mish'(z)*gamma;  dY = rstd * (d - mean_g(d) - xhat * mean_g(d * xhat))
  // dA is read from the staged slab or from the resident gradient (left untouched: the residual path of the
  // next layer reads it again); dY always lands in the slab the K loop reads.
  if (haspro) {
    const float* pin = smem + (DS(proRes) ? DS(oPro) : DS(oSrc));
    if (active) {
      const float icnt = DG(picnt);
      for (int inst = pteam; inst < S * GN_GROUPS; inst += pnteams) {
        if (inst != pteam) pinst = pro_fetch(inst, pg_, px_, po_, pi_, prstd);
        f32x4 d[2];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          d[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (pinst && po_[j] >= 0) {
            d[j] = *(const f32x4*)&pin[pi_[j]] * pg_[j];
            s1 += hsum(d[j]);
            s2 += hsum(d[j] * px_[j]);
          }
        }
        const float m1 = team_sum(s1, ts_pro) * icnt, m2 = team_sum(s2, ts_pro) * icnt;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (pinst && po_[j] >= 0) *(f32x4*)&slab[po_[j]] = (d[j] - m1 - px_[j] * m2) * prstd;
      }
    }
    lds_barrier();
  }

  CSTAMP(2);
  // ---- implicit GEMM on the f32 matrix cores, weights double-buffered in registers ------------
  f32x4 acc[MI][NI], acc2[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  if (DIRECT && active) {
    // One 16 x (16*NI) tile per wave over the whole reduction, in blocks of up to WH steps that never
    // straddle the two sources.  A block with n steps uses the LAST n register slots and enters a fully
    // unrolled step sequence at slot WH-n (one branch per block, none per step): all A operands of the
    // block are requested from LDS before its first MFMA -- with one or two waves per SIMD nothing else
    // hides the LDS latency -- and the weights of the next block travel meanwhile.
    const int r = rowbase + lr;
    int s = dLout.div(r), lo = r - s * Lout;
    if (s >= S) { s = 0; lo = 0; }                    // rows past the last candidate: any valid address
    const int offA = DS(oSrc) + (s * Rb + lo * stride + a0) * ldA + 4 * g;
    const int off2 = DS(o2) + (s * Rb2 + h2 + lo) * ld2 + 4 * g;
    constexpr int SH = NI == 1 ? 3 : 2;
    const int nb1 = (KK1 + WH - 1) >> SH, nb = nb1 + ((KK - KK1 + WH - 1) >> SH);
    const FastDiv dCb(cbpt, DG(mCbpt));
    auto run = [&](const f32x4 (&w)[WREG], int i) {
      const Blk bk = blk_of(i, SH, KK1, KK);
      const int kb = bk.kb, n = bk.n;
      f32x4 a[WH];
#define TREBI_STEP(u, ACC)                                                                              \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)       \
      ACC[0][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], w[(u) * NI + ni][j], ACC[0][ni], 0, 0, 0);
#define TREBI_STEPS(ACC)                                                     \
  switch (n) {                                                               \
    case 8: if (WH > 7) { TREBI_STEP(7 % WH, ACC) } [[fallthrough]];         \
    case 7: if (WH > 6) { TREBI_STEP(6 % WH, ACC) } [[fallthrough]];         \
    case 6: if (WH > 5) { TREBI_STEP(5 % WH, ACC) } [[fallthrough]];         \
    case 5: if (WH > 4) { TREBI_STEP(4 % WH, ACC) } [[fallthrough]];         \
    case 4: if (WH > 3) { TREBI_STEP(3 % WH, ACC) } [[fallthrough]];         \
    case 3: if (WH > 2) { TREBI_STEP(2 % WH, ACC) } [[fallthrough]];         \
    case 2: if (WH > 1) { TREBI_STEP(1 % WH, ACC) } [[fallthrough]];         \
    case 1: { TREBI_STEP(0, ACC) } [[fallthrough]];                          \
    default: break;                                                          \
  }
      if (i < nb1) {
        int tau = dCb.div(kb);
        int cb = kb - tau * cbpt;
        int aoff = offA + tau * ldA + cb * 16;
#pragma unroll
        for (int u = 0; u < WH; ++u) {
          a[u] = *(const f32x4*)&smem[u < n ? aoff : offA];       // unused slots: any valid address
          aoff += 16;
          if (++cb == cbpt) { cb = 0; aoff += ldA - cbpt * 16; }
        }
        TREBI_STEPS(acc)
      } else {
        int aoff = off2 + (kb - KK1) * 16;
#pragma unroll
        for (int u = 0; u < WH; ++u) {
          a[u] = *(const f32x4*)&smem[u < n ? aoff : off2];
          aoff += 16;
        }
        TREBI_STEPS(acc2)
      }
#undef TREBI_STEPS
#undef TREBI_STEP
    };
    // Steady state for long reductions: full blocks of source 1 in a branch-free body (so that the waits the
    // compiler inserts are exact: a block only waits for ITS slot, the two younger requests stay in flight).
    // A refill past the last full block is clamped (harmless reload) and redone properly after the loop.
    const int nfull1 = KK1 >> SH;
    auto full = [&](const f32x4 (&w)[WREG], int bi) {
      const int kb = bi << SH;
      const int tau = dCb.div(kb);
      int cb = kb - tau * cbpt;
      int aoff = offA + tau * ldA + cb * 16;
      f32x4 a[WH];
#pragma unroll
      for (int u = 0; u < WH; ++u) {
        a[u] = *(const f32x4*)&smem[aoff];
        aoff += 16;
        if (++cb == cbpt) { cb = 0; aoff += ldA - cbpt * 16; }
      }
#pragma unroll
      for (int u = 0; u < WH; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], w[u * NI + ni][j], acc[0][ni], 0, 0, 0);
    };
    auto wfull = [&](f32x4 (&w)[WREG], int bi) {
      bi = bi < nfull1 ? bi : nfull1 - 1;
      const float* wu = pw + (size_t)(bi << SH) * wstep + wlane;
#pragma unroll
      for (int u = 0; u < WH; ++u) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) w[u * NI + ni] = gld4(wu + ni * 64);
        wu += wstep;
      }
    };
    int i = 0;
    if (nfull1 >= 5) {
      for (; i + 3 <= nfull1; i += 3) {
        full(wreg[0], i);
        wfull(wreg[0], i + 3);
        full(wreg[1], i + 1);
        wfull(wreg[1], i + 4);
        full(wreg[2], i + 2);
        wfull(wreg[2], i + 5);
      }
      // slots whose proper block is not a full block of source 1 were clamped: request the real ones
      if (i + 0 >= nfull1 && i + 0 < nb) wblk<NI>(wreg[0], pw, wstep, wlane, blk_of(i + 0, SH, KK1, KK));
      if (i + 1 >= nfull1 && i + 1 < nb) wblk<NI>(wreg[1], pw, wstep, wlane, blk_of(i + 1, SH, KK1, KK));
      if (i + 2 >= nfull1 && i + 2 < nb) wblk<NI>(wreg[2], pw, wstep, wlane, blk_of(i + 2, SH, KK1, KK));
    }
    // remaining blocks (and short reductions entirely): blocks i..i+2 are in the slots
    for (; i < nb; i += 3) {
      run(wreg[0], i);
      if (i + 3 < nb) wblk<NI>(wreg[0], pw, wstep, wlane, blk_of(i + 3, SH, KK1, KK));
      if (i + 1 < nb) {
        run(wreg[1], i + 1);
        if (i + 4 < nb) wblk<NI>(wreg[1], pw, wstep, wlane, blk_of(i + 4, SH, KK1, KK));
      }
      if (i + 2 < nb) {
        run(wreg[2], i + 2);
        if (i + 5 < nb) wblk<NI>(wreg[2], pw, wstep, wlane, blk_of(i + 5, SH, KK1, KK));
      }
    }
  }
  if (!DIRECT && active) {
    int rowA[MI], row2[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int r = rowbase + mi * 16 + lr;
      int s = dLout.div(r), lo = r - s * Lout;
      bool ok = s < S;
      rowA[mi] = ok ? (s * Rb + lo * stride + a0) * ldA + 4 * g : 4 * g;
      row2[mi] = ok ? (s * Rb2 + h2 + lo) * ld2 + 4 * g : 4 * g;
    }
    const FastDiv dCb(cbpt, DG(mCbpt));
    int cb, aoff;
    {
      int tau = dCb.div(k_begin);
      cb = k_begin - tau * cbpt;
      aoff = tau * ldA + cb * 16;
    }
    auto compute = [&](const f32x4 (&w)[WREG], int kbase) {
#pragma unroll
      for (int u = 0; u < WH; ++u) {
        const int kk = kbase + u;
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
                  acc[mi][ni] =
                      __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], w[u * NI + ni][j], acc[mi][ni], 0, 0, 0);
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
                  acc2[mi][ni] =
                      __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], w[u * NI + ni][j], acc2[mi][ni], 0, 0, 0);
          }
        }
      }
    };
    for (int kb = k_begin; kb < k_end; kb += 2 * WH) {
      const bool more = kb + WH < k_end;
      if (more) wload(wreg[1], pw, wstep, wlane, NI, kb + WH, k_end);
      compute(wreg[0], kb);
      if (more) {
        if (kb + 2 * WH < k_end) wload(wreg[0], pw, wstep, wlane, NI, kb + 2 * WH, k_end);
        compute(wreg[1], kb + WH);
      }
    }
  }
  if (colt) {
    colp[tid] = c0;
    colp[TN + tid] = c1;
    colp[2 * TN + tid] = c2;
    colp[3 * TN + tid] = c3;
  }
  if (hastb && active && !DIRECT) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + u * nthr < S * TN) tbrow[tid + u * nthr] = tbv[u];
    for (int e = tid + 2 * nthr; e < S * TN; e += nthr) tbrow[e] = tb_load(e);
  }
  CSTAMP(3);
  // the next stage's first weight block travels while this stage reduces and normalises
  if (has_next) wprefetch<DIRECT>(winfo_of(dvn), wave, lane, wreg);

  const int kstride = TM * TNp;
  if (!DIRECT) {
  // ---- split-K partials -> LDS.  C/D layout: col = lane&15, row = 4*(lane>>4) + reg -------
  if (active) {
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
  }
    lds_barrier();
  }
  CSTAMP(4);
  if constexpr (DIRECT) {
    // ---- direct epilogue: this wave's 16 x (16*NI) tile is final; everything happens in registers ----
    // MFMA C layout: lane holds column lr of each sub-tile, rows 4*g + q.  The 16 rows belong to ONE
    // candidate (Lout is 16 or 32, checked on the host); with Lout == 32 the other half of the candidate is
    // the wave with the neighbouring row tile, and the two exchange (mean, M2) through LDS once.
    const FastDiv dOs(DP(osplit) > 0 ? DP(osplit) : 1, DG(mOs));
    const int osplit = DP(osplit);
    const unsigned ldo = DP(ldo);
    const int RbR = DS(RbR), hR = DS(hR), ldR = DS(ldR), RbD = DS(RbD), ldD = DS(ldD);
    const bool dstRes = DS(dstRes) != 0;
    // Rows 4*g + q of the tile map to (candidate sq[q], position loq[q]).  A (candidate, group) lives in one
    // wave -- reduced over q, the group's lanes of a row (DPP) and the lane groups g that share the candidate
    // -- except for two cases that need ONE exchange of (mean, M2) with a partner wave through LDS: Lout == 32
    // (the candidate spans two row tiles) and gch == 32 (the group spans two 16-column tiles).
    const int gch = DP(gch);
    const bool two = Lout > 16, twoc = gch > 16;
    const bool xchg = hasgn && (two || twoc);
    float v[NI][4], ml[NI][4], m2[NI][4];
    int cols[NI];
    // per-column parameters: this chain's LDS table [stage][bias, gamma, beta, time bias, bias2][parN], filled at
    // kernel start from the table the host built per (net, timestep).  (They used to be global loads issued at the
    // top of the stage: consuming them here made the epilogue wait -- vmcnt retires in order -- for the next
    // stage's 24 weight requests issued in between.)
    float dbias[NI], dgam[NI], dbet[NI], dbi2[NI], dtb[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float* pc = ptab + n0 + (ng * NI + ni) * 16 + lr;
      dbias[ni] = pc[0]; dgam[ni] = pc[parN]; dbet[ni] = pc[2 * parN]; dtb[ni] = pc[3 * parN]; dbi2[ni] = pc[4 * parN];
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      cols[ni] = n0 + (ng * NI + ni) * 16 + lr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[ni][q] = acc[0][ni][q] + dbias[ni];
        ml[ni][q] = 0.f;
        m2[ni][q] = 0.f;
      }
    }
    float* xch = red;                                  // [wave][NI][64][2]
    const int lrows = Lout < 16 ? Lout : 16;           // rows of a candidate inside this wave's tile
    const int gl = gch < 16 ? gch : 16;                // columns of a group inside a 16-column tile
    if (hasgn && active) {
      const float inl = fast_rcp((float)(lrows * gl));      // a power of two: exact, without the IEEE division sequence
      // per-q partials -> per-q totals over this wave's part of the (candidate, group)
      auto reduce = [&](float (&t)[4]) {
        if (Lout >= 4) {
          float u = (t[0] + t[1]) + (t[2] + t[3]);
          u = row_sum(u, gl);
          if (Lout >= 8) u += __shfl_xor(u, 16, 64);
          if (Lout >= 16) u += __shfl_xor(u, 32, 64);
          t[0] = t[1] = t[2] = t[3] = u;
        } else if (Lout == 2) {
          float a = row_sum(t[0] + t[1], gl), b = row_sum(t[2] + t[3], gl);
          t[0] = t[1] = a;
          t[2] = t[3] = b;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) t[q] = row_sum(t[q], gl);
        }
      };
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = v[ni][q];
        reduce(t);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ml[ni][q] = t[q] * inl;
          const float d = v[ni][q] - ml[ni][q];
          t[q] = d * d;
        }
        reduce(t);
#pragma unroll
        for (int q = 0; q < 4; ++q) m2[ni][q] = t[q];
        if (xchg) {
          float* o = xch + ((wave * NI + ni) * 64 + lane) * 2;
          o[0] = ml[ni][0];
          o[1] = m2[ni][0];
        }
      }
    }
    CSTAMP(11);
    if (xchg) lds_barrier();
    CSTAMP(12);
    if (active) {
      const FastDiv dG(DG(gch4), DG(mGch4));
      float* sy = (FEAT & 2) ? DP(save_y) : nullptr;
      float* sg = (FEAT & 2) ? DP(save_g) : nullptr;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = cols[ni];
        int ph = 0, ch = col;
        if (osplit > 0) {
          ph = dOs.div(col);
          ch = col - ph * osplit;
        }
        float mean = 0.f, rstd = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (hasgn && (q == 0 || Lout < 4)) {
            float M2 = m2[ni][q], cnt = (float)(lrows * gl), icnt = fast_rcp(cnt);     // powers of two
            mean = ml[ni][q];
            if (xchg) {                                  // Chan's combination of two equal-sized parts
              const int pwv = two ? ((ks & 1) ? wave - NG : wave + NG) : (wave ^ 1);
              const float* o = xch + ((pwv * NI + ni) * 64 + lane) * 2;
              const float mp = o[0], m2p = o[1];
              const float dlt = mean - mp;
              M2 = M2 + m2p + dlt * dlt * (0.5f * cnt);
              mean = 0.5f * (mean + mp);
              cnt *= 2.f;
              icnt *= 0.5f;
            }
            rstd = __frsqrt_rn(M2 * icnt + GN_EPS);
          }
          if (!vq[q]) continue;
          const int sw = sq[q], lo = loq[q];
          float val = v[ni][q];
          if (hasgn) {
            if ((FEAT & 2) && DP(save_stats) && lo == 0 && (col & (gch - 1)) == 0) {
              float* o = DP(save_stats) + ((unsigned)(b0 + sw) * GN_GROUPS + dG.div(col >> 2)) * 2;
              gst(o, mean);
              gst(o + 1, rstd);
            }
            const float xh = (val - mean) * rstd;
            float a, b = 0.f;
            if (sg) mish_both(xh * dgam[ni] + dbet[ni], &a, &b);
            else a = mish_f(xh * dgam[ni] + dbet[ni]);
            if (sy && col < N) {
              const unsigned so = ((unsigned)(b0 + sw) * Lout + lo) * (unsigned)N + col;
              gst(sy + so, xh);
              if (sg) gst(sg + so, b * dgam[ni]);
            }
            val = a;
          }
          val += dtb[ni];
          if (has2) val += acc2[0][ni][q] + dbi2[ni];
          if (hasres) val += resb[(sw * RbR + hR + lo) * ldR + col - n0];
          const int pos = osplit > 0 ? 2 * lo + ph : lo;
          if (dstRes) dreg[(sw * RbD + 2 + pos) * ldD + ch] = val;
          if (pout && col < N) {
            const unsigned orow = (unsigned)(b0 + sw) * Lout + lo;
            float* dst = osplit > 0 ? pout + (orow * 2 + ph) * ldo + ch : pout + orow * ldo + col;
            gst(dst, val);
          }
        }
      }
    }
    CSTAMP(5);
    return;
  }
  if (!active) return;

  const FastDiv dOs(DP(osplit) > 0 ? DP(osplit) : 1, DG(mOs));
  const int osplit = DP(osplit);
  const unsigned ldo = DP(ldo);
  const bool veco = DG(veco) != 0;
  const int RbR = DS(RbR), hR = DS(hR), ldR = DS(ldR), RbD = DS(RbD), ldD = DS(ldD);
  const bool dstRes = DS(dstRes) != 0;
  // finish 4 adjacent output columns of (candidate s, position lo): time bias, second-source sum,
  // residual, then the resident copy and / or the global store
  auto finish = [&](f32x4 v, int s, int lo, int col) {
    const int row = s * Lout + lo;
    if (hastb) v += *(const f32x4*)&tbrow[s * TN + col];
    if (has2) {
      f32x4 v2 = *(const f32x4*)&colp[3 * TN + col];
      for (int k = 0; k < KS; ++k) v2 += *(const f32x4*)&red2[k * kstride + row * TNp + col];
      v += v2;
    }
    if (hasres) v += *(const f32x4*)&resb[(s * RbR + hR + lo) * ldR + col];
    int pos = lo, ch = col;
    if (osplit > 0) {
      int ph = dOs.div(col);
      ch = col - ph * osplit;
      pos = 2 * lo + ph;
    }
    if (dstRes) *(f32x4*)&dreg[(s * RbD + 2 + pos) * ldD + ch] = v;
    if (pout) {
      const unsigned orow = (unsigned)b0 * Lout + row;
      float* dst = osplit > 0 ? pout + (orow * 2 + (pos - 2 * lo)) * ldo + ch : pout + orow * ldo + col;
      if (veco) {
        if (col < N) gst4(dst, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < N) gst(dst + e, v[e]);
      }
    }
  };

  if (hasgn) {
    // split-K reduce + bias, GroupNorm statistics, normalise, Mish: a team per (candidate, group), the
    // team's values (at most 2 float4 per lane, checked on the host) stay in registers between passes
    const int gch = DP(gch), g4 = DG(gch4), cnt4 = DG(cnt4), gpt = DG(gpt), ninst = S * gpt;
    const int ts = DG(ts_epi);
    const int team = ts == 64 ? (tid >> 6) : (tid >> 4), tl = tid & (ts - 1),
              nteams = ts == 64 ? (nthr >> 6) : (nthr >> 4);
    const FastDiv dG(g4, DG(mGch4)), dGpt(gpt, DG(mGpt));
    const float icnt = DG(icnt);
    float* sy = DP(save_y);
    float* sg = DP(save_g);
    for (int inst = team; inst < ninst; inst += nteams) {
      int s = dGpt.div(inst), gi = inst - s * gpt;
      if (s >= nvs) continue;
      f32x4 v[2];
      int lo_[2], c_[2];
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = tl + j * ts;
        lo_[j] = -1;
        v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (e < cnt4) {
          int lo = dG.div(e), c = gi * gch + 4 * (e - lo * g4);
          lo_[j] = lo;
          c_[j] = c;
          const int o = (s * Lout + lo) * TNp + c;
          f32x4 a = *(const f32x4*)&colp[c];
          for (int k = 0; k < KS; ++k) a += *(const f32x4*)&red[k * kstride + o];
          v[j] = a;
          sum += hsum(a);
        }
      }
      const float mean = team_sum(sum, ts) * icnt;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (lo_[j] >= 0) {
          f32x4 d = v[j] - mean;
          sq += hsum(d * d);
        }
      const float rstd = __frsqrt_rn(team_sum(sq, ts) * icnt + GN_EPS);
      if (tl == 0 && DP(save_stats)) {
        float* o = DP(save_stats) + ((unsigned)(b0 + s) * GN_GROUPS + gi) * 2;
        gst(o, mean);
        gst(o + 1, rstd);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (lo_[j] >= 0) {
          const int col = c_[j];
          f32x4 xh = (v[j] - mean) * rstd;
          f32x4 ga = *(const f32x4*)&colp[TN + col];
          f32x4 zz = xh * ga + *(const f32x4*)&colp[2 * TN + col];
          f32x4 m, dm;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a, b;
            mish_both(zz[q], &a, &b);
            m[q] = a;
            dm[q] = b;
          }
          if (sy) {
            unsigned so = ((unsigned)b0 * Lout + s * Lout + lo_[j]) * (unsigned)N + col;
            gst4(sy + so, xh);
            gst4(sg + so, dm * ga);
          }
          finish(m, s, lo_[j], col);
        }
    }
  } else {
    const int nitems = DG(nRes4);            // TM * TN / 4
    for (int e = tid; e < nitems; e += nthr) {
      int row = dTN4.div(e), col = 4 * (e - row * DG(tn4));
      if (row >= nvs * Lout || col >= N) continue;
      int s = dLout.div(row);
      f32x4 v = *(const f32x4*)&colp[col];
      for (int k = 0; k < KS; ++k) v += *(const f32x4*)&red[k * kstride + row * TNp + col];
      finish(v, s, row - s * Lout, col);
    }
  }
  CSTAMP(5);
}

}  // namespace

// DIRECT: every stage is a direct stage (one kernel per stage kind keeps each kernel's register
// footprint within two waves per SIMD without spilling).
// HEAD: the chain contains the value head as a stage (head_stage).  An instantiation of its own: with the head's code in the
// kernel every direct chain ran ~6 % slower (the same stage code, other register allocation and waits).
template <bool DIRECT, bool HEAD = false, int FEAT = 3>
__global__ __launch_bounds__(512) void resident_chain_kernel(const RStage* __restrict__ stages, int nstages,
                                                             int t_uniform, const long long* tptr,
                                                             const float* __restrict__ cpar, int parN, int parT, int oTab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4 wreg[3][WREG];
  const int z = blockIdx.z, tid = threadIdx.x;
  // Direct chains: the per-column parameters of ALL stages at this timestep are one contiguous block per (net, t) of a
  // table the host built with the launch plan ([z][t][stage][bias, gamma, beta, time bias, bias2][parN]); requested
  // here, carried in registers past the first stage's own requests and written to LDS in front of its first barrier.
  f32x4 tabv[TABV];
  int tabn4 = 0;
  float* tabdst = smem + oTab;
  if (DIRECT) {
    tabn4 = nstages * 5 * parN / 4;
    const int tt = t_uniform < 0 ? 0 : (t_uniform >= parT ? parT - 1 : t_uniform);
    const float* src = cpar + (size_t)(z * parT + tt) * (size_t)(4 * tabn4);
#pragma unroll
    for (int k = 0; k < TABV; ++k)
      if (tid + 512 * k < tabn4) tabv[k] = gld4(src + 4 * (tid + 512 * k));
  }
  const int skey = (nstages < 48 ? nstages : 47) * 2 + (gridDim.z - 1);
  (void)skey;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DV dv, dvn;
  dload(dv, stages, z, lane);
  dvn = dv;
  wprefetch<DIRECT>(winfo_of(dv), wave, lane, wreg);
  for (int i = 0; i < nstages; ++i) {
    const bool has_next = i + 1 < nstages;
    if (has_next) dload(dvn, stages + i + 1, z, lane);      // consumed a whole stage later
    const int b0 = blockIdx.x * DG(S);
    if (DS(drain)) drain_barrier();
#define RC_ARGS dv, dvn, has_next, smem, tid, b0, t_uniform, tptr, wreg, skey, i, tabdst + i * 5 * parN, parN, tabv, tabdst, \
                (DIRECT && i == 0) ? tabn4 : 0
    if (HEAD && DS(head)) {
      head_stage(dv, dvn, has_next, smem, tid, b0, wreg, tabdst + i * 5 * parN, parN);
    } else if (DIRECT) {
      rc_stage<1, 1, true, FEAT>(RC_ARGS);
    } else {
      switch (DS(MI) * 4 + DS(NI)) {
        case 1 * 4 + 1: rc_stage<1, 1, false>(RC_ARGS); break;
        case 2 * 4 + 1: rc_stage<2, 1, false>(RC_ARGS); break;
        case 1 * 4 + 2: rc_stage<1, 2, false>(RC_ARGS); break;
        default: break;
      }
    }
#undef RC_ARGS
    dv = dvn;
    lds_barrier();
#ifdef TREBI_STAMPS
    { const int sidx = i; CSTAMP(6); }
#endif
  }
}

// One stage's rows of a direct chain's parameter table, for every timestep: dst + t * t_stride holds
// [bias | gamma | beta | time-bias row of t | bias2][parN] (0, 1, 0, 0, 0 where the layer has none / past N).
__global__ void chain_params_kernel(const float* bias, const float* gamma, const float* beta, const float* tbias,
                                    const float* bias2, int N, int parN, float* dst, size_t t_stride) {
  const int t = blockIdx.x;
  float* d = dst + (size_t)t * t_stride;
  for (int n = threadIdx.x; n < parN; n += blockDim.x) {
    const bool ok = n < N;
    d[n] = (ok && bias) ? bias[n] : 0.f;
    d[parN + n] = (ok && gamma) ? gamma[n] : 1.f;
    d[2 * parN + n] = (ok && beta) ? beta[n] : 0.f;
    d[3 * parN + n] = (ok && tbias) ? tbias[(size_t)t * N + n] : 0.f;
    d[4 * parN + n] = (ok && bias2) ? bias2[n] : 0.f;
  }
}
hipError_t launch_chain_params(const ConvP& p, float* dst, int parN, int T, size_t t_stride, hipStream_t st) {
  hipLaunchKernelGGL(chain_params_kernel, dim3(T), dim3(64), 0, st, p.bias, p.gn ? p.gamma : nullptr,
                     p.gn ? p.beta : nullptr, p.tbias, p.bias2, p.N, parN, dst, t_stride);
  return hipGetLastError();
}

bool chain_variant_ok(int MI, int NI) {
  return (NI == 1 && (MI == 1 || MI == 2)) || (NI == 2 && MI == 1);
}

hipError_t chain_init_device() {
  hipError_t e = hipFuncSetAttribute((const void*)resident_chain_kernel<true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)resident_chain_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)resident_chain_kernel<true, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)resident_chain_kernel<true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)resident_chain_kernel<true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)resident_chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
  return e;
}

hipError_t launch_chain(const RStage* dev_stages, int nstages, int direct, int nz, int nblocks, int ny, size_t lds_bytes,
                        int t_uniform, const long long* t, const float* cpar, int parN, int parT, int oTab, int has_head,
                        int feat, hipStream_t st) {
  if (has_head && !direct) return hipErrorInvalidValue;
  if (direct) {
    // the parameter table is indexed by the launch's (uniform) timestep and must fit the registers that carry it
    if (!cpar || t_uniform < 0 || parN < 16 || (parN & 15) || nstages * 5 * parN > 4 * 512 * TABV) return hipErrorInvalidValue;
    if (has_head)
      hipLaunchKernelGGL((resident_chain_kernel<true, true>), dim3(nblocks, ny, nz), dim3(512), lds_bytes, st, dev_stages,
                         nstages, t_uniform, t, cpar, parN, parT, oTab);
    else if (feat == 0)
      hipLaunchKernelGGL((resident_chain_kernel<true, false, 0>), dim3(nblocks, ny, nz), dim3(512), lds_bytes, st, dev_stages,
                         nstages, t_uniform, t, cpar, parN, parT, oTab);
    else if (feat == 1)
      hipLaunchKernelGGL((resident_chain_kernel<true, false, 1>), dim3(nblocks, ny, nz), dim3(512), lds_bytes, st, dev_stages,
                         nstages, t_uniform, t, cpar, parN, parT, oTab);
    else if (feat == 2)
      hipLaunchKernelGGL((resident_chain_kernel<true, false, 2>), dim3(nblocks, ny, nz), dim3(512), lds_bytes, st, dev_stages,
                         nstages, t_uniform, t, cpar, parN, parT, oTab);
    else
      hipLaunchKernelGGL(resident_chain_kernel<true>, dim3(nblocks, ny, nz), dim3(512), lds_bytes, st, dev_stages,
                         nstages, t_uniform, t, cpar, parN, parT, oTab);
  } else
    hipLaunchKernelGGL(resident_chain_kernel<false>, dim3(nblocks, 1, nz), dim3(512), lds_bytes, st, dev_stages,
                       nstages, t_uniform, t, cpar, parN, parT, oTab);
  return hipGetLastError();
}

}  // namespace trebi
