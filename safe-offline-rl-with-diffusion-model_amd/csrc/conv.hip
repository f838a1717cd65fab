// conv.hip -- the implicit-GEMM convolution of the TREBI hot path on gfx950 (CDNA4).
//
// Every Conv1d / ConvTranspose1d / input-gradient convolution of TemporalUnet and ValueFunction
// (reference: diffuser/models/temporal.py:18-46, helpers.py:30-63) is one `conv_body`:
//
//   M = (candidate, position) rows, N = output channels, K = (tap, input channel), on
//   v_mfma_f32_16x16x4_f32 (exact float32: bit for bit an fmaf chain).
//   A workgroup owns S whole candidates x TN columns (TN a multiple of the GroupNorm group):
//     1. every wave preloads the packed weights of its K-chunk from L2 into registers,
//     2. the candidates' activations are staged once into LDS (zero halo), all loads of a thread
//        issued before its first LDS write -- one memory round trip per layer,
//     3. backward only: GroupNorm backward of the producing block in LDS (the staged gradient was
//        already multiplied by the saved mish'(z)*gamma on the way in),
//     4. K loop: one ds_read_b128 + 4 MFMA per 16 channels per 16x16 tile,
//     5. split-K partials through LDS, then per (candidate, group) teams of 16/64 lanes: reduce,
//        bias, GroupNorm statistics (two passes, DPP reductions), normalise, Mish (one exp), time
//        bias, 1x1 / identity residual, float4 stores.
//
// The kernel is latency/issue-bound at planning batch sizes (64 candidates), so the code is written
// for a short instruction stream: all derived integers and division multipliers come from the host
// (ConvGeo), kernel-argument fields are read once and pinned in scalar registers, everything moves
// in float4, and layers whose workgroups are independent are chained inside one launch
// (conv_chain_kernel) with a workgroup barrier instead of a kernel boundary.
#include <type_traits>
#include "device_util.h"

namespace trebi {

#ifdef TREBI_STAMPS
__device__ unsigned long long g_stamps[32];
__device__ unsigned long long g_wave_stamps[8][16];
extern "C" int trebi_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
extern "C" int trebi_debug_read_wave_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_stamps), sizeof(g_wave_stamps));
}
#endif

static inline unsigned magic_of(unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; }

bool conv_vec_ok(const ConvP& q) {
  bool v = ((q.ldx & 3) == 0) && ((q.Cin & 3) == 0) && ((((uintptr_t)q.x) & 15) == 0);
  if (q.x2) v = v && ((q.ldx2 & 3) == 0) && ((q.Cin2 & 3) == 0) && ((((uintptr_t)q.x2) & 15) == 0);
  return v;
}

// The layers and tiles that take the channel-block-major K loop (conv_body, W5).
bool conv_w5_ok(const ConvP& p, const Tile& tl) {
  static const bool w5_on = !(getenv("TREBI_WIN5") && atoi(getenv("TREBI_WIN5")) == 0);
  // from 4096 candidates up: measured +3 % there and at 8192, -1 % at 1024, -3 % at 2048 (value net and UNet passes alone)
  static const int w5_min_b = getenv("TREBI_WIN5_MINB") ? atoi(getenv("TREBI_WIN5_MINB")) : 4096;
  return w5_on && p.B >= w5_min_b && tl.pm && !tl.bf && tl.KS == 1 && p.stride == 1 && p.ntaps == 5 && p.tap_lo == 0 && p.pad == 2 &&
         p.Lin == p.Lout && (p.Lout == tl.MI || (tl.MI >= 4 && 2 * p.Lout == tl.MI)) && ((p.Cin_p >> 4) & 1) == 0;
}

ConvGeo conv_make_geo(const ConvP& p, const Tile& tl) {
  ConvGeo g{};
  g.S = tl.S; g.NG = tl.NG; g.KS = tl.KS; g.nthr = 64 * tl.NG * tl.KS;
  g.TM = 16 * tl.MI; g.TN = 16 * tl.NI * tl.NG; g.TNp = g.TN + 4;
  g.R = (p.Lout - 1) * p.stride + p.ntaps;
  // Row stride == 8 (mod 16) floats: a ds_read_b128 is served in four groups of 16 lanes, each made of 8 tile rows with
  // channel chunk c and the other 8 rows with chunk c+1 (MI355X_MICROARCH.md, LDS).  With a stride of 4*(2*odd) floats the
  // first 8 rows fall on the 8 even bank quads and the others, one chunk on, on the odd ones -- for every tap shift.  (The
  // C + 4 stride this replaces was 2-way on every group: SQ_LDS_BANK_CONFLICT 44-70 % of SQ_LDS_IDX_ACTIVE at 8192
  // candidates, 9-20 % after.)
  g.ldA = p.Cin_p + 8;
  g.ld2 = p.x2 ? p.Cin2_p + 8 : 0;
  const int li_min = p.tap_lo - p.pad;
  g.vlo = li_min > 0 ? li_min : 0;
  int vhi = li_min + g.R < p.Lin ? li_min + g.R : p.Lin;
  g.nv = vhi - g.vlo;
  g.fh = g.vlo - li_min;
  g.hr = g.R - g.nv;
  g.pm = tl.pm;
  g.fhk = g.fh;
  if (tl.pm) {                      // only the valid rows are staged; the K loop skips what would read the halo
    g.R = g.nv;
    g.fh = 0;
    g.hr = 0;
  }
  g.cbpt = p.Cin_p >> 4;
  g.KK1 = p.ntaps * g.cbpt;
  g.KK = g.KK1 + (p.x2 ? (p.Cin2_p >> 4) : 0);
  g.chunk = (g.KK + g.KS - 1) / g.KS;
  if (tl.bf) g.chunk = (g.KK / 2 + g.KS - 1) / g.KS * 2;      // bf16x3 consumes 16-channel units in (hi, lo) pairs
  // Candidate stride: where one 16-row MFMA tile spans several candidates the same rule needs the candidates' row
  // blocks on complementary bank quads: 16 candidates (position-major) == 8 (mod 16) floats, 2 candidates of 8 rows
  // == 0 (mod 64), 4 candidates of 4 rows == 32 (mod 64).
  auto cand_stride = [&](int base) {
    auto pad_to = [](int v, int mod, int want) { return v + (((want - v) % mod) + mod) % mod; };
    if (tl.pm) return pad_to(base, 16, 8);
    if (p.stride == 1 && p.Lout == 8) return pad_to(base, 64, 0);
    if (p.stride == 1 && p.Lout == 4) return pad_to(base, 64, 32);
    return base;
  };
  g.CS = cand_stride(g.R * g.ldA);
  g.CS2 = p.x2 ? cand_stride(p.Lout * g.ld2) : 0;
  int nA = g.S * g.CS, n2 = p.x2 ? g.S * g.CS2 : 0, nY = p.pro ? g.S * p.Lin * g.ldA : 0;
  // The channel-block-major loop (conv_w5_ok) is done with the slab before it turns to the second source, so the second
  // source is staged into the slab's space THEN (one exposed staging pass, two barriers) instead of next to it: 16
  // candidates of the 256-channel backward layers and 32 of the forward ones fit the 160 KB that way.
  static const bool late_on = !(getenv("TREBI_X2LATE") && atoi(getenv("TREBI_X2LATE")) == 0);
  // (only where the W5 kernel WILL be launched: it exists for vector-addressable sources only -- launch_conv_t -- so the
  // carve may not assume it for a source like the caller's x with a row stride of 23 floats)
  // (a second source LARGER than the slab -- the up path's 1x1 residual conv reads the 2C-channel concat -- takes the slab's
  // place all the same: the shared space is the larger of the two)
  static const bool late_big = !(getenv("TREBI_X2LATE_BIG") && atoi(getenv("TREBI_X2LATE_BIG")) == 0);
  g.x2late = (late_on && p.x2 && (n2 <= nA || late_big) && conv_w5_ok(p, tl) && conv_vec_ok(p)) ? 1 : 0;
  g.oSlab2 = g.x2late ? 0 : nA;
  if (g.x2late) { nA = nA > n2 ? nA : n2; n2 = 0; }
  g.oY = nA + n2;
  g.red1 = g.KS * g.TM * g.TNp;
  int red = g.red1 * (p.x2 ? 2 : 1), stage = nA + n2 + nY;
  int body = ((stage > red ? stage : red) + 3) & ~3;
  g.oRes = body;
  g.resg = (p.res && tl.resg) ? 1 : 0;
  g.oColp = g.oRes + ((p.res && !g.resg) ? g.TM * g.TNp : 0);
  g.oTb = g.oColp + 4 * g.TN;
  g.oPstat = g.oTb + (p.tbias ? g.S * g.TN : 0);
  g.total = g.oPstat + (p.pro ? 2 * g.S * GN_GROUPS : 0);
  g.c4n = p.Cin_p >> 2;
  g.c4n2 = p.x2 ? (p.Cin2_p >> 2) : 1;
  g.nValid4 = g.S * g.nv * g.c4n;
  g.nHalo4 = g.S * g.hr * g.c4n;
  g.nX24 = p.x2 ? g.S * p.Lout * g.c4n2 : 0;
  g.tn4 = g.TN / 4;
  g.nRes4 = g.TM * g.tn4;
  const int nteams16 = 4 * tl.NG * tl.KS;
  g.gpt = p.gn ? g.TN / p.gch : 1;
  g.gch4 = p.gn ? p.gch / 4 : 1;
  g.pgch4 = p.pro ? p.pro_gch / 4 : 1;
  g.cnt4 = p.gn ? p.Lout * g.gch4 : 1;
  g.pcnt4 = p.pro ? p.Lin * g.pgch4 : 1;
  g.icnt = 1.f / (float)(g.cnt4 * 4);
  g.picnt = 1.f / (float)(g.pcnt4 * 4);
  g.ts_epi = (g.S * g.gpt * 4 <= nteams16) ? 64 : 16;
  g.ts_pro = (g.S * GN_GROUPS * 4 <= nteams16) ? 64 : 16;
  const int ophase = p.osplit > 0 ? 2 : 1;
  (void)ophase;
  g.veco = ((p.ldo & 3) == 0) && ((p.N & 3) == 0) && ((((uintptr_t)p.out) & 15) == 0) &&
           (p.osplit <= 0 || (p.osplit & 3) == 0);
  g.vecr = p.res ? (((p.ldres & 3) == 0) && ((p.N & 3) == 0) && ((((uintptr_t)p.res) & 15) == 0)) : 0;
  g.mLout = magic_of(p.Lout); g.mTN4 = magic_of(g.tn4); g.mNv = magic_of(g.nv > 0 ? g.nv : 1);
  g.mHr = magic_of(g.hr > 0 ? g.hr : 1); g.mGch4 = magic_of(g.gch4); g.mPGch4 = magic_of(g.pgch4);
  g.mGpt = magic_of(g.gpt); g.mOs = magic_of(p.osplit > 0 ? p.osplit : 1); g.mC4 = magic_of(g.c4n);
  g.mC4b = magic_of(g.c4n2); g.mNG = magic_of(g.NG); g.mCbpt = magic_of(g.cbpt); g.mTN = magic_of(g.TN);
  return g;
}

size_t conv_lds_bytes(const ConvP& p, const Tile& tl) {
  return (size_t)conv_make_geo(p, tl).total * sizeof(float);
}

// WB = reduction steps whose weights a wave preloads into registers before anything else.
// The launch descriptor (this net's ConvP followed by the ConvGeo) is read ONCE from the kernel-argument
// segment with vector loads, one dword per lane, and kept in 4 VGPRs (see DV in device_util.h).
constexpr int CPW = (int)(sizeof(ConvP) / 4);
constexpr int CG0 = (int)offsetof(ConvLaunch, g);
constexpr int CVIEW_DW = CPW + (int)(sizeof(ConvGeo) / 4);
static_assert(CVIEW_DW <= 256 && sizeof(ConvP) % 4 == 0 && CG0 % 4 == 0, "launch descriptor view must fit 4 VGPRs");
#define CP(f) dfield<decltype(ConvP::f), (int)(offsetof(ConvP, f) / 4)>(dv)
#define CG(f) dfield<decltype(ConvGeo::f), CPW + (int)(offsetof(ConvGeo, f) / 4)>(dv)

// The fields the weight preload needs travel as separate kernel arguments in front of the launch descriptor and are
// PRELOADED into scalar registers by the command processor (-mllvm -amdgpu-kernarg-preload-count, see the
// Makefile): the wave requests its weights at once, while the descriptor (read from the kernel-argument segment,
// a cold ~0.9 us fetch after every kernel boundary) is still on its way.
struct ConvPre {
  const float* w;                   // this net's packed weights
  int Npad, TN, NG;
  unsigned mNG;
  int chunk, KK, KK1;
};

// BF: operands as bf16 hi/lo pairs (bf16x3, see device_util.h): the packed weights `pre.w` are then the bf16x3 copy
// (same addressing: a 16-byte unit per (step, lane group, column); unit 2k is the hi half, 2k+1 the lo half of the
// 32-channel step k), the staged activations are converted once on their way into LDS (a row keeps its byte size: hi
// plane then lo plane), and the K loop issues three 16x16x32 bf16 MFMAs per 32 channels.  Layers with the fused
// GroupNorm-backward prologue are not taken in this mode (the host keeps them on the fp32 pipe).
// W5 (position-major fp32 tiles only, an instantiation of its own so that no other K loop shares its registers): the
// channel-block-major reduction for 5-tap stride-1 "same" convolutions whose L positions are the wave's MI tiles.  Tile mi
// is output position mi of 16 candidates and input position q sits (q - mi) taps away, so per 16-channel block the MI
// input positions are read from LDS ONCE and feed every (tile, tap) pair that exists -- 14 MFMA groups from 4 reads at
// L = 4, where the tap-major loop reads one operand per pair.  The operand reads are what the large-batch K loop pays
// for: no branch inside a block, accumulators in place, the next block's five weight rows in flight.
template <int MI, int NI, bool VEC, bool PM, bool BF, bool W5 = false>
__device__ __forceinline__ void conv_body(const DV& dv, const ConvPre pre, float* smem, const int by, const int n0,
                                          const int tid) {
  constexpr int WB = (NI == 1) ? (MI <= 2 ? 24 : 16) : 8;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* pw = pre.w;
  const int Npad = pre.Npad, TN = pre.TN, NG = pre.NG, chunk = pre.chunk, KK = pre.KK, KK1 = pre.KK1;
  constexpr bool active = true;     // the launch has exactly 64*NG*KS threads
  STAMP(0);

  // ---- wave geometry + weight preload (registers) -------------------------------------------
  // v_mfma_f32_16x16x4_f32: lane l feeds A[row l&15][k l>>4] and B[k l>>4][col l&15].  The
  // reduction index is permuted so that lane-group g = l>>4 owns channels 16*kk+4g .. +3 of
  // step kk and the four MFMAs of the step consume one float4 of A (LDS) and one of B (global).
  const int g = lane >> 4, lr = lane & 15;
  const int ks = FastDiv(NG, pre.mNG).div(wave), ng = wave - ks * NG;
  const int k_begin = active ? ks * chunk : 0;
  const int k_end = active ? ((k_begin + chunk < KK) ? k_begin + chunk : KK) : 0;
  // uniform base (scalar registers) + one 32-bit per-lane offset: the loads need no per-lane 64-bit math
  // 24-bit multiply on purpose: for a 32-bit multiply-add the compiler emits v_mad_u64_u32 with an undefined high
  // addend, allocates that to the register the descriptor is being loaded into, and then waits for the descriptor
  // (vmcnt(0)) in front of the weight loads
  const unsigned wlane = (__umul24((unsigned)g, (unsigned)Npad) + (unsigned)(n0 + ng * NI * 16 + lr)) * 4u;
  const unsigned wstep = (unsigned)Npad * 16u;     // floats per kk step
  constexpr int WT = 5;                                  // taps of the W5 loop
  f32x4 wreg[W5 ? 2 * WT : WB][NI];                      // W5: two groups of WT weight rows
  // A reduction longer than one register block (large batches: no split-K) is software-pipelined in half
  // blocks: while one half's MFMAs run, the other half's registers are refilled for the next block.  Those
  // loads are unconditional -- steps past the layer re-read its last step -- so that the compiler's vmcnt
  // waits are exact counts instead of vmcnt(0).
  constexpr int HB = WB / 2;
  const bool multi = k_end - k_begin > WB;
  // The planning-batch shape (one 16x16 tile per wave, the whole chunk in registers): straight-line step sequences
  // entered by one branch per block of BS steps, so every A operand of a block is requested from LDS before its
  // first MFMA (a branch per step exposed the LDS latency at every step), and two accumulators per source, one for
  // even and one for odd slots (a single chain pays the 40-cycle dependent-MFMA latency instead of the 32-cycle issue).
  constexpr bool STR = MI == 1 && NI == 1 && !PM && !BF;
  constexpr int BS = 8;
  const int n1 = (k_end < KK1 ? k_end : KK1) > k_begin ? (k_end < KK1 ? k_end : KK1) - k_begin : 0;   // first-source steps
  const int n2 = k_end - k_begin - n1;                                                              // second-source steps
  const int k_last = KK - 1;
  auto wload = [&](int u0, int kbase) {
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      int k = kbase + u;
      k = k < k_last ? k : k_last;
      const float* wu = pw + (unsigned)k * wstep;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wreg[u0 + u][ni] = gld4(wu + wlane + ni * 64);
    }
  };
  // W5: weight GROUP G < cb5 = the five taps of channel block G; G >= cb5 = five consecutive steps of the second source
  // (clamped to the last step: the loads stay unconditional, so the vmcnt waits are exact counts).
  const int cb5 = (int)((unsigned)KK1 / 5u);
  auto wrow = [&](auto slot_tag, int G, int u) {
    constexpr int slot = decltype(slot_tag)::value;
    int k2 = KK1 + WT * (G - cb5) + u;
    k2 = k2 < k_last ? k2 : k_last;
    const int k = G < cb5 ? u * cb5 + G : k2;
    const float* wu = pw + (unsigned)k * wstep;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wreg[slot * WT + u][ni] = gld4(wu + wlane + ni * 64);
  };
  auto wgroup = [&](auto slot_tag, int G) {
    constexpr int slot = decltype(slot_tag)::value;
#pragma unroll
    for (int u = 0; u < WT; ++u) {
      int k2 = KK1 + WT * (G - cb5) + u;
      k2 = k2 < k_last ? k2 : k_last;
      const int k = G < cb5 ? u * cb5 + G : k2;
      const float* wu = pw + (unsigned)k * wstep;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wreg[slot * WT + u][ni] = gld4(wu + wlane + ni * 64);
    }
  };
  if constexpr (W5) {
    wgroup(std::integral_constant<int, 0>{}, 0);
  } else if (multi) {
    wload(0, k_begin);
    wload(HB, k_begin + HB);
  } else if constexpr (STR) {
    // slot u < n1: step k_begin + u of the first source; the LAST n2 slots: the second source's steps counted
    // from the end (slot WB-1-j <-> step k_end-1-j), so that both step sequences below index registers statically
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int kk = u < n1 ? k_begin + u : (u >= WB - n2 ? k_end - WB + u : -1);
      if (kk >= 0) wreg[u][0] = gld4(pw + (unsigned)kk * wstep + wlane);
    }
  } else {
    const float* wu = pw + (size_t)k_begin * wstep;
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      if (k_begin + u < k_end) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wreg[u][ni] = gld4(wu + wlane + ni * 64);
      }
      wu += wstep;
    }
  }
  STAMP(1);

  // ---- every other argument field used below, read once from the descriptor and pinned in scalar registers ----
  const float* px = CP(x);  float* pout = CP(out);
  int ldx = CP(ldx), Cin = CP(Cin), Lin = CP(Lin), Lout = CP(Lout), stride = CP(stride), B = CP(B), N = CP(N);
  int S = CG(S), KS = CG(KS), nthr = CG(nthr), TM = CG(TM), TNp = CG(TNp), ldA = CG(ldA), ld2 = CG(ld2);
  int cbpt = CG(cbpt), nv = CG(nv), fh = CG(fh), CS = CG(CS), CS2 = CG(CS2);
  PIN(px); PIN(pout); PIN(ldx); PIN(Cin); PIN(Lin); PIN(Lout); PIN(stride); PIN(B); PIN(N);
  PIN(S); PIN(KS); PIN(nthr); PIN(TM); PIN(TNp); PIN(ldA); PIN(ld2);
  PIN(cbpt); PIN(nv); PIN(fh); PIN(CS); PIN(CS2);
  const bool has2 = CP(x2) != nullptr, haspro = CP(pro) != 0, hasres = CP(res) != nullptr, hastb = CP(tbias) != nullptr,
             hasgn = CP(gn) != 0;
  const int t_uniform = CP(t_uniform);
  const long long* tptr = CP(t);
  const int b0 = by * S;
  float* slab = smem;
  float* slab2 = smem + CG(oSlab2);
  float* yslab = smem + CG(oY);
  float* restile = smem + CG(oRes);
  float* colp = smem + CG(oColp);          // [4][TN]: bias, gamma, beta, bias2
  float* tbrow = smem + CG(oTb);           // [S][TN]
  float* pstat = smem + CG(oPstat);        // [S*8][2] mean, rstd of the producing block
  const FastDiv dLout(Lout, CG(mLout)), dTN4(CG(tn4), CG(mTN4)), dTN(TN, CG(mTN));
  const int nvs = (B - b0 < S) ? B - b0 : S;       // candidates of this tile that exist

  // second source rows at the output positions -> slab2 (with the other streams below, or -- x2late -- after the first
  // source's reduction, into the space of the slab)
  const bool x2late = W5 && CG(x2late) != 0;
  auto stage_x2 = [&]() {
    const FastDiv dC4(CG(c4n2), CG(mC4b));
    const int c4n = CG(c4n2), total = CG(nX24), C2 = CP(Cin2);
    const float* p2 = CP(x2);
    const unsigned ldx2 = CP(ldx2);
    for (int base = tid; base < total; base += 2 * nthr) {
      f32x4 v[2];
      int so[2] = {-1, -1}, sc[2] = {0, 0};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int idx = base + u * nthr;
        if (idx < total) {
          int row = dC4.div(idx), c4 = idx - row * c4n;
          const int s2 = dLout.div(row);
          so[u] = s2 * CS2 + (row - s2 * Lout) * ld2 + 4 * c4;
          sc[u] = 4 * c4;
          int left = row < nvs * Lout ? C2 - 4 * c4 : 0;
          v[u] = load4<VEC>(p2 + ((unsigned)b0 * Lout + row) * ldx2 + 4 * c4, left);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (so[u] >= 0) {
          if constexpr (BF) store4_bf16x2(slab2 + so[u] - sc[u], sc[u], 4 * c4n, v[u]);
          else *(f32x4*)&slab2[so[u]] = v[u];
        }
    }
  };

  // ---- stage everything this workgroup reads from global: all loads first, then the LDS writes ----
  if (active) {
    float c0 = 0.f, c1 = 1.f, c2 = 0.f, c3 = 0.f;
    const bool colt = tid < TN;
    if (colt) {
      int n = n0 + tid;
      bool ok = n < N;
      if (ok && CP(bias)) c0 = gld(CP(bias) + n);
      if (ok && hasgn) { c1 = gld(CP(gamma) + n); c2 = gld(CP(beta) + n); }
      if (ok && CP(bias2)) c3 = gld(CP(bias2) + n);
    }
    auto tb_load = [&](int e) -> float {
      int s = dTN.div(e), c = e - s * TN;
      int n = n0 + c;
      if (s >= nvs || n >= N) return 0.f;
      long long tt = t_uniform >= 0 ? t_uniform : gldll(tptr + b0 + s);
      tt = tt < 0 ? 0 : (tt >= CP(T) ? CP(T) - 1 : tt);
      return gld(CP(tbias) + (unsigned)tt * (unsigned)N + n);
    };
    auto ps_load = [&](int i) -> float {
      return (i < nvs * GN_GROUPS * 2) ? gld(CP(pro_stats) + (unsigned)b0 * GN_GROUPS * 2 + i) : 0.f;
    };
    float tbv[2] = {0.f, 0.f}, psv = 0.f;
    if (hastb) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < S * TN) tbv[u] = tb_load(tid + u * nthr);
    }
    if (haspro && tid < S * GN_GROUPS * 2) psv = ps_load(tid);
    // identity residual tile
    f32x4 rv[2];
    int ro[2] = {-1, -1};
    const int nRes4 = CG(nRes4);
    auto res_item = [&](int e, int* o) -> f32x4 {
      int row = dTN4.div(e), c4 = e - row * CG(tn4);
      *o = row * TNp + 4 * c4;
      int left = N - n0 - 4 * c4;
      if (row >= nvs * Lout) left = 0;
      const float* sp = CP(res) + ((unsigned)b0 * Lout + row) * (unsigned)CP(ldres) + n0 + 4 * c4;
      return CG(vecr) ? load4<true>(sp, left) : load4<false>(sp, left);
    };
    const bool resl = hasres && !CG(resg);          // residual staged as an LDS tile
    if (resl) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * nthr < nRes4) rv[u] = res_item(tid + u * nthr, &ro[u]);
    }
    // primary source: the valid rows of the S candidates (the halo is zero-filled separately).  With a
    // prologue the source is dL/d(Mish output); it is multiplied by the saved mish'(z)*gamma on the way
    // in (the slab then holds dL/dxhat) and the saved xhat goes to yslab with the same indexing.
    {
      const FastDiv dC4(CG(c4n), CG(mC4)), dNv(nv, CG(mNv));
      const int c4n = CG(c4n), total = CG(nValid4), vlo = CG(vlo);
      const int Cin_p_ = cbpt * 16;
      const float* pg = CP(pro_g);
      const float* py = CP(pro_y);
      for (int base = tid; base < total; base += 4 * nthr) {
        f32x4 v[4], m[4], y[4];
        int so[4], yo[4], sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int idx = base + u * nthr;
          so[u] = -1;
          if (idx < total) {
            int rowv = dC4.div(idx), c4 = idx - rowv * c4n;
            int s = dNv.div(rowv), i = rowv - s * nv;
            so[u] = s * CS + (fh + i) * ldA + 4 * c4;
            sc[u] = 4 * c4;
            yo[u] = (s * Lin + i) * ldA + 4 * c4;
            unsigned grow = (unsigned)(b0 + s) * Lin + vlo + i;
            int left = s < nvs ? Cin - 4 * c4 : 0;
            v[u] = load4<VEC>(px + grow * (unsigned)ldx + 4 * c4, left);
            if (haspro) {
              m[u] = load4<true>(pg + grow * (unsigned)Cin + 4 * c4, left);
              y[u] = load4<true>(py + grow * (unsigned)Cin + 4 * c4, left);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (so[u] >= 0) {
            if (haspro) {
              *(f32x4*)&slab[so[u]] = v[u] * m[u];
              *(f32x4*)&yslab[yo[u]] = y[u];
            } else if constexpr (BF) {
              store4_bf16x2(slab + so[u] - sc[u], sc[u], Cin_p_, v[u]);
            } else {
              *(f32x4*)&slab[so[u]] = v[u];
            }
          }
      }
      const FastDiv dHr(CG(hr), CG(mHr));
      const int nh = CG(nHalo4), hr = CG(hr);
      for (int idx = tid; idx < nh; idx += nthr) {
        int h = dC4.div(idx), c4 = idx - h * c4n;
        int s = dHr.div(h), k = h - s * hr;
        int j = k < fh ? k : k + nv;
        *(f32x4*)&slab[s * CS + j * ldA + 4 * c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (has2 && !x2late) stage_x2();
    if (haspro) {
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
    if (resl) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (ro[u] >= 0) *(f32x4*)&restile[ro[u]] = rv[u];
      for (int e = tid + 2 * nthr; e < nRes4; e += nthr) {
        int o;
        f32x4 v = res_item(e, &o);
        *(f32x4*)&restile[o] = v;
      }
    }
  }
  STAMP(2);
  __syncthreads();
  STAMP(3);

  // Teams of 16 (or 64) lanes each own one (candidate, GroupNorm group): all passes over a group stay
  // inside the team (DPP reductions), so neither the prologue nor the epilogue needs a barrier between
  // its passes.  Every team works on float4s (groups are multiples of 4 channels).

  // ---- prologue: GroupNorm backward of the producing Conv1dBlock, in LDS ----------------------
  //   slab = dxhat = dA * mish'(z) * gamma (multiplied while staging), yslab = xhat (saved by the forward)
  //   dY = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat * xhat))
  if (haspro) {
    if (active) {
      const int gch = CP(pro_gch), cnt4 = CG(pcnt4), pg4 = CG(pgch4), ninst = S * GN_GROUPS;
      const int ts = CG(ts_pro);
      const int team = ts == 64 ? (tid >> 6) : (tid >> 4), tl = tid & (ts - 1),
                nteams = ts == 64 ? (nthr >> 6) : (nthr >> 4);
      const FastDiv dG(pg4, CG(mPGch4));
      const float icnt = CG(picnt);
      for (int inst = team; inst < ninst; inst += nteams) {
        int s = inst >> 3, gi = inst & 7;
        if (s >= nvs) continue;
        const float rstd = pstat[2 * inst + 1];
        const int ybase = s * Lin * ldA + gi * gch, dbase = s * CS + fh * ldA + gi * gch;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
        for (int e = tl; e < cnt4; e += ts) {
          int li = dG.div(e), o = li * ldA + 4 * (e - li * pg4);
          f32x4 d = *(const f32x4*)&slab[dbase + o];
          f32x4 x = *(const f32x4*)&yslab[ybase + o];
          s1 += hsum(d);
          s2 += hsum(d * x);
        }
        const float m1 = team_sum(s1, ts) * icnt, m2 = team_sum(s2, ts) * icnt;
#pragma unroll 2
        for (int e = tl; e < cnt4; e += ts) {
          int li = dG.div(e), o = li * ldA + 4 * (e - li * pg4);
          f32x4 d = *(const f32x4*)&slab[dbase + o];
          f32x4 x = *(const f32x4*)&yslab[ybase + o];
          *(f32x4*)&slab[dbase + o] = (d - m1 - x * m2) * rstd;
        }
      }
    }
    __syncthreads();
  }
  STAMP(4);

  // ---- implicit GEMM on the f32 matrix cores --------------------------------------------
  constexpr bool pm = PM;            // position-major row tiles (Tile::pm), its own instantiation
  const int fhk = CG(fhk);
  int rowA[MI], row2[MI], jlo[MI], plo[MI], psb[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    if (pm) {                        // tile mi = output position mi % Lout of candidates 16*(mi / Lout) ..+15
      const int sb = dLout.div(mi), lo = mi - sb * Lout, s = sb * 16 + lr;
      plo[mi] = lo; psb[mi] = sb;
      jlo[mi] = lo * stride - fhk;   // staged row of tap 0; tap tau reads row jlo + tau, valid in [0, nv)
      rowA[mi] = s * CS + jlo[mi] * ldA + 4 * g;
      row2[mi] = s * CS2 + lo * ld2 + 4 * g;
    } else {
      int r = mi * 16 + lr;
      int s = dLout.div(r), lo = r - s * Lout;
      bool ok = s < S;
      plo[mi] = 0; psb[mi] = 0; jlo[mi] = 0;
      rowA[mi] = ok ? s * CS + lo * stride * ldA + 4 * g : 4 * g;
      row2[mi] = ok ? s * CS2 + lo * ld2 + 4 * g : 4 * g;
    }
  }
  f32x4 acc[MI][NI], acc2[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const FastDiv dCb(cbpt, CG(mCbpt));
  auto kstep = [&](int u, int kk, int& tau, int& cb, int& aoff) {
    if (kk < k_end) {
      f32x4 a[MI];
      if (kk < KK1) {
        if constexpr (PM) {
          // a (tile, tap) pair whose input row is zero padding is skipped as a whole (wave-uniform)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            if ((unsigned)(jlo[mi] + tau) < (unsigned)nv) a[mi] = *(const f32x4*)&slab[rowA[mi] + aoff];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            if ((unsigned)(jlo[mi] + tau) < (unsigned)nv) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], wreg[u][ni][j], acc[mi][ni], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)&slab[rowA[mi] + aoff];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], wreg[u][ni][j], acc[mi][ni], 0, 0, 0);
        }
        aoff += 16;
        if (++cb == cbpt) { cb = 0; ++tau; aoff += ldA - cbpt * 16; }
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
  };
  // bf16x3: register slots u / u+1 hold the hi / lo weight halves of the 32-channel step (kk, kk+1); in float units a
  // row's hi plane advances 16 per step and the lo plane sits Cin_p/2 behind it (byte for byte the fp32 row).
  const int lo1 = cbpt * 8, lo2 = CP(Cin2_p) >> 1;
  auto mma3 = [&](f32x4(&ac)[MI][NI], const f32x4(&ah)[MI], const f32x4(&al)[MI], const bool(&on)[MI], int u) {
#pragma unroll
    for (int term = 0; term < 3; ++term)       // hi*hi, hi*lo, lo*hi; tile-major inside a term: no back-to-back dependency
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        if (on[mi]) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            ac[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(term == 2 ? al[mi] : ah[mi]),
                                                                 as_bf16x8(wreg[term == 1 ? u + 1 : u][ni]), ac[mi][ni],
                                                                 0, 0, 0);
        }
  };
  auto kstep2 = [&](int u, int kk, int& tau, int& cb, int& aoff) {
    if (kk < k_end) {
      f32x4 ah[MI], al[MI];
      bool on[MI];
      if (kk < KK1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          on[mi] = !PM || (unsigned)(jlo[mi] + tau) < (unsigned)nv;       // zero padding is skipped as a whole
          if (on[mi]) {
            ah[mi] = *(const f32x4*)&slab[rowA[mi] + aoff];
            al[mi] = *(const f32x4*)&slab[rowA[mi] + aoff + lo1];
          }
        }
        mma3(acc, ah, al, on, u);
        aoff += 16;
        cb += 2;
        if (cb == cbpt) { cb = 0; ++tau; aoff += ldA - cbpt * 8; }
      } else {
        const int a2 = ((kk - KK1) >> 1) * 16;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          on[mi] = true;
          ah[mi] = *(const f32x4*)&slab2[row2[mi] + a2];
          al[mi] = *(const f32x4*)&slab2[row2[mi] + a2 + lo2];
        }
        mma3(acc2, ah, al, on, u);
      }
    }
  };
  if constexpr (W5) {
    const int K2 = KK - KK1, n2g = (K2 + WT - 1) / WT;
    int fb[MI];
#pragma unroll
    for (int q = 0; q < MI; ++q) fb[q] = rowA[q] + 2 * ldA;          // rowA[q] is tap 0 of tile q: input position q - 2
    // NB candidate blocks of L = MI / NB positions each: tile mi = position mi % L of block mi / L
    auto first = [&](auto slot_tag, auto nb_tag, int G) {
      constexpr int slot = decltype(slot_tag)::value;
      constexpr int LL = MI / decltype(nb_tag)::value;
      f32x4 f[MI];
#pragma unroll
      for (int q = 0; q < MI; ++q) f[q] = *(const f32x4*)&slab[fb[q] + G * 16];
      // The next block's weight rows, each requested in front of its own tap's MFMAs: a whole block of lead time for every row
      // and the wave's requests spread over the block (4-8 % faster than one burst in front of the block).  MFMAs and
      // global loads may not cross the fences (the scheduler otherwise sinks the requests to their use, one block on).
#pragma unroll
      for (int tau = 0; tau < WT; ++tau) {
        wrow(std::integral_constant<int, slot ^ 1>{}, G + 1, tau);
        __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int q = mi % LL + tau - 2;                          // input position within the tile's block
            if (q >= 0 && q < LL) {
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[q >= 0 && q < LL ? mi - mi % LL + q : 0][j],
                                                                   wreg[slot * WT + tau][ni][j], acc[mi][ni], 0, 0, 0);
            }
          }
      }
    };
    auto second = [&](auto slot_tag, int Gs) {
      constexpr int slot = decltype(slot_tag)::value;
      wgroup(std::integral_constant<int, slot ^ 1>{}, cb5 + Gs + 1);
      __builtin_amdgcn_sched_barrier(0x386);
#pragma unroll
      for (int u = 0; u < WT; ++u) {
        const int k2 = WT * Gs + u;
        if (k2 < K2) {
          f32x4 a[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *(const f32x4*)&slab2[row2[mi] + k2 * 16];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc2[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], wreg[slot * WT + u][ni][j], acc2[mi][ni], 0, 0, 0);
        }
      }
    };
    if (Lout == MI) {
      for (int G = 0; G < cb5; G += 2) {                             // cb5 is even (launch_conv_t)
        first(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, G);
        first(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, G + 1);
      }
    } else if constexpr (MI >= 4) {                                  // two candidate blocks per row tile set
      for (int G = 0; G < cb5; G += 2) {
        first(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, G);
        first(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, G + 1);
      }
    }
    if (x2late && n2g > 0) {
      __syncthreads();          // every wave is done with the first source's slab
      stage_x2();
      __syncthreads();
    }
    for (int Gs = 0; Gs < n2g; Gs += 2) {
      second(std::integral_constant<int, 0>{}, Gs);
      if (Gs + 1 < n2g) second(std::integral_constant<int, 1>{}, Gs + 1);
    }
  } else if constexpr (BF) {
    if (multi) {
      for (int base = k_begin; base < k_end; base += WB) {
        int tau = dCb.div(base), cb = base - tau * cbpt;
        int aoff = tau * ldA + cb * 8;
#pragma unroll
        for (int u = 0; u < HB; u += 2) kstep2(u, base + u, tau, cb, aoff);
        wload(0, base + WB);
#pragma unroll
        for (int u = HB; u < WB; u += 2) kstep2(u, base + u, tau, cb, aoff);
        wload(HB, base + WB + HB);
      }
    } else {
      int tau = dCb.div(k_begin), cb = k_begin - tau * cbpt;
      int aoff = tau * ldA + cb * 8;
#pragma unroll
      for (int u = 0; u < WB; u += 2) kstep2(u, k_begin + u, tau, cb, aoff);
    }
  } else if (multi) {
    for (int base = k_begin; base < k_end; base += WB) {
      int tau = dCb.div(base), cb = base - tau * cbpt;
      int aoff = tau * ldA + cb * 16;
#pragma unroll
      for (int u = 0; u < HB; ++u) kstep(u, base + u, tau, cb, aoff);
      wload(0, base + WB);
#pragma unroll
      for (int u = HB; u < WB; ++u) kstep(u, base + u, tau, cb, aoff);
      wload(HB, base + WB + HB);
    }
  } else if constexpr (STR) {
    f32x4 accb = {0.f, 0.f, 0.f, 0.f}, acc2b = {0.f, 0.f, 0.f, 0.f};
#define TREBI_MF(u, SLOT, EVEN, ODD)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
    if ((u) & 1) ODD = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], wreg[SLOT][0][j], ODD, 0, 0, 0);       \
    else EVEN = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], wreg[SLOT][0][j], EVEN, 0, 0, 0);             \
  }
#define TREBI_SEQ(n, SLOT, EVEN, ODD)                                       \
  switch (n) {                                                              \
    default: TREBI_MF(7, SLOT(7), EVEN, ODD) [[fallthrough]];               \
    case 7: TREBI_MF(6, SLOT(6), EVEN, ODD) [[fallthrough]];                \
    case 6: TREBI_MF(5, SLOT(5), EVEN, ODD) [[fallthrough]];                \
    case 5: TREBI_MF(4, SLOT(4), EVEN, ODD) [[fallthrough]];                \
    case 4: TREBI_MF(3, SLOT(3), EVEN, ODD) [[fallthrough]];                \
    case 3: TREBI_MF(2, SLOT(2), EVEN, ODD) [[fallthrough]];                \
    case 2: TREBI_MF(1, SLOT(1), EVEN, ODD) [[fallthrough]];                \
    case 1: TREBI_MF(0, SLOT(0), EVEN, ODD)                                 \
  }
#pragma unroll
    for (int blk = 0; blk < WB / BS; ++blk) {           // first source: slots blk*BS + u
      const int nb = n1 - blk * BS;
      if (nb > 0) {
        const int kb = k_begin + blk * BS;
        int tau = dCb.div(kb), cb = kb - tau * cbpt;
        int aoff = rowA[0] + tau * ldA + cb * 16;
        const int aoff0 = aoff;
        f32x4 a[BS];
#pragma unroll
        for (int u = 0; u < BS; ++u) {
          a[u] = *(const f32x4*)&slab[u < nb ? aoff : aoff0];     // unused slots: any valid address
          aoff += 16;
          if (++cb == cbpt) { cb = 0; aoff += ldA - cbpt * 16; }
        }
#define TREBI_SLOT1(u) (blk * BS + (u))
        TREBI_SEQ(nb, TREBI_SLOT1, acc[0][0], accb)
#undef TREBI_SLOT1
      }
    }
#pragma unroll
    for (int blk = 0; blk < WB / BS; ++blk) {           // second source, from the end: slot WB-1-(blk*BS+u)
      const int nb = n2 - blk * BS;
      if (nb > 0) {
        const int a2 = row2[0] + (k_end - 1 - blk * BS - KK1) * 16;
        f32x4 a[BS];
#pragma unroll
        for (int u = 0; u < BS; ++u) a[u] = *(const f32x4*)&slab2[u < nb ? a2 - 16 * u : a2];
#define TREBI_SLOT2(u) (WB - 1 - blk * BS - (u))
        TREBI_SEQ(nb, TREBI_SLOT2, acc2[0][0], acc2b)
#undef TREBI_SLOT2
      }
    }
#undef TREBI_SEQ
#undef TREBI_MF
    acc[0][0] += accb;
    acc2[0][0] += acc2b;
  } else {
    int tau = dCb.div(k_begin), cb = k_begin - tau * cbpt;
    int aoff = tau * ldA + cb * 16;
#pragma unroll
    for (int u = 0; u < WB; ++u) kstep(u, k_begin + u, tau, cb, aoff);
  }
  STAMP(5);
  __syncthreads();     // every wave is done with the staged activations; LDS becomes the reduce buffer

  // ---- split-K partials -> LDS.  C/D layout: col = lane&15, row = 4*(lane>>4) + reg -------
  float* red = smem;
  float* red2 = smem + CG(red1);
  const int kstride = TM * TNp;
  if (active) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = pm ? (psb[mi] * 16 + g * 4 + q) * Lout + plo[mi] : mi * 16 + g * 4 + q;
          int o = ks * kstride + row * TNp + (ng * NI + ni) * 16 + lr;
          red[o] = acc[mi][ni][q];
          if (has2) red2[o] = acc2[mi][ni][q];
        }
  }
  __syncthreads();
  STAMP(6);
  if (!active) return;

  const FastDiv dOs(CP(osplit) > 0 ? CP(osplit) : 1, CG(mOs));
  const int osplit = CP(osplit);
  const unsigned ldo = CP(ldo);
  const bool veco = CG(veco) != 0, resg = CG(resg) != 0;
  // finish 4 adjacent output columns: time bias, second-source sum, residual, store
  auto finish = [&](f32x4 v, int s, int row, int col) {
    const int o = row * TNp + col;
    if (hastb) v += *(const f32x4*)&tbrow[s * TN + col];
    if (has2) {
      f32x4 v2 = *(const f32x4*)&colp[3 * TN + col];
      for (int k = 0; k < KS; ++k) v2 += *(const f32x4*)&red2[k * kstride + o];
      v += v2;
    }
    const int n = n0 + col;
    unsigned orow = (unsigned)b0 * Lout + row;
    if (hasres) {
      if (resg) {
        const float* sp = CP(res) + orow * (unsigned)CP(ldres) + n;
        v += CG(vecr) ? load4<true>(sp, N - n) : load4<false>(sp, N - n);
      } else {
        v += *(const f32x4*)&restile[o];
      }
    }
    float* dst;
    if (osplit > 0) {
      int ph = dOs.div(n);
      dst = pout + (orow * 2 + ph) * ldo + (n - ph * osplit);
    } else {
      dst = pout + orow * ldo + n;
    }
    if (veco) {
      if (n < N) gst4(dst, v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < N) gst(dst + e, v[e]);
    }
  };

  if (hasgn) {
    // ---- split-K reduce + bias, GroupNorm statistics (two passes), normalise, Mish: per team --------
    const int gch = CP(gch), g4 = CG(gch4), cnt4 = CG(cnt4), gpt = CG(gpt), ninst = S * gpt;
    const int ts = CG(ts_epi);
    const int team = ts == 64 ? (tid >> 6) : (tid >> 4), tl = tid & (ts - 1),
              nteams = ts == 64 ? (nthr >> 6) : (nthr >> 4);
    const FastDiv dG(g4, CG(mGch4)), dGpt(gpt, CG(mGpt));
    const float icnt = CG(icnt);
    float* sy = CP(save_y);
    float* sg = CP(save_g);
    // Two (candidate, group) instances per trip: the three passes of one instance are a dependent chain (LDS read ->
    // cross-lane sum -> LDS read ...) with two float4 per lane at most, so a second, independent instance fills the gaps.
    // A slot without an instance repeats the first one (same values to the same addresses) and skips the global stores.
    // (Small batches have one instance per team: they take the one-slot form, a repeated slot would double their work.)
    const bool want_dm = sg != nullptr;
    auto gn_epilogue = [&](auto iu_tag) {
    constexpr int IU = decltype(iu_tag)::value;
    for (int inst0 = team; inst0 < ninst; inst0 += IU * nteams) {
      int s_[IU], gi_[IU], tbase_[IU];
      bool on_[IU];
      float mean_[IU], rstd_[IU];
#pragma unroll
      for (int q = 0; q < IU; ++q) {
        int inst = inst0 + q * nteams;
        on_[q] = inst < ninst;
        inst = on_[q] ? inst : inst0;
        s_[q] = dGpt.div(inst);
        gi_[q] = inst - s_[q] * gpt;
        on_[q] = on_[q] && s_[q] < nvs;
        tbase_[q] = s_[q] * Lout * TNp + gi_[q] * gch;
      }
      float sum_[IU] = {};
#pragma unroll 2
      for (int e = tl; e < cnt4; e += ts) {
        const int lo = dG.div(e), c = 4 * (e - lo * g4);
#pragma unroll
        for (int q = 0; q < IU; ++q) {
          const int o = tbase_[q] + lo * TNp + c;
          f32x4 v = *(const f32x4*)&colp[gi_[q] * gch + c];
          for (int k = 0; k < KS; ++k) v += *(const f32x4*)&red[k * kstride + o];
          if (q == 0 || on_[q]) *(f32x4*)&red[o] = v;
          sum_[q] += hsum(v);
        }
      }
#pragma unroll
      for (int q = 0; q < IU; ++q) mean_[q] = team_sum(sum_[q], ts) * icnt;
      float sq_[IU] = {};
#pragma unroll 2
      for (int e = tl; e < cnt4; e += ts) {
        const int lo = dG.div(e), c = 4 * (e - lo * g4);
#pragma unroll
        for (int q = 0; q < IU; ++q) {
          f32x4 d = *(const f32x4*)&red[tbase_[q] + lo * TNp + c] - mean_[q];
          sq_[q] += hsum(d * d);
        }
      }
#pragma unroll
      for (int q = 0; q < IU; ++q) rstd_[q] = __frsqrt_rn(team_sum(sq_[q], ts) * icnt + GN_EPS);
      if (tl == 0 && CP(save_stats)) {
#pragma unroll
        for (int q = 0; q < IU; ++q)
          if (on_[q]) {
            float* o = CP(save_stats) + ((unsigned)(b0 + s_[q]) * GN_GROUPS + (n0 / gch + gi_[q])) * 2;
            gst(o, mean_[q]);
            gst(o + 1, rstd_[q]);
          }
      }
      for (int e = tl; e < cnt4; e += ts) {
        const int lo = dG.div(e), c = 4 * (e - lo * g4);
#pragma unroll
        for (int q = 0; q < IU; ++q) {
          const int col = gi_[q] * gch + c, row = s_[q] * Lout + lo;
          f32x4 xh = (*(const f32x4*)&red[row * TNp + col] - mean_[q]) * rstd_[q];
          f32x4 ga = *(const f32x4*)&colp[TN + col];
          f32x4 z = xh * ga + *(const f32x4*)&colp[2 * TN + col];
          f32x4 m, dm;
          if (want_dm) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float a, b;
              mish_both(z[j], &a, &b);
              m[j] = a;
              dm[j] = b;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = mish_f(z[j]);
          }
          if (on_[q]) {
            if (sy) {
              unsigned so = ((unsigned)b0 * Lout + row) * (unsigned)N + n0 + col;
              gst4(sy + so, xh);
              if (want_dm) gst4(sg + so, dm * ga);          // training keeps xhat only (train.hip recomputes mish')
            }
            finish(m, s_[q], row, col);
          }
        }
      }
    }
    };
    if (ninst > nteams) gn_epilogue(std::integral_constant<int, 2>{});
    else gn_epilogue(std::integral_constant<int, 1>{});
  } else {
    const int nitems = CG(nRes4);            // TM * TN / 4
    for (int e = tid; e < nitems; e += nthr) {
      int row = dTN4.div(e), col = 4 * (e - row * CG(tn4));
      if (row >= nvs * Lout || n0 + col >= N) continue;
      int s = dLout.div(row);
      f32x4 v = *(const f32x4*)&colp[col];
      for (int k = 0; k < KS; ++k) v += *(const f32x4*)&red[k * kstride + row * TNp + col];
      finish(v, s, row, col);
    }
  }
  STAMP(9);
}

constexpr int CPRE_DW = 12;          // dwords of kernel arguments in front of the descriptor (11 used + alignment)
template <int MI, int NI, bool VEC, bool PM, bool BF, bool W5 = false>
__global__ __launch_bounds__(512) void conv_gemm_kernel(const float* w0, const float* w1, int Npad, int TN, int NG,
                                                        unsigned mNG, int chunk, int KK, int KK1, ConvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef const __attribute__((address_space(4))) unsigned* kargp;
  kargp ka = (kargp)__builtin_amdgcn_kernarg_segment_ptr() + CPRE_DW;
  const int lane = threadIdx.x & 63, z = blockIdx.z;
  DV dv;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane + 64 * k;
    const int src = j < CPW ? z * CPW + j : CG0 / 4 + (j - CPW);
    dv.v[k] = j < CVIEW_DW ? ka[src] : 0u;
  }
  const ConvPre pre{z ? w1 : w0, Npad, TN, NG, mNG, chunk, KK, KK1};
  // blockIdx.x = column tile, blockIdx.y = candidate group: workgroups are dealt to the 8 XCDs round-robin by
  // linear id, so the workgroups that share a weight slice land on the same XCD (one L2 fetches the slice
  // instead of all eight; the activations are tiny)
  conv_body<MI, NI, VEC, PM, BF, W5>(dv, pre, smem, blockIdx.y, blockIdx.x * TN, threadIdx.x);
}

template <int MI, int NI, bool PM>
static hipError_t launch_conv_t(const ConvLaunch& L, int nz, const Tile& tl, hipStream_t st) {
  const ConvP& p = L.p[0];
  ConvLaunch LL = L;
  LL.g = conv_make_geo(p, tl);
  size_t lds = (size_t)LL.g.total * sizeof(float);
  dim3 grid(p.Npad / (16 * tl.NI * tl.NG), (p.B + tl.S - 1) / tl.S, nz);
  dim3 block(64 * tl.NG * tl.KS);
  bool vec = conv_vec_ok(L.p[0]) && (nz < 2 || conv_vec_ok(L.p[1]));
  const ConvGeo& g = LL.g;
  const float* w0 = L.p[0].w;
  const float* w1 = L.p[nz > 1 ? 1 : 0].w;
  static_assert(offsetof(ConvLaunch, p) == 0 && alignof(ConvLaunch) == 8, "descriptor offset in the kernel arguments");
  if (g.x2late && (tl.bf || !vec)) return hipErrorInvalidValue;   // only the W5 kernel stages the second source late
  if (tl.bf) {
    // bf16x3: vector-addressable sources, whole 32-channel steps per tap and per wave, no fused backward prologue
    if (!vec || p.pro || (p.Cin_p & 31) || (p.x2 && (p.Cin2_p & 31)) || (g.chunk & 1) || (g.KK & 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, true, PM, true>), grid, block, lds, st, w0, w1, p.Npad, g.TN, g.NG, g.mNG,
                       g.chunk, g.KK, g.KK1, LL);
  } else if (vec) {
    // the channel-block-major loop (conv_body, W5): 5-tap stride-1 "same" convolution whose positions are the wave's MI tiles,
    // the whole reduction in every wave, an even number of channel blocks
    bool w5 = false;
    if constexpr (PM)
      w5 = conv_w5_ok(p, tl) && g.fhk == 2 && g.nv == p.Lout && g.KK1 == 5 * g.cbpt;
    if (!w5 && g.x2late) return hipErrorInvalidValue;      // the LDS carve assumes the W5 kernel
    if (w5) {
      if constexpr (PM)
        hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, true, PM, false, true>), grid, block, lds, st, w0, w1, p.Npad, g.TN, g.NG,
                           g.mNG, g.chunk, g.KK, g.KK1, LL);
    } else
      hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, true, PM, false>), grid, block, lds, st, w0, w1, p.Npad, g.TN, g.NG, g.mNG,
                         g.chunk, g.KK, g.KK1, LL);
  }
  else
    hipLaunchKernelGGL((conv_gemm_kernel<MI, NI, false, PM, false>), grid, block, lds, st, w0, w1, p.Npad, g.TN, g.NG, g.mNG,
                       g.chunk, g.KK, g.KK1, LL);
  return hipGetLastError();
}

// the shapes the kernel is instantiated for: every (MI, NI) row-major; position-major where 16*MI == S*Lout can hold
#define TREBI_TILES(X) X(1, 1, false) X(2, 1, false) X(4, 1, false) X(8, 1, false) X(1, 2, false) X(2, 2, false) \
  X(4, 2, false) X(2, 1, true) X(4, 1, true) X(8, 1, true) X(2, 2, true) X(4, 2, true)

// Dynamic LDS above 64 KiB has to be opted into per kernel (once per device, at context creation).
hipError_t conv_init_device() {
  hipError_t e = hipSuccess;
#define TREBI_ATTR(mi, ni, pm)                                                                 \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni, true, pm, false>,          \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni, true, pm, true>,           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni, false, pm, false>,         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  TREBI_TILES(TREBI_ATTR)
#undef TREBI_ATTR
#define TREBI_ATTR5(mi, ni)                                                                    \
  if (e == hipSuccess)                                                                       \
    e = hipFuncSetAttribute((const void*)conv_gemm_kernel<mi, ni, true, true, false, true>,  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  TREBI_ATTR5(2, 1) TREBI_ATTR5(4, 1) TREBI_ATTR5(8, 1) TREBI_ATTR5(2, 2) TREBI_ATTR5(4, 2)
#undef TREBI_ATTR5
  return e;
}

hipError_t launch_conv(const ConvLaunch& L, int nz, const Tile& tl, hipStream_t st) {
  if (!tl.bf && narrow_ok(L, nz)) return launch_narrow(L, nz, st);
#define TREBI_CASE(mi, ni, posmajor) \
  if (tl.MI == mi && tl.NI == ni && (tl.pm != 0) == posmajor) return launch_conv_t<mi, ni, posmajor>(L, nz, tl, st);
  TREBI_TILES(TREBI_CASE)
#undef TREBI_CASE
  return hipErrorInvalidValue;
}

}  // namespace trebi
