// Internal declarations shared by the kernel file and the host-side planner.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trebi {

constexpr int GN_GROUPS = 8;       // helpers.py:51
constexpr float GN_EPS = 1e-5f;    // torch.nn.GroupNorm default

// One implicit-GEMM convolution over channels-last activations:
//   out[b, lo, n] = epi( sum_{tau, ci} x[b, lo*stride + tau - pad, ci] * W[tau][ci][n] + bias[n] )
//                   + sum_{ci} x2[b, lo, ci] * W2[ci][n] + bias2[n]  + res[b, lo, n]
// epi = identity, or GroupNorm(8)+Mish (+ per-timestep bias).  With `pro` set the primary
// source is first turned from dL/d(Mish output) into dL/d(conv output) of the *previous*
// Conv1dBlock (GroupNorm+Mish backward) while it is staged in LDS.
struct ConvP {
  // primary source
  const float* x;
  int ldx, Cin, Cin_p;             // Cin_p = Cin rounded up to 16
  int Lin, Lout, stride, pad;
  int tap_lo, ntaps;               // structurally non-zero taps: tau = tap_lo .. tap_lo+ntaps-1
  // optional second source, a 1x1 convolution read at the output positions
  const float* x2;
  int ldx2, Cin2, Cin2_p;
  // weights, packed [(ntaps*Cin_p + Cin2_p)/16][4][Npad][4]
  const float* w;
  const float* bias;               // [N] or null
  const float* bias2;              // [N] or null
  int N, Npad;
  // destination
  float* out;
  int ldo;
  int osplit;                      // >0: column n -> position 2*lo + n/osplit, channel n%osplit
  // epilogue
  int gn, gch;
  const float* gamma;
  const float* beta;
  float* save_y;                   // [B*Lout, N] normalised conv output xhat (for the backward)
  float* save_g;                   // [B*Lout, N] mish'(gamma*xhat+beta)*gamma (for the backward)
  float* save_stats;               // [B, 8, 2] (mean, rstd)
  const float* tbias;              // [T, N] or null
  const long long* t;              // [B]
  int t_uniform;                   // >= 0: every candidate is at this timestep (skips the t[b] load)
  int T;
  const float* res;                // identity residual or null
  int ldres;
  // prologue: GroupNorm+Mish backward applied to the primary source
  int pro, pro_gch;
  const float* pro_y;              // [B*Lin, Cin] xhat saved by the forward
  const float* pro_g;              // [B*Lin, Cin] mish'(z)*gamma saved by the forward
  const float* pro_stats;          // [B, 8, 2]
  int B;
};

// Everything about one launch that does not depend on the net (tile geometry, the LDS carve, loop
// bounds and the magic multipliers n / d == umulhi(n, m), m = ceil(2^32/d), m == 0 for d == 1), computed
// once on the host so that no workgroup spends instructions on derived integers or divisions.
struct ConvGeo {
  int S, NG, KS, nthr;                            // nthr = 64*NG*KS threads work in this stage
  int TM, TN, TNp;
  int R, ldA, ld2, cbpt, KK1, KK, chunk;
  int CS, CS2;                                    // candidate stride of slab / slab2 in floats (padded, see conv_make_geo)
  int nv, vlo, fh, hr;                            // valid input rows [vlo, vlo+nv) per candidate, front / total halo rows
  int pm, fhk;                                    // position-major tiles (see Tile::pm): flag, front halo of the K loop
  int resg;                                       // identity residual read from global memory in the epilogue (no LDS tile)
  int x2late;                                     // second source staged INTO the slab after the first source's reduction (W5)
  // LDS carve, float offsets:  [slab | slab2 | yslab] aliased by [red | red2] ; res | colp | tbrow | pstat
  int oSlab2, oY, red1, oRes, oColp, oTb, oPstat, total;
  int c4n, c4n2, nValid4, nHalo4, nX24, nRes4, tn4;   // float4 counts of the staged streams
  int gpt, cnt4, pcnt4, gch4, pgch4, ts_epi, ts_pro;  // GroupNorm teams (float4 granularity)
  int veco, vecr;                                 // output / identity residual are float4-addressable
  float icnt, picnt;                              // 1 / elements of a GroupNorm instance (epilogue, prologue): no device division
  unsigned mLout, mTN4, mNv, mHr, mGch4, mPGch4, mGpt, mOs, mC4, mC4b, mNG, mCbpt, mTN;
};

struct ConvLaunch {
  ConvP p[2];                      // blockIdx.z selects (reward / cost net run in one launch)
  ConvGeo g;                       // both nets of a launch have identical shapes
};

// One stage of a resident chain: consecutive layers whose workgroups own whole candidates AND all output
// columns run in ONE launch; a layer's output stays in LDS for the layers that consume it (it is also
// written to global memory when something outside the chain reads it).  Resident tensors use one canonical
// LDS layout: per candidate (L + 4) rows of ld = C + 8 floats, two zero halo rows in front and behind.
// All offsets are in floats from the start of dynamic LDS.  The table lives in device memory.
struct RStage {
  ConvP p[2];
  ConvGeo g;                       // tile geometry (the LDS carve in g is not used by the chain)
  int MI, NI, vec;                 // tile variant; global sources are float4-addressable
  // primary source: region, rows per candidate, row read by tap 0 of output 0, row of input position 0
  int srcRes, oSrc, Rb, a0, pr0;
  int x2Res, o2, Rb2, h2;          // second (1x1) source, read at row s*Rb2 + h2 + lo
  int resRes, oR, RbR, hR, ldR;    // identity residual, read at row s*RbR + hR + lo
  int dstRes, oD, RbD, ldD, Ld;    // destination region (canonical layout), Ld positions per candidate
  int oRed, oRed2, oColp, oTb;     // split-K partials, per-column parameters, time-bias rows
  int drain;                       // a global source was written by an earlier stage: drain stores first
  unsigned mLd4;                   // magic multiplier of ldD / 4
  int proRes, oPro;                // backward stage whose incoming gradient is resident (read-only) at oPro
  int direct;                      // no split-K: a wave owns a 16-row x (16*NI)-column tile and finishes it in registers
  // The value head (temporal.py:237-238) as a stage of a direct chain: p.x / the resident source = the flattened mid_down2
  // output [fc = p.Cin], p.w = W1 [hd = p.N][fc], p.tbias = the head's time table, p.gamma = w2, p.bias2 = b2 (replicated), p.out = y [B],
  // p.save_y = dL/dv [B, fc] (global copy, or null); the destination region takes dL/dv for the backward's first stage.
  int head;
  int oHead;                       // LDS scratch of the head stage: W1 [hd][fc + 4], dh [S][hd], v [S][fc]
};

struct Tile {
  int MI, NI;                      // 16-row / 16-column sub-tiles per wave
  int NG, KS;                      // waves = NG (column groups) x KS (split of the reduction)
  int S;                           // whole samples per workgroup
  // Position-major rows: MFMA row tile mi holds output position mi % Lout of 16 candidates (S a multiple of
  // 16, 16*MI == S*Lout), so a (tile, tap) pair that would only read zero padding is skipped as a whole and the
  // halo rows are never staged.  At L = 4 that is 14 of 20 (position, tap) pairs and half the LDS slab.
  int pm = 0;
  // Identity residual read straight from global memory in the epilogue instead of being staged as an LDS tile:
  // costs a memory round trip there, frees 16*MI x (TN+4) floats -- what lets 16 candidates fit the backward layers.
  int resg = 0;
  // bf16x3 operands (conv.hip, device_util.h): the launch's weight pointers are the bf16x3 copies
  int bf = 0;
};

size_t conv_lds_bytes(const ConvP& p, const Tile& tl);
hipError_t conv_init_device();
hipError_t launch_conv(const ConvLaunch& L, int nz, const Tile& tl, hipStream_t st);
// narrow.hip: the 32 / 64-channel Conv1dBlocks at large batches (persistent workgroups, register-resident weights)
bool narrow_ok(const ConvLaunch& L, int nz);
hipError_t launch_narrow(const ConvLaunch& L, int nz, hipStream_t st);
ConvGeo conv_make_geo(const ConvP& p, const Tile& tl);
bool conv_w5_ok(const ConvP& p, const Tile& tl);     // the layer / tile pairs that take the channel-block-major K loop
bool conv_vec_ok(const ConvP& p);
bool chain_variant_ok(int MI, int NI);
hipError_t chain_init_device();
// cpar: per-column parameter table of a direct chain, [nz][parT][nstages][5][parN] floats (bias, gamma, beta, time-bias row,
// bias2; defaults 0,1,0,0,0 past N), filled by launch_chain_params; oTab: float offset of its LDS copy.
hipError_t launch_chain(const RStage* dev_stages, int nstages, int direct, int nz, int nblocks, int ny, size_t lds_bytes,
                        int t_uniform, const long long* t, const float* cpar, int parN, int parT, int oTab, int has_head,
                        int feat /* bit 0: a stage has a GroupNorm-backward prologue, bit 1: a stage saves GroupNorm operands */,
                        hipStream_t st);
hipError_t launch_chain_params(const ConvP& p, float* dst /* + ((z*T + t)*nstages + stage)*5*parN */, int parN, int T,
                               size_t t_stride, hipStream_t st);

// Value head (temporal.py:237-238) forward + the seed of the backward.
struct HeadP {
  const float* v;                  // [B, fc] flattened mid_down2 output
  const float* w1;                 // [hd, fc] (the time-embedding columns are folded into tb)
  const float* tb;                 // [T, hd]  W1[:, fc:] @ temb(t) + b1
  const float* w2;                 // [hd]
  float b2;
  const long long* t;
  int T, fc, hd, B;
  float* y;                        // [B]
  float* dv;                       // [B, fc] or null
};
struct HeadLaunch { HeadP p[2]; };
hipError_t launch_head(const HeadLaunch& L, int nz, hipStream_t st);

// Time-embedding tables (computed once per network at load).
hipError_t launch_time_embed(const float* w1, const float* b1, const float* w2, const float* b2,
                             int dim, int T, float* temb /*[T,dim]*/, hipStream_t st);
hipError_t launch_time_linear(const float* w, int ldw, int col0, const float* b, const float* temb,
                              int dim, int T, int N, int apply_mish, float* out /*[T,N]*/,
                              hipStream_t st);

struct StepP {
  int H, D, act, obs, T;
  float scale;
  int t_stopgrad, scale_by_std, state_mask;
  float cgw;
  int guide_type;
  int n_cond;
  int cond_index[4];
  const float* cond[4];
};
hipError_t launch_guide_update(const StepP& s, float* x, const float* rg, const float* cg,
                               const float* cy, const float* thr, const long long* t,
                               const float* plv, int B, hipStream_t st);
hipError_t launch_ddpm_update(const StepP& s, float* x, const float* x0, const float* noise,
                              const long long* t, const float* plv, const float* c1,
                              const float* c2, int B, float* chain_out, hipStream_t st);
hipError_t launch_init_x(const StepP& s, float* x, const float* noise0, int B, float* chain_out,
                         hipStream_t st);
hipError_t launch_policy_prepare(const float* obs, int n_obs, int obs_dim, const float* mean, const float* stdv, int B,
                                 float* cond, const float* thr, int n_thr, float* thr_out, hipStream_t st);
hipError_t launch_policy_action(const float* x, int H, int D, int act, const long long* index, int n, int B,
                                const float* mean, const float* stdv, float* out, hipStream_t st);
hipError_t launch_pack_bf16x3(const float* w32, float* w16, int steps16, int Npad, hipStream_t st);
hipError_t launch_fill_t(long long* t, int B, int value, hipStream_t st);
hipError_t launch_fill_f(float* p, int n, float value, hipStream_t st);
hipError_t launch_feasibility(const float* ry, const float* cy, const float* thr, int n, float* out,
                              hipStream_t st);
hipError_t launch_argsort_desc(const float* v, int n, long long* order, hipStream_t st);
hipError_t launch_group_argmax(const float* v, int ng, int per, long long* best, hipStream_t st);


// ---- training step (train.hip; reference: diffusion.py:229-261, helpers.py:152-167, training.py:19-35,104-122) ----
// GroupNorm+Mish backward of one Conv1dBlock, standalone (the planning path fuses it into the next conv's prologue;
// training also needs dY in memory for the weight gradient, and the gamma / beta / bias / time-bias gradients).
struct GnbP {
  const float* dA; int ldA;        // dL/d(Mish output) [B*L, ldA] (a column slice of a wider buffer is fine)
  const float* xhat;               // [B*L, C] saved by the forward
  const float* stats;              // [B, 8, 2] (mean, rstd)
  const float* gamma; const float* beta;
  float* dY;                       // [B*L, C] dL/d(conv output)
  int B, L, C, gch;
  float* pcol;                     // [nblk][3][C] per-workgroup partial sums: d gamma, d beta, d bias(conv)
  float* ptime;                    // [nblk][T][C] per-workgroup partial sums of dA over positions, by timestep; or null
  const long long* t; int T;
};
hipError_t launch_gnb(const GnbP& p, int nblk, hipStream_t st);

// Weight gradient of a convolution as a GEMM over (candidate, position):
//   dW[cb][ca][tap] = sum_{b, l} A[b, l*stride + tap - pad, ca] * Bm[b, l, cb]
// (Conv1d: A = layer input, Bm = dY;  ConvTranspose1d: A = dY, Bm = layer input: both give the reference's
// [dim0][dim1][k] parameter layout).  Workgroup (tile, ks) writes its partial tile to slab[ks][tap][cb][ca].
struct WgP {
  const float* A; int ldA, CA, LA;
  const float* Bm; int ldB, CB, LB;
  int stride, pad, k;
  int B, S, spw;                   // S candidates per LDS stage (multiple of 4), spw candidates per workgroup
  int nwa;                         // waves = 16-channel A tiles per workgroup (TA = 16*nwa)
  int NB;                          // 16-channel B tiles per workgroup (1, 2 or 4)
  int SA, SB, ldAs, ldBs;          // LDS strides (floats): per candidate / per row
  int CA16, CB16, vecA, vecB;
  unsigned mLA, mLB;               // magic multipliers of LA, LB (0 for 1)
  int pf;                          // 1: the next chunk is prefetched into registers behind the MFMA loop
  float* slab; long slab_stride;   // floats per ks slice = k*CB16*CA16
};
hipError_t train_init_device();
hipError_t launch_wgrad(const WgP& p, int ksplit, hipStream_t st);
hipError_t launch_wreduce(const float* slab, int ksplit, long slab_stride, int k, int CA, int CB, int CA16, int CB16,
                          float* grad, hipStream_t st);
hipError_t launch_colsum(const float* src, int ld, long rows, int C, float* partial /*[nblk][C]*/, int nblk, hipStream_t st);
struct RJob { const float* src; float* dst; int nblk; int stride; int n; int assign; };   // dst (+)= sum_blk src[blk*stride + i]
hipError_t launch_reduce_jobs(const RJob* dev_jobs, int njobs, int max_n, hipStream_t st);
hipError_t launch_add_cols(float* dst, int ldd, const float* src, int lds, long rows, int C, hipStream_t st);
hipError_t launch_gather(float* dst, const int* map, long n, const float* params, hipStream_t st);
hipError_t launch_make_map(const float* coded, int* map, long n, hipStream_t st);
hipError_t launch_qsample(const StepP& s, const float* x0, const float* noise, const long long* t, const float* sa,
                          const float* sb, float* out, int B, hipStream_t st);
// weighted loss (helpers.py:159-167) + its gradient wrt the net output; loss_out[0] += weighted loss, [1] += a0 loss
constexpr int LOSS_BLOCKS = 512;
hipError_t launch_loss(const StepP& s, const float* pred, const float* targ, const float* w, int l1, float inv_count,
                       float inv_a0, float gscale, float* dout, float* partial, float* loss_out, int B, hipStream_t st);
hipError_t launch_adam_ema(float* p, float* g, float* m, float* v, float* ema, long n, float b1, float b2, float step_size,
                           float inv_bc2_sqrt, float eps, int ema_mode, float ema_decay, hipStream_t st);
hipError_t launch_time_embed_train(const float* w1, const float* b1, const float* w2, const float* b2, int dim, int T,
                                   float* e /*[T,dim]*/, float* a1 /*[T,4dim]*/, float* temb /*[T,dim]*/, hipStream_t st);
struct TBJob { const float* G; const float* Wt; float* gW; float* gb; float* DT; int C; int pad; };
hipError_t launch_time_block_bwd(const TBJob* dev_jobs, int njobs, const float* temb, int dim, int T, hipStream_t st);
hipError_t launch_time_mlp_bwd(const float* DT, int njobs, const float* e, const float* a1, const float* temb,
                               const float* W3, int dim, int T, float* gW1, float* gb1, float* gW3, float* gb3,
                               hipStream_t st);

}  // namespace trebi
