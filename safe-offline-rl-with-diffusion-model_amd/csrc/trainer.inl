// trainer.inl -- host side of the TemporalUnet training step (included at the end of trebi.hip; shares its
// Builder / Op / Resolver machinery).  Reference: GaussianDiffusion.loss / p_losses / q_sample
// (diffuser/models/diffusion.py:229-261), WeightedL2 (helpers.py:152-167,198-201), Trainer.train's inner step and
// EMA (diffuser/utils/training.py:19-35,94-98,104-122).
//
// Design.  Parameters, gradients, Adam moments and the EMA copy are FLAT float buffers in the reference's
// state_dict order and layouts (caller-owned: the Python mirror views them as the modules' parameters, and the
// data-parallel all-reduce is one collective on the flat gradient).  The kernels run on packed copies of the
// weights; packing is a pure permutation + zero fill, so the trainer builds its net ONCE from index-coded tensors
// (value = 1 + flat index, exact in float32 below 2^24), turns every uploaded region into an int32 gather map,
// and after each optimiser step re-packs with one gather per weight chunk and re-runs the small time-table kernels.
// Forward = the planning path's conv_gemm_kernel launches with the GroupNorm operands saved; backward = the step
// list built below (gnb_kernel -> wgrad_kernel + wreduce_kernel, conv_gemm_kernel as input-gradient conv).

namespace {

struct ParamIndex {
  std::unordered_map<std::string, long> off;
  std::vector<ParamSpec> specs;
  long total = 0;
};
ParamIndex param_index(const trebi_net_desc& d) {
  ParamIndex pi;
  pi.specs = param_specs(d);
  for (auto& s : pi.specs) {
    pi.off[s.name] = pi.total;
    pi.total += s.numel;
  }
  return pi;
}

// ------------------------------------------------------------------------------------
// backward step list of a TemporalUnet (temporal.py:120-151 walked in reverse)
// ------------------------------------------------------------------------------------
void build_unet_train_bwd(Builder& B, const ArchPlan& a) {
  trebi_net* net = B.net;
  const trebi_net_desc& d = net->desc;
  const UnetRec& rec = net->rec;
  const int n = d.n_mults;
  if (n < 2) unsupported("the training step needs at least two resolution levels (dim_mults of length >= 2)");
  const ParamIndex pi = param_index(d);
  auto off = [&](const std::string& name) {
    auto it = pi.off.find(name);
    if (it == pi.off.end()) throw BadShape{"internal: no parameter '" + name + "'"};
    return it->second;
  };
  size_t maxLC = (size_t)rec.final_L * d.dim;
  for (auto* v : {&rec.down0, &rec.down1, &rec.up0, &rec.up1})
    for (auto& r : *v) maxLC = std::max(maxLC, (size_t)r.L * r.cout);
  maxLC = std::max(maxLC, (size_t)rec.mid1.L * rec.mid1.cout);
  net->buf_dy = net->new_buf(maxLC);          // dL/d(conv output) of the Conv1dBlock in flight

  auto push = [&](TOp& t) { net->tops.push_back(t); };
  auto gnb = [&](const std::string& cb, BufRef dA, int xhat, int stats, const GnW& gn, int L, int C, int time_blk) {
    if (gn.gch % 4 || (16 % (gn.gch / 4)) || C % 4) unsupported("GroupNorm backward needs groups of 4..64 channels");
    TOp t;
    t.kind = 1; t.name = cb + " (gn+mish bwd)";
    t.dA = dA; t.xhat = xhat; t.stats = stats; t.dY = net->buf_dy; t.gn = gn; t.L = L; t.C = C;
    t.off_gamma = off(cb + ".block.2.weight"); t.off_beta = off(cb + ".block.2.bias"); t.off_bias = off(cb + ".block.0.bias");
    t.time_blk = time_blk;
    push(t);
  };
  auto wgrad = [&](const std::string& wname, BufRef A, int CA, int LA, BufRef Bm, int CB, int LB, int stride, int pad,
                   int k) {
    if (k > 5) unsupported("weight gradient of a %d-tap convolution", k);
    TOp t;
    t.kind = 2; t.name = wname + " (wgrad)";
    t.A = A; t.CA = CA; t.LA = LA; t.Bm = Bm; t.CB = CB; t.LB = LB; t.stride = stride; t.pad = pad; t.k = k;
    t.off_w = off(wname);
    t.flops = 2.0 * CA * CB * Builder::valid_pairs(LA, LB, stride, pad, 0, k);
    push(t);
  };
  auto colsum = [&](const std::string& bname, BufRef src, int L, int C) {
    TOp t;
    t.kind = 3; t.name = bname + " (bias grad)";
    t.src = src; t.L = L; t.C = C; t.off_bias = off(bname);
    push(t);
  };
  auto dgrad = [&](Op& op) {
    if (!op.ntaps) unsupported("internal: empty input-gradient conv %s", op.name.c_str());
    TOp t;
    t.kind = 0; t.name = op.name; t.conv = op; t.flops = op.flops;
    push(t);
  };
  auto add = [&](BufRef dst, BufRef src, int L, int C) {
    TOp t;
    t.kind = 4; t.name = "skip gradient add";
    t.dst2 = dst; t.src = src; t.L = L; t.C = C;
    push(t);
  };

  // ResidualTemporalBlock backward (temporal.py:36-46): g = dL/d(out) -> *dst = dL/d(in) (dst null: not needed)
  auto rb_bwd = [&](const RBInfo& r, BufRef g, const BufRef* dst) {
    const int tb = (int)net->time_blocks.size();
    net->time_blocks.push_back(r.prefix);
    const BufRef dy{net->buf_dy, r.cout, 0};
    const std::string b0 = r.prefix + ".blocks.0", b1 = r.prefix + ".blocks.1";
    gnb(b1, g, r.y1, r.st1, r.gn1, r.L, r.cout, -1);
    wgrad(b1 + ".block.0.weight", {r.a0, r.cout, 0}, r.cout, r.L, dy, r.cout, r.L, 1, 2, 5);
    Op kb = B.conv_dgrad(b1 + ".block.0", r.cout, r.cout, 5, r.L, nullptr);
    const int da0 = net->new_buf((size_t)r.L * r.cout);
    kb.src = dy; kb.dst = {da0, r.cout, 0};
    dgrad(kb);
    gnb(b0, {da0, r.cout, 0}, r.y0, r.st0, r.gn0, r.L, r.cout, tb);          // + the time-bias gradient (temporal.py:44)
    wgrad(b0 + ".block.0.weight", r.src, r.cin, r.L, dy, r.cout, r.L, 1, 2, 5);
    if (r.has_res) {
      wgrad(r.prefix + ".residual_conv.weight", r.src, r.cin, r.L, g, r.cout, r.L, 1, 0, 1);
      colsum(r.prefix + ".residual_conv.bias", g, r.L, r.cout);
    }
    if (dst) {
      const std::string rname = r.prefix + ".residual_conv";
      Op kc = B.conv_dgrad(b0 + ".block.0", r.cin, r.cout, 5, r.L, r.has_res ? &rname : nullptr);
      kc.src = dy;
      if (r.has_res) kc.src2 = g; else kc.res = g;
      kc.dst = *dst;
      dgrad(kc);
    }
  };

  // ---- final_conv (temporal.py:115-118) ----
  const int H = rec.final_L, dim = d.dim, D = d.transition_dim;
  const BufRef dout{EXT_GRAD, 0, 0};
  wgrad("final_conv.1.weight", {rec.final_fb, dim, 0}, dim, H, dout, D, H, 1, 0, 1);
  colsum("final_conv.1.bias", dout, H, D);
  Op f1 = B.conv_dgrad("final_conv.1", dim, D, 1, H, nullptr);
  const int dfb = net->new_buf((size_t)H * dim);
  f1.src = dout; f1.dst = {dfb, dim, 0};
  dgrad(f1);
  gnb("final_conv.0", {dfb, dim, 0}, rec.final_y, rec.final_st, rec.final_gn, H, dim, -1);
  wgrad("final_conv.0.block.0.weight", rec.final_src, dim, H, {net->buf_dy, dim, 0}, dim, H, 1, 2, 5);
  Op f0 = B.conv_dgrad("final_conv.0.block.0", dim, dim, 5, H, nullptr);
  const int gf = net->new_buf((size_t)H * dim);
  f0.src = {net->buf_dy, dim, 0}; f0.dst = {gf, dim, 0};
  dgrad(f0);
  BufRef g{gf, dim, 0};

  // ---- up path in reverse (temporal.py:141-146) ----
  std::vector<int> dcat(n, NONE);                    // gradient of the concat buffer of level j  [L_j, 2 C_j]
  for (int k = n - 2; k >= 0; --k) {
    const int j = n - 1 - k;
    const RBInfo &r0 = rec.up0[k], &r1 = rec.up1[k];
    const ResampleInfo& u = rec.ups[k];
    if (!u.name.empty()) {                           // ConvTranspose1d: g is dL/d(out) [2 Lin, C]
      wgrad(u.name + ".weight", g, u.C, 2 * u.Lin, u.src, u.C, u.Lin, 2, 1, 4);
      colsum(u.name + ".bias", g, 2 * u.Lin, u.C);
      Op ud = B.convT_dgrad(u.name, u.C, u.Lin);
      const int b = net->new_buf((size_t)u.Lin * u.C);
      ud.src = g; ud.dst = {b, u.C, 0};
      dgrad(ud);
      g = {b, u.C, 0};
    }
    const BufRef d1{net->new_buf((size_t)r1.L * r1.cin), r1.cin, 0};
    rb_bwd(r1, g, &d1);
    dcat[j] = net->new_buf((size_t)r0.L * r0.cin);
    const BufRef dc{dcat[j], r0.cin, 0};
    rb_bwd(r0, d1, &dc);
    g = {dcat[j], r0.cin, 0};                        // left half: the main path (first rec.catC[k] channels)
  }
  // ---- mid blocks ----
  const BufRef dm2{net->new_buf((size_t)rec.mid2.L * rec.mid2.cin), rec.mid2.cin, 0};
  rb_bwd(rec.mid2, g, &dm2);
  const BufRef dm1{net->new_buf((size_t)rec.mid1.L * rec.mid1.cin), rec.mid1.cin, 0};
  rb_bwd(rec.mid1, dm2, &dm1);
  g = dm1;
  net->top_split = (int)net->tops.size();
  // ---- down path in reverse; the skip gradients (right halves of dcat) join before downs.i.1 ----
  for (int i = n - 1; i >= 0; --i) {
    const RBInfo &r0 = rec.down0[i], &r1 = rec.down1[i];
    const int C = r1.cout, L = r1.L;
    const ResampleInfo& dn = rec.downs[i];
    if (!dn.name.empty()) {                          // Conv1d(C, C, 3, 2, 1): g is dL/d(out) [L/2, C]
      const int Lo = down_len(L);
      wgrad(dn.name + ".weight", dn.src, C, L, g, C, Lo, 2, 1, 3);
      colsum(dn.name + ".bias", g, Lo, C);
      Op dd = B.down_dgrad(dn.name, C, L);
      const int b = net->new_buf((size_t)L * C);
      dd.src = g; dd.dst = {b, C, 0};
      dgrad(dd);
      g = {b, C, 0};
    }
    if (dcat[i] != NONE) {                           // h[i] was popped by the up path: add its gradient
      const BufRef skipg{dcat[i], 2 * C, C};
      add(skipg, g, L, C);
      g = skipg;
    }
    const BufRef d1{net->new_buf((size_t)L * r1.cin), r1.cin, 0};
    rb_bwd(r1, g, &d1);
    if (i == 0) {
      rb_bwd(r0, d1, nullptr);                       // the gradient wrt the noisy input is not needed
    } else {
      const BufRef d0{net->new_buf((size_t)L * r0.cin), r0.cin, 0};
      rb_bwd(r0, d1, &d0);
      g = d0;
    }
  }
  for (auto& t : net->tops) net->flops_train_bwd += t.flops;
  (void)a;
}

struct WgCfg {
  int nwa, NB, S, spw, ksplit, SA, SB, ldAs, ldBs, CA16, CB16, pf;
  size_t slab;                      // floats
  long slab_stride;
};
WgCfg wg_config(const TOp& t, int B) {
  WgCfg c{};
  c.CA16 = rup(t.CA, 16); c.CB16 = rup(t.CB, 16);
  c.nwa = c.CA16 % 64 == 0 ? 4 : (c.CA16 % 32 == 0 ? 2 : 1);
  c.NB = c.CB16 % 64 == 0 ? 4 : (c.CB16 % 32 == 0 ? 2 : 1);
  c.ldAs = 16 * c.nwa + 4; c.ldBs = 16 * c.NB + 4;
  // candidate stride = 16 (mod 64) floats: the four candidates of one MFMA operand read land in disjoint bank groups
  auto pad16 = [](int v) { return v + (((16 - v % 64) % 64) + 64) % 64; };
  c.SA = pad16(t.LA * c.ldAs); c.SB = pad16(t.LB * c.ldBs);
  static const int lds_kb = getenv("TREBI_WG_LDS_KB") ? atoi(getenv("TREBI_WG_LDS_KB")) : 72;        // diagnostics
  static const int wg_target = getenv("TREBI_WG_TARGET") ? atoi(getenv("TREBI_WG_TARGET")) : 512;
  int S = (lds_kb * 1024 / 4) / (c.SA + c.SB) / 4 * 4;
  S = std::max(4, std::min(64, S));
  // the kernel prefetches the next chunk into registers where that takes at most 8 float4 per thread and operand
  const int Spf = std::min(std::min(S, 128 / t.LA / 4 * 4), 128 * c.nwa / (t.LB * c.NB) / 4 * 4);
  c.pf = Spf >= 4;
  if (c.pf) S = Spf;
  if ((size_t)S * (c.SA + c.SB) * sizeof(float) > 160 * 1024)
    unsupported("weight gradient of a layer of length %d (%d x %d channels) is not tiled", std::max(t.LA, t.LB), t.CA, t.CB);
  c.S = S;
  const int ntiles = (c.CA16 / (16 * c.nwa)) * (c.CB16 / (16 * c.NB));
  // two-wave workgroups (the 32-channel layers: one tile set covers the whole weight matrix) get more of them: at 512 there
  // is ONE wave per SIMD and nothing hides a wait
  static const int wg_target2 = getenv("TREBI_WG_TARGET2") ? atoi(getenv("TREBI_WG_TARGET2")) : 1024;   // measured 13.23 (512) -> 13.11 (1024), 13.09 (2048), 13.11 (4096) ms per step
  const int target = c.nwa <= 2 ? wg_target2 : wg_target;
  int ks = std::max(1, std::min((target + ntiles - 1) / ntiles, (B + S - 1) / S));      // ~two workgroups per CU
  c.spw = rup((B + ks - 1) / ks, S);
  c.ksplit = (B + c.spw - 1) / c.spw;
  c.slab_stride = (long)t.k * c.CB16 * c.CA16;
  c.slab = (size_t)c.ksplit * c.slab_stride;
  return c;
}

constexpr int NBLK = 256;           // workgroups (= partial tables) of the GroupNorm-backward and column-sum kernels

}  // namespace

struct trebi_trainer {
  trebi_ctx* ctx = nullptr;
  trebi_train_desc d{};
  trebi_net* net = nullptr;
  int maxB = 0;
  ParamIndex pi;
  std::vector<int*> maps;           // per weight chunk of the net: packed position -> flat parameter index (or < 0)
  void* arena = nullptr;
  float *ws = nullptr, *xnoisy = nullptr, *pred = nullptr, *dout = nullptr, *wslab = nullptr;
  float *lw = nullptr, *sa = nullptr, *sb = nullptr, *G = nullptr, *DT = nullptr, *lossp = nullptr;
  std::vector<float*> part;         // per backward step: its partial-sum table (kinds 1 and 3), or null
  std::vector<float*> part_time;    // per backward step: its per-timestep partial table (kind 1 with a time block)
  // reduction jobs, those of the tail bucket (final_conv, up path, mid blocks: tops[0..top_split)) first
  RJob* dev_rjobs = nullptr;
  int n_rjobs = 0, n_rjobs_tail = 0, max_rjob_n = 0;
  TBJob* dev_tbjobs = nullptr;
  int n_tb = 0, n_tb_tail = 0;
  size_t split_off = 0;             // flat offset of the first tail-bucket parameter ("ups.0..." or "mid_block1...")
  hipEvent_t bucket_event = nullptr;   // caller-owned: recorded when grads[split_off, total) are final
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *ema = nullptr;
  bool bound = false;
  std::map<int, std::vector<Tile>> tiles_fwd, tiles_bwd;
  ~trebi_trainer() {
    for (int* p : maps) (void)hipFree(p);
    if (arena) (void)hipFree(arena);
    if (dev_rjobs) (void)hipFree(dev_rjobs);
    if (dev_tbjobs) (void)hipFree(dev_tbjobs);
    delete net;
  }
};

static int trainer_net_desc(const trebi_train_desc* d, trebi_net_desc* nd) {
  if (!d) return fail(TREBI_BAD_ARG, "null training descriptor");
  if (d->n_mults < 1 || d->n_mults > 8) return fail(TREBI_BAD_ARG, "n_mults must be in 1..8");
  if (d->observation_dim < 0 || d->action_dim < 0 || d->horizon < 1 || d->n_timesteps < 1)
    return fail(TREBI_BAD_ARG, "inconsistent training descriptor");
  if (d->n_cond < 0 || d->n_cond > 4) return fail(TREBI_UNSUPPORTED_CONFIG, "at most 4 conditioned timesteps");
  memset(nd, 0, sizeof *nd);
  nd->kind = 0;
  nd->horizon = d->horizon;
  nd->transition_dim = d->observation_dim + d->action_dim;
  nd->dim = d->dim;
  nd->n_mults = d->n_mults;
  for (int i = 0; i < d->n_mults; ++i) nd->dim_mults[i] = d->dim_mults[i];
  nd->max_timesteps = d->n_timesteps;
  return check_desc(nd);
}

extern "C" int trebi_trainer_create(trebi_ctx* ctx, const trebi_train_desc* desc, int max_batch, trebi_trainer** out) {
  if (!ctx || !out || max_batch < 1) return fail(TREBI_BAD_ARG, "null argument");
  trebi_net_desc nd;
  if (int e = trainer_net_desc(desc, &nd)) return e;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(train_init_device());
  std::unique_ptr<trebi_trainer> tr(new trebi_trainer);
  tr->ctx = ctx; tr->d = *desc; tr->maxB = max_batch;
  tr->pi = param_index(nd);
  const long P = tr->pi.total;
  {                                  // the tail bucket is the state_dict's tail: ups.*, mid_block*, final_conv.* back to back
    long first = P;
    for (auto& sp : tr->pi.specs) {
      const bool tail = sp.name.rfind("ups.", 0) == 0 || sp.name.rfind("mid_block", 0) == 0 || sp.name.rfind("final_conv", 0) == 0;
      const long o = tr->pi.off[sp.name];
      if (tail) first = std::min(first, o);
      else if (o >= first) return fail(TREBI_HIP_ERROR, "internal: parameter order (%s after the up path)", sp.name.c_str());
    }
    tr->split_off = (size_t)first;
  }
  if (P + 1 >= (1 << 24))
    return fail(TREBI_UNSUPPORTED_CONFIG, "%ld parameters: the index-coded pack needs fewer than 2^24", P);
  // ---- build the net from index-coded tensors, then turn every uploaded region into a gather map ----
  {
    std::vector<float> coded((size_t)P);
    for (long i = 0; i < P; ++i) coded[i] = (float)(i + 1);
    std::vector<const char*> names;
    std::vector<const float*> tens;
    std::vector<int64_t> numels;
    for (auto& s : tr->pi.specs) {
      names.push_back(s.name.c_str());
      tens.push_back(coded.data() + tr->pi.off[s.name]);
      numels.push_back(s.numel);
    }
    if (int e = net_create_impl(ctx, &nd, names.data(), tens.data(), numels.data(), (int)names.size(), true, &tr->net))
      return e;
  }
  trebi_net* net = tr->net;
  // The backward keeps per-timestep tables in LDS: gnb_kernel T x C floats (+ 24 KB of staging) for a block with a time bias,
  // time_mlp_bwd_kernel T x 14 dim + 4 dim^2 floats.  Refuse here what does not fit (the reference's default
  // n_timesteps = 1000 does not; its configs use 20), not at the first accumulate with a bare launch error.
  {
    const size_t T = (size_t)desc->n_timesteps, dim = (size_t)nd.dim;
    size_t gnb_max = 0;
    for (auto& t : net->tops)
      if (t.kind == 1) gnb_max = std::max(gnb_max, ((size_t)(t.time_blk >= 0 ? T : 0) * t.C + (size_t)64 * 3 * t.gn.gch + 64) * sizeof(float));
    const size_t tmlp = (T * 14 * dim + 4 * dim * dim) * sizeof(float);
    if (gnb_max > 160 * 1024 || tmlp > 160 * 1024)
      return fail(TREBI_UNSUPPORTED_CONFIG, "n_timesteps = %d with dim = %d: the training backward keeps per-timestep tables in "
                  "LDS (GroupNorm backward %zu bytes, time-MLP backward %zu bytes; limit 163840)", desc->n_timesteps, nd.dim,
                  gnb_max, tmlp);
  }
  for (size_t c = 0; c < net->allocs.size(); ++c) {
    int* mp = nullptr;
    HIPCHK(hipMalloc((void**)&mp, net->alloc_bytes[c]));
    tr->maps.push_back(mp);
    HIPCHK(hipMemset(mp, 0xFF, net->alloc_bytes[c]));                    // -1: not a parameter (zero fill, tables)
  }
  for (auto& u : net->uploads) {
    size_t c = 0;
    for (; c < net->allocs.size(); ++c)
      if ((char*)u.first >= (char*)net->allocs[c] && (char*)u.first < (char*)net->allocs[c] + net->alloc_bytes[c]) break;
    if (c == net->allocs.size()) return fail(TREBI_HIP_ERROR, "internal: upload outside the weight chunks");
    const size_t o = ((char*)u.first - (char*)net->allocs[c]) / sizeof(float);
    HIPCHK(launch_make_map(u.first, tr->maps[c] + o, (long)u.second, 0));
  }
  HIPCHK(hipDeviceSynchronize());
  // ---- one arena: activations / gradients workspace, slabs, partial tables, small tables ----
  const int B = max_batch, H = desc->horizon, D = nd.transition_dim, T = desc->n_timesteps, dim = nd.dim;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  size_t n_ws = al(net->ws_floats(B)), n_x = al((size_t)B * H * D), n_slab = 0, n_part = 0;
  try {
    for (auto& t : net->tops) {
      if (t.kind == 2) n_slab = std::max(n_slab, wg_config(t, B).slab);
      if (t.kind == 1) n_part += al((size_t)NBLK * 3 * t.C) + (t.time_blk >= 0 ? al((size_t)NBLK * T * t.C) : 0);
      if (t.kind == 3) n_part += al((size_t)NBLK * t.C);
    }
  } catch (const Unsupported& u) {
    return fail(TREBI_UNSUPPORTED_CONFIG, "%s", u.msg.c_str());
  }
  const int nblk_t = (int)net->time_blocks.size();
  size_t n_G = 0;
  std::vector<int> blkC(nblk_t, 0);
  for (auto& t : net->tops)
    if (t.kind == 1 && t.time_blk >= 0) blkC[t.time_blk] = t.C;
  for (int c : blkC) n_G += al((size_t)T * c);
  const size_t n_DT = al((size_t)nblk_t * T * dim), n_lw = al((size_t)H * D), n_tab = al(2 * (size_t)T);
  const size_t n_lossp = al(2 * LOSS_BLOCKS);
  const size_t total = n_ws + 3 * n_x + al(n_slab) + n_part + n_G + n_DT + n_lw + n_tab + n_lossp;
  HIPCHK(hipMalloc(&tr->arena, total * sizeof(float)));
  float* p = (float*)tr->arena;
  tr->ws = p; p += n_ws;
  tr->xnoisy = p; p += n_x;
  tr->pred = p; p += n_x;
  tr->dout = p; p += n_x;
  tr->wslab = p; p += al(n_slab);
  tr->part.assign(net->tops.size(), nullptr);
  tr->part_time.assign(net->tops.size(), nullptr);
  for (size_t i = 0; i < net->tops.size(); ++i) {
    const TOp& t = net->tops[i];
    if (t.kind == 1) {
      tr->part[i] = p; p += al((size_t)NBLK * 3 * t.C);
      if (t.time_blk >= 0) { tr->part_time[i] = p; p += al((size_t)NBLK * T * t.C); }
    } else if (t.kind == 3) {
      tr->part[i] = p; p += al((size_t)NBLK * t.C);
    }
  }
  tr->G = p; p += n_G;
  tr->DT = p; p += n_DT;
  tr->lw = p; p += n_lw;
  tr->sa = p; tr->sb = p + T; p += n_tab;
  tr->lossp = p; p += n_lossp;
  *out = tr.release();
  return 0;
}
extern "C" void trebi_trainer_destroy(trebi_trainer* tr) { delete tr; }

extern "C" int64_t trebi_trainer_param_count(const trebi_trainer* tr) { return tr ? (int64_t)tr->pi.total : 0; }

extern "C" int64_t trebi_trainer_bucket_split(const trebi_trainer* tr) { return tr ? (int64_t)tr->split_off : 0; }

extern "C" int trebi_trainer_set_bucket_event(trebi_trainer* tr, void* event) {
  if (!tr) return fail(TREBI_BAD_ARG, "null trainer");
  tr->bucket_event = (hipEvent_t)event;
  return 0;
}

extern "C" double trebi_trainer_flops(const trebi_trainer* tr, int backward) {
  if (!tr) return 0.0;
  return backward ? tr->net->flops_train_bwd : tr->net->flops_fwd;
}

// Re-pack the kernels' weight copies from the flat parameters and re-run the time tables (temporal.py:28-32,70-75).
static int trainer_repack(trebi_trainer* tr, hipStream_t st) {
  trebi_net* net = tr->net;
  for (size_t c = 0; c < net->allocs.size(); ++c)
    HIPCHK(launch_gather((float*)net->allocs[c], tr->maps[c], (long)(net->alloc_bytes[c] / sizeof(float)), tr->params, st));
  const int dim = net->desc.dim, T = net->T;
  for (auto& j : net->time_jobs) {
    if (j.kind == 0)
      HIPCHK(launch_time_embed_train(j.w, j.b, j.w2, j.b2, dim, T, net->tt_e, net->tt_a1, net->tt_temb, st));
    else
      HIPCHK(launch_time_linear(j.w, j.ldw, j.col0, j.b, net->tt_temb, dim, T, j.N, j.mish, j.out, st));
  }
  if (net->prec == 1)
    if (int e = net_refresh_bf16(net, st)) return e;
  return 0;
}

extern "C" int trebi_trainer_repack(trebi_trainer* tr, void* stream) {
  if (!tr || !tr->bound) return fail(TREBI_BAD_ARG, "trainer is null or not bound");
  HIPCHK(hipSetDevice(tr->ctx->device));
  return trainer_repack(tr, (hipStream_t)stream);
}

extern "C" int trebi_trainer_bind(trebi_trainer* tr, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                  float* ema, const float* loss_weights, const float* sqrt_alphas_cumprod,
                                  const float* sqrt_one_minus_alphas_cumprod) {
  if (!tr || !params || !grads || !loss_weights || !sqrt_alphas_cumprod || !sqrt_one_minus_alphas_cumprod)
    return fail(TREBI_BAD_ARG, "null argument");
  HIPCHK(hipSetDevice(tr->ctx->device));
  HIPCHK(hipDeviceSynchronize());
  trebi_net* net = tr->net;
  const int H = tr->d.horizon, D = net->desc.transition_dim, T = tr->d.n_timesteps, dim = net->desc.dim;
  tr->params = params; tr->grads = grads; tr->m = exp_avg; tr->v = exp_avg_sq; tr->ema = ema;
  HIPCHK(hipMemcpy(tr->lw, loss_weights, (size_t)H * D * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(tr->sa, sqrt_alphas_cumprod, T * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(tr->sb, sqrt_one_minus_alphas_cumprod, T * sizeof(float), hipMemcpyHostToDevice));
  // ---- reduction jobs: partial tables -> flat gradient / per-block time tables ----
  std::vector<RJob> rj;
  std::vector<TBJob> tb(net->time_blocks.size());
  std::vector<float*> Gof(net->time_blocks.size(), nullptr);
  {
    float* g = tr->G;
    std::vector<int> blkC(net->time_blocks.size(), 0);
    for (auto& t : net->tops)
      if (t.kind == 1 && t.time_blk >= 0) blkC[t.time_blk] = t.C;
    for (size_t i = 0; i < blkC.size(); ++i) {
      Gof[i] = g;
      g += ((size_t)T * blkC[i] + 63) / 64 * 64;
      const std::string& pf = net->time_blocks[i];
      tb[i].G = Gof[i];
      tb[i].Wt = params + tr->pi.off[pf + ".time_mlp.1.weight"];
      tb[i].gW = grads + tr->pi.off[pf + ".time_mlp.1.weight"];
      tb[i].gb = grads + tr->pi.off[pf + ".time_mlp.1.bias"];
      tb[i].DT = tr->DT + (size_t)i * T * dim;
      tb[i].C = blkC[i];
      tb[i].pad = 0;
    }
  }
  for (int pass = 0; pass < 2; ++pass) {             // pass 0: the tail bucket's steps, pass 1: the down path's
    const size_t lo = pass ? (size_t)net->top_split : 0, hi = pass ? net->tops.size() : (size_t)net->top_split;
    for (size_t i = lo; i < hi; ++i) {
      const TOp& t = net->tops[i];
      if (t.kind == 1) {
        rj.push_back({tr->part[i], grads + t.off_gamma, NBLK, 3 * t.C, t.C, 0});
        rj.push_back({tr->part[i] + t.C, grads + t.off_beta, NBLK, 3 * t.C, t.C, 0});
        rj.push_back({tr->part[i] + 2 * t.C, grads + t.off_bias, NBLK, 3 * t.C, t.C, 0});
        if (t.time_blk >= 0) rj.push_back({tr->part_time[i], Gof[t.time_blk], NBLK, T * t.C, T * t.C, 1});
      } else if (t.kind == 3) {
        rj.push_back({tr->part[i], grads + t.off_bias, NBLK, t.C, t.C, 0});
      }
    }
    if (!pass) tr->n_rjobs_tail = (int)rj.size();
  }
  {                                                  // time blocks in the same two groups (each keeps its own DT slot)
    std::vector<TBJob> a, b;
    for (size_t i = 0; i < tb.size(); ++i)
      ((size_t)(tb[i].gW - grads) >= tr->split_off ? a : b).push_back(tb[i]);
    tr->n_tb_tail = (int)a.size();
    tb = a;
    tb.insert(tb.end(), b.begin(), b.end());
  }
  tr->n_rjobs = (int)rj.size();
  tr->max_rjob_n = 0;
  for (auto& j : rj) tr->max_rjob_n = std::max(tr->max_rjob_n, j.n);
  if (tr->dev_rjobs) (void)hipFree(tr->dev_rjobs);
  if (tr->dev_tbjobs) (void)hipFree(tr->dev_tbjobs);
  tr->dev_rjobs = nullptr; tr->dev_tbjobs = nullptr;
  HIPCHK(hipMalloc((void**)&tr->dev_rjobs, rj.size() * sizeof(RJob)));
  HIPCHK(hipMemcpy(tr->dev_rjobs, rj.data(), rj.size() * sizeof(RJob), hipMemcpyHostToDevice));
  tr->n_tb = (int)tb.size();
  HIPCHK(hipMalloc((void**)&tr->dev_tbjobs, tb.size() * sizeof(TBJob)));
  HIPCHK(hipMemcpy(tr->dev_tbjobs, tb.data(), tb.size() * sizeof(TBJob), hipMemcpyHostToDevice));
  tr->bound = true;
  if (int e = trainer_repack(tr, 0)) return e;
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

extern "C" int trebi_trainer_set_precision(trebi_trainer* tr, int mode, int min_batch) {
  if (!tr) return fail(TREBI_BAD_ARG, "null trainer");
  if (int e = trebi_net_set_precision(tr->net, mode, min_batch)) return e;
  if (tr->bound && mode == 1)
    if (int e = net_refresh_bf16(tr->net, 0)) return e;
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

namespace {
int launch_op(const Op& op, const Resolver& r, Tile tl, hipStream_t st) {
  ConvLaunch L{};
  L.p[0] = r.conv(op);
  tl.bf = 0;                                     // the chooser's assumption; the operands decide
  if (use_bf(r.net, op, r.B) && conv_vec_ok(L.p[0])) {
    tl.bf = 1;
    L.p[0].w = op.w16;
  }
  if (tl.S * op.Lout > 16 * tl.MI || op.Npad % (16 * tl.NI * tl.NG) || 64 * tl.NG * tl.KS > 512 || !L.p[0].x || !L.p[0].out)
    return fail(TREBI_UNSUPPORTED_CONFIG, "internal tiling error at layer %s", op.name.c_str());
  HIPCHK(launch_conv(L, 1, tl, st));
  return 0;
}
}  // namespace

extern "C" int trebi_trainer_accumulate(trebi_trainer* tr, const float* x_start, const float* const* cond,
                                        const int64_t* tsteps, const float* noise, int B, float grad_scale,
                                        float* loss_out, void* stream) {
  if (!tr || !x_start || !tsteps || !noise || !loss_out) return fail(TREBI_BAD_ARG, "null argument");
  if (!tr->bound) return fail(TREBI_BAD_ARG, "trebi_trainer_bind has not been called");
  if (B < 1 || B > tr->maxB) return fail(TREBI_BAD_SHAPE, "batch %d outside 1..%d (max_batch of trebi_trainer_create)", B, tr->maxB);
  if (tr->d.n_cond > 0 && !cond) return fail(TREBI_BAD_ARG, "cond is null");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(tr->ctx->device));
  trebi_net* net = tr->net;
  const trebi_train_desc& d = tr->d;
  const int H = d.horizon, D = net->desc.transition_dim, T = d.n_timesteps, dim = net->desc.dim;
  StepP sp{};
  sp.H = H; sp.D = D; sp.act = d.action_dim; sp.obs = d.observation_dim; sp.T = T; sp.n_cond = d.n_cond;
  for (int k = 0; k < d.n_cond; ++k) {
    if (!cond[k]) return fail(TREBI_BAD_ARG, "cond[%d] is null", k);
    if (d.cond_index[k] < 0 || d.cond_index[k] >= H) return fail(TREBI_BAD_SHAPE, "conditioned index outside the horizon");
    sp.cond_index[k] = d.cond_index[k];
    sp.cond[k] = cond[k];
  }
  // ---- tiles of every conv launch for this batch size (resolved once) ----
  // (per batch size AND operand mode: the chooser prices the fp32-only K loops differently)
  const int tkey = B | (net->prec == 1 ? 1 << 30 : 0);
  auto itf = tr->tiles_fwd.find(tkey);
  if (itf == tr->tiles_fwd.end()) {
    std::vector<Tile> tf(net->fwd.size()), tb(net->tops.size());
    try {
      for (size_t i = 0; i < net->fwd.size(); ++i) tf[i] = choose_tile(net->fwd[i], B, 1, use_bf(net, net->fwd[i], B));
      for (size_t i = 0; i < net->tops.size(); ++i)
        if (net->tops[i].kind == 0) tb[i] = choose_tile(net->tops[i].conv, B, 1, use_bf(net, net->tops[i].conv, B));
    } catch (const Unsupported& u) {
      return fail(TREBI_UNSUPPORTED_CONFIG, "%s", u.msg.c_str());
    }
    tr->tiles_bwd[tkey] = tb;
    itf = tr->tiles_fwd.emplace(tkey, tf).first;
  }
  const std::vector<Tile>& tf = itf->second;
  const std::vector<Tile>& tbw = tr->tiles_bwd[tkey];

  if (poison_on())
    if (int e = poison_ws(tr->ws, net->ws_floats(B), st)) return e;
  // ---- q_sample + conditioning, forward, loss (diffusion.py:240-254) ----
  HIPCHK(launch_qsample(sp, x_start, noise, (const long long*)tsteps, tr->sa, tr->sb, tr->xnoisy, B, st));
  Binding b;
  b.x = tr->xnoisy; b.ldx = D; b.out = tr->pred; b.ldo = D; b.grad = tr->dout; b.ldg = D;
  b.t = (const long long*)tsteps; b.t_uniform = -1; b.ws = tr->ws;
  Resolver r{net, net->offsets(B), b, B};
  for (size_t i = 0; i < net->fwd.size(); ++i)
    if (int e = launch_op(net->fwd[i], r, tf[i], st)) return e;
  const float inv_count = 1.f / ((float)B * H * D), inv_a0 = d.action_dim > 0 ? 1.f / ((float)B * d.action_dim) : 0.f;
  HIPCHK(launch_loss(sp, tr->pred, d.predict_epsilon ? noise : x_start, tr->lw, d.loss_l1, inv_count, inv_a0,
                     grad_scale * inv_count, tr->dout, tr->lossp, loss_out, B, st));

  // ---- backward ----
  // partial tables -> gradients and the per-block time Linears, per bucket: the tail bucket's parameters
  // (final_conv, up path, mid blocks) are final once tops[0..top_split) have run, the rest at the end.
  auto finish_bucket = [&](int which) -> int {
    const int r0 = which ? tr->n_rjobs_tail : 0, r1 = which ? tr->n_rjobs : tr->n_rjobs_tail;
    const int b0 = which ? tr->n_tb_tail : 0, b1 = which ? tr->n_tb : tr->n_tb_tail;
    if (r1 > r0) HIPCHK(launch_reduce_jobs(tr->dev_rjobs + r0, r1 - r0, tr->max_rjob_n, st));
    if (b1 > b0) HIPCHK(launch_time_block_bwd(tr->dev_tbjobs + b0, b1 - b0, net->tt_temb, dim, T, st));
    if (!which && tr->bucket_event) HIPCHK(hipEventRecord(tr->bucket_event, st));
    return 0;
  };
  for (size_t i = 0; i < net->tops.size(); ++i) {
    if ((int)i == net->top_split)
      if (int e = finish_bucket(0)) return e;
    const TOp& t = net->tops[i];
    int ld = 0;
    switch (t.kind) {
      case 0:
        if (int e = launch_op(t.conv, r, tbw[i], st)) return e;
        break;
      case 1: {
        GnbP g{};
        g.dA = r.ptr(t.dA, &g.ldA);
        g.xhat = r.buf(t.xhat); g.stats = r.buf(t.stats);
        g.gamma = t.gn.gamma; g.beta = t.gn.beta;
        g.dY = r.buf(t.dY);
        g.B = B; g.L = t.L; g.C = t.C; g.gch = t.gn.gch;
        g.pcol = tr->part[i]; g.ptime = tr->part_time[i];
        g.t = (const long long*)tsteps; g.T = T;
        if ((g.ldA & 3) || (((uintptr_t)g.dA) & 15)) return fail(TREBI_UNSUPPORTED_CONFIG, "internal: unaligned gradient at %s", t.name.c_str());
        HIPCHK(launch_gnb(g, NBLK, st));
        break;
      }
      case 2: {
        WgCfg c;
        try { c = wg_config(t, B); } catch (const Unsupported& u) { return fail(TREBI_UNSUPPORTED_CONFIG, "%s", u.msg.c_str()); }
        WgP w{};
        w.A = r.ptr(t.A, &w.ldA); w.CA = t.CA; w.LA = t.LA;
        w.Bm = r.ptr(t.Bm, &w.ldB); w.CB = t.CB; w.LB = t.LB;
        w.stride = t.stride; w.pad = t.pad; w.k = t.k;
        w.B = B; w.S = c.S; w.spw = c.spw; w.nwa = c.nwa; w.NB = c.NB;
        w.SA = c.SA; w.SB = c.SB; w.ldAs = c.ldAs; w.ldBs = c.ldBs; w.CA16 = c.CA16; w.CB16 = c.CB16;
        w.vecA = ((w.ldA & 3) == 0) && ((t.CA & 3) == 0) && ((((uintptr_t)w.A) & 15) == 0);
        w.vecB = ((w.ldB & 3) == 0) && ((t.CB & 3) == 0) && ((((uintptr_t)w.Bm) & 15) == 0);
        w.slab = tr->wslab; w.slab_stride = c.slab_stride;
        w.pf = c.pf && w.vecA && w.vecB;
        w.mLA = t.LA > 1 ? 0xFFFFFFFFu / (unsigned)t.LA + 1u : 0u;
        w.mLB = t.LB > 1 ? 0xFFFFFFFFu / (unsigned)t.LB + 1u : 0u;
        if (!w.A || !w.Bm) return fail(TREBI_HIP_ERROR, "internal: unresolved operand at %s", t.name.c_str());
        HIPCHK(launch_wgrad(w, c.ksplit, st));
#ifdef TREBI_STAMPS
        if (getenv("TREBI_WGRAD_STAMPS")) {      // diagnostic build: phase cycles of workgroup (0,0), thread 0
          unsigned long long sb[32];
          (void)hipDeviceSynchronize();
          (void)trebi_debug_read_wstamps(sb);
          fprintf(stderr, "wgrad %-40s CA16xCB16 %llu NB/nwa/k %llu S %llu LB %llu chunks %llu ksplit %d | total %llu commit %llu barrier %llu fetch %llu mma %llu\n",
                  t.name.c_str(), sb[9], sb[8], sb[6], sb[7], sb[5], c.ksplit, sb[4], sb[0], sb[1], sb[2], sb[3]);
        }
#endif
        HIPCHK(launch_wreduce(tr->wslab, c.ksplit, c.slab_stride, t.k, t.CA, t.CB, c.CA16, c.CB16, tr->grads + t.off_w, st));
        break;
      }
      case 3: {
        const float* src = r.ptr(t.src, &ld);
        HIPCHK(launch_colsum(src, ld, (long)B * t.L, t.C, tr->part[i], NBLK, st));
        break;
      }
      case 4: {
        int lds = 0;
        float* dst = const_cast<float*>(r.ptr(t.dst2, &ld));
        const float* src = r.ptr(t.src, &lds);
        HIPCHK(launch_add_cols(dst, ld, src, lds, (long)B * t.L, t.C, st));
        break;
      }
    }
  }
  // ---- partial tables -> gradients; the time-embedding path (tables over the T timesteps) ----
  if (int e = finish_bucket(1)) return e;
  HIPCHK(launch_time_mlp_bwd(tr->DT, tr->n_tb, net->tt_e, net->tt_a1, net->tt_temb,
                             tr->params + tr->pi.off["time_mlp.3.weight"], dim, T,
                             tr->grads + tr->pi.off["time_mlp.1.weight"], tr->grads + tr->pi.off["time_mlp.1.bias"],
                             tr->grads + tr->pi.off["time_mlp.3.weight"], tr->grads + tr->pi.off["time_mlp.3.bias"], st));
  return 0;
}

extern "C" int trebi_trainer_step(trebi_trainer* tr, int64_t adam_step, float lr, float beta1, float beta2, float eps,
                                  int ema_mode, float ema_decay, void* stream) {
  if (!tr || !tr->bound) return fail(TREBI_BAD_ARG, "trainer is null or not bound");
  if (!tr->m || !tr->v) return fail(TREBI_BAD_ARG, "no Adam moment buffers were bound");
  if (adam_step < 1) return fail(TREBI_BAD_ARG, "adam_step counts from 1");
  if (ema_mode < 0 || ema_mode > 2 || (ema_mode && !tr->ema)) return fail(TREBI_BAD_ARG, "bad ema_mode / no EMA buffer bound");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(tr->ctx->device));
  const double bc1 = 1.0 - std::pow((double)beta1, (double)adam_step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)adam_step);
  HIPCHK(launch_adam_ema(tr->params, tr->grads, tr->m, tr->v, tr->ema, tr->pi.total, beta1, beta2, (float)(lr / bc1),
                         (float)(1.0 / std::sqrt(bc2)), eps, ema_mode, ema_decay, st));
  return trainer_repack(tr, st);
}
