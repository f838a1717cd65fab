// Host side of libtrebi_hip.so: architecture plans, weight packing, op scheduling, the
// captured sampling loop and the extern "C" boundary declared in include/trebi_hip.h.
//
// Nothing here computes on the CPU: the host only lays weights out for the kernels
// (pure permutation / zero padding), sizes tiles, and enqueues launches.
#include "common.h"
#include "../../include/trebi_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include <array>

using namespace trebi;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(TREBI_HIP_ERROR, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                  __FILE__, __LINE__);                                                        \
  } while (0)

struct Unsupported {
  std::string msg;
};
static void unsupported(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Unsupported{buf};
}
struct BadShape {
  std::string msg;
};

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
struct trebi_ctx {
  int device;
};

// Scratch of the one entry point that allocates nothing (trebi_unet_forward_static): a static device array.
// In host code `g_static_scratch` is only a stand-in for the symbol, never an address a kernel can use: each
// GPU holds its own copy at its own address, looked up once per device in trebi_ctx_create (that device
// current) and kept here.  Host code uses g_scratch_ptr[device] only.
static __device__ __align__(256) unsigned char g_static_scratch[TREBI_STATIC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

// ------------------------------------------------------------------------------------
// plan data structures
// ------------------------------------------------------------------------------------
namespace {

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

struct BufRef {            // a [B, L, C] view: workspace buffer (id >= 0) or one of the call's tensors
  int id = -100;           // -100: none
  int ld = 0;              // row stride in floats (0: taken from the binding for external tensors)
  int coloff = 0;
};
constexpr int EXT_X = -1, EXT_OUT = -2, EXT_GRAD = -3, NONE = -100;

struct GnW {
  const float* gamma = nullptr;
  const float* beta = nullptr;
  int gch = 0;
};

struct Op {
  int kind = 0;            // 0 conv, 1 head
  std::string name;
  int Lin = 0, Lout = 0, stride = 1, pad = 0, tap_lo = 0, ntaps = 1;
  int Cin = 0, Cin_p = 0, Cin2 = 0, Cin2_p = 0, N = 0, Npad = 0, osplit = 0;
  const float* w = nullptr;
  const float* w16 = nullptr;       // bf16x3 copy of the packed weights (trebi_net_set_precision), or null
  const float* bias = nullptr;
  const float* bias2 = nullptr;
  bool gn = false;
  GnW gnw;
  const float* tbias = nullptr;
  BufRef src, src2, dst, res;
  int save_y = NONE, save_g = NONE, save_st = NONE;
  bool pro = false;
  GnW pro_gn;
  int pro_y = NONE, pro_g = NONE, pro_st = NONE;
  double flops = 0;        // algorithmic (zero-padding taps excluded), per candidate
  // head
  int v_buf = NONE, dv_buf = NONE;
};

// What the training backward needs to know about the forward graph of a TemporalUnet (filled by build_unet).
struct RBInfo {                      // one ResidualTemporalBlock
  std::string prefix;
  int cin = 0, cout = 0, L = 0;
  bool has_res = false;
  BufRef src, dst;                   // block input / output
  int a0 = NONE;                     // blocks.0 output (+ time bias) = blocks.1 input
  int y0 = NONE, st0 = NONE, y1 = NONE, st1 = NONE;   // saved xhat / (mean, rstd) of the two Conv1dBlocks
  GnW gn0, gn1;
};
struct ResampleInfo { std::string name; int C = 0, Lin = 0; BufRef src, dst; };
struct UnetRec {
  std::vector<RBInfo> down0, down1, up0, up1;
  std::vector<ResampleInfo> downs, ups;            // name empty: no resampling conv at that level
  std::vector<int> catC;                           // per up step: channels of the left (main path) half of the concat
  RBInfo mid1, mid2;
  BufRef final_src;
  int final_fb = NONE, final_y = NONE, final_st = NONE, final_L = 0;
  GnW final_gn;
};

// One step of the training backward (trainer.inl).
struct TOp {
  int kind = 0;                      // 0 input-gradient conv, 1 GroupNorm+Mish backward, 2 weight gradient, 3 column sum, 4 add
  std::string name;
  Op conv;
  // kind 1
  BufRef dA;
  int xhat = NONE, stats = NONE, dY = NONE, L = 0, C = 0;
  GnW gn;
  long off_gamma = -1, off_beta = -1, off_bias = -1;
  int time_blk = -1;                 // index of the ResidualTemporalBlock whose time bias this gradient feeds, or -1
  // kind 2:  dW[cb][ca][tap] = sum A[b, l*stride + tap - pad, ca] * Bm[b, l, cb]
  BufRef A, Bm;
  int CA = 0, LA = 0, CB = 0, LB = 0, stride = 1, pad = 0, k = 1;
  long off_w = -1;
  // kind 3: off_bias += column sums of src [B*L, C];  kind 4: dst2 += src
  BufRef src, dst2;
  double flops = 0;                  // algorithmic, per sample
};

struct Binding {
  const float* x = nullptr;
  int ldx = 0;
  float* out = nullptr;
  int ldo = 0;
  float* grad = nullptr;
  int ldg = 0;
  const long long* t = nullptr;
  int t_uniform = -1;
  float* ws = nullptr;
  float* y = nullptr;      // head output
};

}  // namespace

struct trebi_net {
  trebi_ctx* ctx = nullptr;
  trebi_net_desc desc{};
  int T = 0;
  std::vector<void*> allocs;                       // device allocations owned by the net
  std::vector<size_t> buf_per_sample;              // workspace buffers, floats per candidate
  std::vector<Op> fwd, bwd;
  // head
  const float* head_w1 = nullptr;
  const float* head_tb = nullptr;
  const float* head_w2 = nullptr;
  float head_b2 = 0.f;
  const float* head_b2_dev = nullptr;              // [hd] copies of head_b2 (a row of the chains' parameter table)
  int fc = 0, hd = 0;
  double flops_fwd = 0, flops_bwd = 0;
  // bf16x3 operand mode (trebi_net_set_precision): copies of the packed weights, engaged from bf_min_b candidates up
  int prec = 0, bf_min_b = 512;
  struct BfJob { const float* w32; float* w16; int steps16, Npad; };
  std::vector<BfJob> bf_jobs;
  void* bf_arena = nullptr;
  // training (trainer.inl): every uploaded region (so that it can be re-packed from the flat parameters), the
  // time-table jobs to re-run after a re-pack, the forward graph record and the backward step list
  bool train = false;
  std::vector<size_t> alloc_bytes;
  std::vector<std::pair<float*, size_t>> uploads;
  struct TimeJob { int kind; const float *w, *b, *w2, *b2; int ldw, col0, N, mish; float* out; };
  std::vector<TimeJob> time_jobs;
  float *tt_e = nullptr, *tt_a1 = nullptr, *tt_temb = nullptr;
  UnetRec rec;
  std::vector<TOp> tops;
  int top_split = 0;                // tops[0..top_split) finish the gradients of final_conv, the up path and the mid blocks
  std::vector<std::string> time_blocks;            // prefixes of the blocks with a time_mlp, in backward order of use
  int buf_dy = NONE, buf_colscratch = NONE;
  double flops_train_bwd = 0;

  ~trebi_net() {
    for (void* p : allocs) (void)hipFree(p);
    if (bf_arena) (void)hipFree(bf_arena);
  }
  int new_buf(size_t per_sample) {
    buf_per_sample.push_back(per_sample);
    return (int)buf_per_sample.size() - 1;
  }
  size_t ws_floats(int B) const {
    size_t n = 0;
    for (size_t s : buf_per_sample) n += (s * (size_t)B + 63) / 64 * 64;
    return n;
  }
  std::vector<size_t> offsets(int B) const {
    std::vector<size_t> o(buf_per_sample.size());
    size_t n = 0;
    for (size_t i = 0; i < o.size(); ++i) {
      o[i] = n;
      n += (buf_per_sample[i] * (size_t)B + 63) / 64 * 64;
    }
    return o;
  }
};

namespace {

// ------------------------------------------------------------------------------------
// expected parameter table (mirrors trebi_amd/spec.py; reference temporal.py:18-46,49-118,153-211)
// ------------------------------------------------------------------------------------
struct ParamSpec {
  std::string name;
  int64_t numel;
};

struct ArchPlan {
  // unet
  struct Lvl { int cin, cout, L; bool down; };
  std::vector<Lvl> downs;
  int mid_c = 0, mid_L = 0;
  struct Up { int cin, cout, L; bool up; };
  std::vector<Up> ups;
  int final_L = 0;
  // value
  struct VB { int cin, cout, L; };
  std::vector<VB> blocks;
  VB mid1{}, mid2{};
  int vfinal_L = 0, c3 = 0, fc_dim = 0;
};

int down_len(int L) { return (L + 2 - 3) / 2 + 1; }

ArchPlan make_arch(const trebi_net_desc& d) {
  ArchPlan a;
  std::vector<int> dims{d.transition_dim};
  for (int i = 0; i < d.n_mults; ++i) dims.push_back(d.dim * d.dim_mults[i]);
  int n = d.n_mults;
  if (d.kind == 0) {
    std::vector<int> hist;
    int h = d.horizon;
    for (int i = 0; i < n; ++i) {
      hist.push_back(h);
      bool last = i >= n - 1;
      a.downs.push_back({dims[i], dims[i + 1], h, !last});
      if (!last) h = std::max(h / 2, 1);
    }
    a.mid_c = dims[n];
    a.mid_L = h;
    for (int k = 0; k < n - 1; ++k) {
      int j = n - 1 - k;                                  // level whose skip is popped
      bool up = hist[n - 1 - k] != hist[n - 2 - k];
      a.ups.push_back({dims[j + 1] * 2, dims[j], h, up});
      h = hist[n - 2 - k];
    }
    a.final_L = h;
  } else {
    int L = d.horizon, h = d.horizon;
    for (int i = 0; i < n; ++i) {
      a.blocks.push_back({dims[i], dims[i + 1], L});
      L = down_len(L);
      if (!(i >= n - 1)) h = h / 2;
    }
    int c = dims[n];
    a.mid1 = {c, c / 2, L};
    L = down_len(L);
    h /= 2;
    a.mid2 = {c / 2, c / 4, L};
    L = down_len(L);
    h /= 2;
    a.vfinal_L = L;
    a.c3 = c / 4;
    a.fc_dim = a.c3 * std::max(h, 1);
  }
  return a;
}

void spec_resblock(std::vector<ParamSpec>& s, const std::string& p, int cin, int cout, int dim) {
  int cc[2][2] = {{cin, cout}, {cout, cout}};
  for (int j = 0; j < 2; ++j) {
    std::string q = p + ".blocks." + std::to_string(j) + ".block";
    s.push_back({q + ".0.weight", (int64_t)cc[j][1] * cc[j][0] * 5});
    s.push_back({q + ".0.bias", cc[j][1]});
    s.push_back({q + ".2.weight", cc[j][1]});
    s.push_back({q + ".2.bias", cc[j][1]});
  }
  s.push_back({p + ".time_mlp.1.weight", (int64_t)cout * dim});
  s.push_back({p + ".time_mlp.1.bias", cout});
  if (cin != cout) {
    s.push_back({p + ".residual_conv.weight", (int64_t)cout * cin});
    s.push_back({p + ".residual_conv.bias", cout});
  }
}

std::vector<ParamSpec> param_specs(const trebi_net_desc& d) {
  ArchPlan a = make_arch(d);
  std::vector<ParamSpec> s;
  int dim = d.dim;
  s.push_back({"time_mlp.1.weight", (int64_t)4 * dim * dim});
  s.push_back({"time_mlp.1.bias", 4 * dim});
  s.push_back({"time_mlp.3.weight", (int64_t)4 * dim * dim});
  s.push_back({"time_mlp.3.bias", dim});
  auto S = [](int i) { return std::to_string(i); };
  if (d.kind == 0) {
    for (size_t i = 0; i < a.downs.size(); ++i) {
      auto& l = a.downs[i];
      spec_resblock(s, "downs." + S(i) + ".0", l.cin, l.cout, dim);
      spec_resblock(s, "downs." + S(i) + ".1", l.cout, l.cout, dim);
      if (l.down) {
        s.push_back({"downs." + S(i) + ".3.conv.weight", (int64_t)l.cout * l.cout * 3});
        s.push_back({"downs." + S(i) + ".3.conv.bias", l.cout});
      }
    }
    for (size_t i = 0; i < a.ups.size(); ++i) {
      auto& u = a.ups[i];
      spec_resblock(s, "ups." + S(i) + ".0", u.cin, u.cout, dim);
      spec_resblock(s, "ups." + S(i) + ".1", u.cout, u.cout, dim);
      if (u.up) {
        s.push_back({"ups." + S(i) + ".3.conv.weight", (int64_t)u.cout * u.cout * 4});
        s.push_back({"ups." + S(i) + ".3.conv.bias", u.cout});
      }
    }
    spec_resblock(s, "mid_block1", a.mid_c, a.mid_c, dim);
    spec_resblock(s, "mid_block2", a.mid_c, a.mid_c, dim);
    s.push_back({"final_conv.0.block.0.weight", (int64_t)dim * dim * 5});
    s.push_back({"final_conv.0.block.0.bias", dim});
    s.push_back({"final_conv.0.block.2.weight", dim});
    s.push_back({"final_conv.0.block.2.bias", dim});
    s.push_back({"final_conv.1.weight", (int64_t)d.transition_dim * dim});
    s.push_back({"final_conv.1.bias", d.transition_dim});
  } else {
    for (size_t i = 0; i < a.blocks.size(); ++i) {
      auto& b = a.blocks[i];
      spec_resblock(s, "blocks." + S(i) + ".0", b.cin, b.cout, dim);
      spec_resblock(s, "blocks." + S(i) + ".1", b.cout, b.cout, dim);
      s.push_back({"blocks." + S(i) + ".2.conv.weight", (int64_t)b.cout * b.cout * 3});
      s.push_back({"blocks." + S(i) + ".2.conv.bias", b.cout});
    }
    spec_resblock(s, "mid_block1", a.mid1.cin, a.mid1.cout, dim);
    s.push_back({"mid_down1.conv.weight", (int64_t)a.mid1.cout * a.mid1.cout * 3});
    s.push_back({"mid_down1.conv.bias", a.mid1.cout});
    spec_resblock(s, "mid_block2", a.mid2.cin, a.mid2.cout, dim);
    s.push_back({"mid_down2.conv.weight", (int64_t)a.mid2.cout * a.mid2.cout * 3});
    s.push_back({"mid_down2.conv.bias", a.mid2.cout});
    int fc = a.fc_dim;
    s.push_back({"final_block.0.weight", (int64_t)(fc / 2) * (fc + dim)});
    s.push_back({"final_block.0.bias", fc / 2});
    s.push_back({"final_block.2.weight", fc / 2});
    s.push_back({"final_block.2.bias", 1});
  }
  return s;
}

int check_desc(const trebi_net_desc* d) {
  if (!d) return fail(TREBI_BAD_ARG, "null descriptor");
  if (d->kind != 0 && d->kind != 1) return fail(TREBI_BAD_ARG, "kind must be 0 (TemporalUnet) or 1 (ValueFunction)");
  if (d->n_mults < 1 || d->n_mults > 8) return fail(TREBI_BAD_ARG, "n_mults must be in 1..8");
  if (d->horizon < 1 || d->transition_dim < 1 || d->dim < 8 || d->max_timesteps < 1)
    return fail(TREBI_BAD_ARG, "horizon, transition_dim, max_timesteps must be >= 1 and dim >= 8");
  if (d->dim % 32)
    return fail(TREBI_UNSUPPORTED_CONFIG, "dim must be a multiple of 32 (GroupNorm(8) groups are processed as float4s)");
  for (int i = 0; i < d->n_mults; ++i)
    if (d->dim_mults[i] < 1) return fail(TREBI_BAD_ARG, "dim_mults must be positive");
  return 0;
}

// ------------------------------------------------------------------------------------
// builder: uploads packed weights and emits ops
// ------------------------------------------------------------------------------------
struct Builder {
  trebi_net* net;
  std::unordered_map<std::string, std::pair<const float*, int64_t>> sd;
  int dim, T;
  const float* temb = nullptr;      // device [T, dim]
  bool train = false;               // training build: keep what the backward needs, record uploads and time jobs

  const float* host(const std::string& name, int64_t numel) {
    auto it = sd.find(name);
    if (it == sd.end()) throw BadShape{"missing tensor '" + name + "'"};
    if (it->second.second != numel)
      throw BadShape{"tensor '" + name + "' has " + std::to_string(it->second.second) + " elements, expected " +
                     std::to_string(numel)};
    return it->second.first;
  }
  // bump allocation out of a few large device chunks: packed weights of consecutive layers sit next
  // to each other (fewer TLB misses for kernels that each touch a different layer)
  char* chunk = nullptr;
  size_t chunk_left = 0;
  float* dalloc(size_t n) {
    size_t bytes = (std::max<size_t>(n, 1) * sizeof(float) + 255) / 256 * 256;
    if (bytes > chunk_left) {
      size_t sz = std::max<size_t>(bytes, (size_t)16 << 20);
      void* p = nullptr;
      if (hipMalloc(&p, sz) != hipSuccess) throw BadShape{"hipMalloc failed"};
      net->allocs.push_back(p);
      net->alloc_bytes.push_back(sz);
      chunk = (char*)p;
      chunk_left = sz;
    }
    float* r = (float*)chunk;
    chunk += bytes;
    chunk_left -= bytes;
    return r;
  }
  const float* upload(const float* h, size_t n) {
    float* d = dalloc(n);
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) throw BadShape{"hipMemcpy failed"};
    if (train) net->uploads.push_back({d, n});
    return d;
  }
  const float* upload(const std::vector<float>& v) { return upload(v.data(), v.size()); }

  // Weff is [ntaps][Cin][N] (+ optional second source [Cin2][N]); pack to the kernel's
  // [(ntaps*Cin_p + Cin2_p)/16][4][Npad][4] order (pure permutation + zero fill).
  const float* pack(const std::vector<float>& Weff, int ntaps, int Cin, int N, const std::vector<float>* W2,
                    int Cin2) {
    int Cin_p = rup(Cin, 16), Cin2_p = W2 ? rup(Cin2, 16) : 0, Npad = rup(N, 16);
    int KK = (ntaps * Cin_p + Cin2_p) / 16;
    // + 8 zero steps of slack: the chain kernels request whole 8-step blocks without bounds checks
    std::vector<float> P((size_t)(KK + 8) * 16 * Npad, 0.f);
    auto put = [&](int kk, int cc, int n, float v) {   // cc in [0,16)
      int g = cc >> 2, j = cc & 3;
      P[(((size_t)kk * 4 + g) * Npad + n) * 4 + j] = v;
    };
    for (int tp = 0; tp < ntaps; ++tp)
      for (int ci = 0; ci < Cin; ++ci)
        for (int n = 0; n < N; ++n)
          put(tp * (Cin_p / 16) + ci / 16, ci % 16, n, Weff[((size_t)tp * Cin + ci) * N + n]);
    if (W2)
      for (int ci = 0; ci < Cin2; ++ci)
        for (int n = 0; n < N; ++n)
          put(ntaps * (Cin_p / 16) + ci / 16, ci % 16, n, (*W2)[(size_t)ci * N + n]);
    return upload(P);
  }

  static void tap_range(int Lin, int Lout, int stride, int pad, int k, int* lo, int* hi) {
    *lo = k;
    *hi = -1;
    for (int tp = 0; tp < k; ++tp) {
      bool any = false;
      for (int o = 0; o < Lout && !any; ++o) {
        int li = o * stride + tp - pad;
        any = li >= 0 && li < Lin;
      }
      if (any) {
        *lo = std::min(*lo, tp);
        *hi = std::max(*hi, tp);
      }
    }
  }
  static double valid_pairs(int Lin, int Lout, int stride, int pad, int tlo, int nt) {
    double c = 0;
    for (int tp = tlo; tp < tlo + nt; ++tp)
      for (int o = 0; o < Lout; ++o) {
        int li = o * stride + tp - pad;
        if (li >= 0 && li < Lin) c += 1;
      }
    return c;
  }

  GnW gn_weights(const std::string& p, int C) {
    GnW g;
    g.gamma = upload(host(p + ".weight", C), C);
    g.beta = upload(host(p + ".bias", C), C);
    g.gch = C / GN_GROUPS;
    return g;
  }

  const float* time_table(const std::string& wname, const std::string& bname, int N, int ldw, int col0,
                          bool mish) {
    const float* w = upload(host(wname, (int64_t)N * ldw), (size_t)N * ldw);
    const float* b = upload(host(bname, N), N);
    float* out = dalloc((size_t)T * N);
    if (launch_time_linear(w, ldw, col0, b, temb, dim, T, N, mish ? 1 : 0, out, 0) != hipSuccess)
      throw BadShape{"time table kernel launch failed"};
    if (train) net->time_jobs.push_back({1, w, b, nullptr, nullptr, ldw, col0, N, mish ? 1 : 0, out});
    return out;
  }

  void finish_geom(Op& op, int Cin, int Cin2, int N) {
    op.Cin = Cin;
    op.Cin_p = rup(Cin, 16);
    op.Cin2 = Cin2;
    op.Cin2_p = Cin2 ? rup(Cin2, 16) : 0;
    op.N = N;
    op.Npad = rup(N, 16);
  }

  // ---- forward convolutions ----------------------------------------------------------
  // Conv1d(Cin, Cout, k, stride, pad) from a reference [Cout, Cin, k] tensor.
  Op conv_fwd(const std::string& wname, int Cin, int Cout, int k, int stride, int pad, int Lin,
              const std::string* res_wname, int Cres) {
    Op op;
    op.name = wname;
    op.Lin = Lin;
    op.Lout = (Lin + 2 * pad - k) / stride + 1;
    op.stride = stride;
    op.pad = pad;
    int lo, hi;
    tap_range(Lin, op.Lout, stride, pad, k, &lo, &hi);
    op.tap_lo = lo;
    op.ntaps = hi - lo + 1;
    const float* w = host(wname + ".weight", (int64_t)Cout * Cin * k);
    std::vector<float> W((size_t)op.ntaps * Cin * Cout);
    for (int tp = 0; tp < op.ntaps; ++tp)
      for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
          W[((size_t)tp * Cin + ci) * Cout + co] = w[((size_t)co * Cin + ci) * k + lo + tp];
    std::vector<float> W2;
    if (res_wname) {
      const float* r = host(*res_wname + ".weight", (int64_t)Cout * Cres);
      W2.resize((size_t)Cres * Cout);
      for (int ci = 0; ci < Cres; ++ci)
        for (int co = 0; co < Cout; ++co) W2[(size_t)ci * Cout + co] = r[(size_t)co * Cres + ci];
      op.bias2 = upload(host(*res_wname + ".bias", Cout), Cout);
    }
    finish_geom(op, Cin, res_wname ? Cres : 0, Cout);
    op.w = pack(W, op.ntaps, Cin, Cout, res_wname ? &W2 : nullptr, Cres);
    op.bias = upload(host(wname + ".bias", Cout), Cout);
    op.flops = 2.0 * Cin * Cout * valid_pairs(Lin, op.Lout, stride, pad, lo, op.ntaps) +
               (res_wname ? 2.0 * Cres * Cout * op.Lout : 0.0);
    return op;
  }

  // ConvTranspose1d(C, C, 4, 2, 1) [Cin, Cout, 4] as a k=3 / pad=1 convolution producing the two
  // output phases side by side: out[2j]   = x[j-1] W[..3] + x[j] W[..1]
  //                             out[2j+1] = x[j]   W[..2] + x[j+1] W[..0]
  Op convT_fwd(const std::string& wname, int C, int Lin) {
    Op op;
    op.name = wname;
    op.Lin = Lin;
    op.Lout = Lin;
    op.stride = 1;
    op.pad = 1;
    int lo, hi;
    tap_range(Lin, Lin, 1, 1, 3, &lo, &hi);
    op.tap_lo = lo;
    op.ntaps = hi - lo + 1;
    const float* w = host(wname + ".weight", (int64_t)C * C * 4);
    const float* b = host(wname + ".bias", C);
    int N = 2 * C;
    std::vector<float> W((size_t)op.ntaps * C * N, 0.f);
    const int ktab[3][2] = {{3, -1}, {1, 2}, {-1, 0}};      // [tap][phase] -> transposed-conv tap
    double pairs = 0;
    for (int tp = 0; tp < op.ntaps; ++tp)
      for (int ph = 0; ph < 2; ++ph) {
        int kt = ktab[lo + tp][ph];
        if (kt < 0) continue;
        for (int o = 0; o < Lin; ++o) {
          int li = o + lo + tp - 1;
          if (li >= 0 && li < Lin) pairs += 1;
        }
        for (int ci = 0; ci < C; ++ci)
          for (int co = 0; co < C; ++co)
            W[((size_t)tp * C + ci) * N + ph * C + co] = w[((size_t)ci * C + co) * 4 + kt];
      }
    std::vector<float> bb(N);
    for (int i = 0; i < N; ++i) bb[i] = b[i % C];
    finish_geom(op, C, 0, N);
    op.osplit = C;
    op.w = pack(W, op.ntaps, C, N, nullptr, 0);
    op.bias = upload(bb);
    op.flops = 2.0 * C * C * pairs;
    return op;
  }

  // ---- input-gradient convolutions (ValueFunction backward) ---------------------------
  // dX = conv(dY, flip(W)^T): K = (tap', Cout), N = Cin.  Stride-1, pad = k/2.
  Op conv_dgrad(const std::string& wname, int Cin, int Cout, int k, int L, const std::string* res_wname) {
    Op op;
    op.name = wname + " (dgrad)";
    op.Lin = L;
    op.Lout = L;
    op.stride = 1;
    op.pad = k / 2;
    int lo, hi;
    tap_range(L, L, 1, k / 2, k, &lo, &hi);
    op.tap_lo = lo;
    op.ntaps = hi - lo + 1;
    const float* w = host(wname + ".weight", (int64_t)Cout * Cin * k);
    std::vector<float> W((size_t)op.ntaps * Cout * Cin);
    for (int tp = 0; tp < op.ntaps; ++tp)
      for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
          W[((size_t)tp * Cout + co) * Cin + ci] = w[((size_t)co * Cin + ci) * k + (k - 1 - (lo + tp))];
    std::vector<float> W2;
    int C2 = 0;
    if (res_wname) {                  // residual 1x1 conv [Cout2, Cin, 1]: dX += dOut @ Wr
      auto it = sd.find(*res_wname + ".weight");
      if (it == sd.end()) throw BadShape{"missing tensor '" + *res_wname + ".weight'"};
      C2 = (int)(it->second.second / Cin);
      const float* r = host(*res_wname + ".weight", (int64_t)C2 * Cin);
      W2.resize((size_t)C2 * Cin);
      for (int co = 0; co < C2; ++co)
        for (int ci = 0; ci < Cin; ++ci) W2[(size_t)co * Cin + ci] = r[(size_t)co * Cin + ci];
    }
    finish_geom(op, Cout, C2, Cin);
    op.w = pack(W, op.ntaps, Cout, Cin, res_wname ? &W2 : nullptr, C2);
    op.flops = 2.0 * Cin * Cout * valid_pairs(L, L, 1, k / 2, lo, op.ntaps) + 2.0 * C2 * Cin * L;
    return op;
  }

  // Backward-to-input of Downsample1d = Conv1d(C, C, 3, 2, 1): with Lin even,
  //   dX[2j]   = dY[j] W[..1]^T,   dX[2j+1] = dY[j] W[..2]^T + dY[j+1] W[..0]^T
  // i.e. a 2-tap convolution over dY producing both phases side by side; Lin == 1 is a 1x1.
  Op down_dgrad(const std::string& wname, int C, int Lin_fwd) {
    Op op;
    op.name = wname + " (dgrad)";
    int Lo = down_len(Lin_fwd);
    const float* w = host(wname + ".weight", (int64_t)C * C * 3);
    if (Lin_fwd == 1) {
      op.Lin = op.Lout = 1;
      op.stride = 1;
      op.pad = 0;
      op.tap_lo = 0;
      op.ntaps = 1;
      std::vector<float> W((size_t)C * C);
      for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < C; ++ci) W[(size_t)co * C + ci] = w[((size_t)co * C + ci) * 3 + 1];
      finish_geom(op, C, 0, C);
      op.w = pack(W, 1, C, C, nullptr, 0);
      op.flops = 2.0 * C * C;
      return op;
    }
    if (Lin_fwd % 2) unsupported("Downsample1d backward needs an even (or unit) length, got %d", Lin_fwd);
    op.Lin = op.Lout = Lo;
    op.stride = 1;
    op.pad = 0;
    int lo, hi;
    tap_range(Lo, Lo, 1, 0, 2, &lo, &hi);
    op.tap_lo = lo;
    op.ntaps = hi - lo + 1;
    int N = 2 * C;
    std::vector<float> W((size_t)op.ntaps * C * N, 0.f);
    const int ktab[2][2] = {{1, 2}, {-1, 0}};           // [tap][phase] -> forward tap
    double pairs = 0;
    for (int tp = 0; tp < op.ntaps; ++tp)
      for (int ph = 0; ph < 2; ++ph) {
        int kt = ktab[lo + tp][ph];
        if (kt < 0) continue;
        for (int o = 0; o < Lo; ++o)
          if (o + lo + tp < Lo) pairs += 1;
        for (int co = 0; co < C; ++co)
          for (int ci = 0; ci < C; ++ci)
            W[((size_t)tp * C + co) * N + ph * C + ci] = w[((size_t)co * C + ci) * 3 + kt];
      }
    finish_geom(op, C, 0, N);
    op.osplit = C;
    op.w = pack(W, op.ntaps, C, N, nullptr, 0);
    op.flops = 2.0 * C * C * pairs;
    return op;
  }

  // Backward-to-input of Upsample1d = ConvTranspose1d(C, C, 4, 2, 1) [Cin, Cout, 4]: y[co, 2j+k-1] += x[ci, j] w[ci, co, k],
  // so dX[ci, j] = sum_{co, k} dY[co, 2j + k - 1] w[ci, co, k]: a Conv1d(k=4, stride 2, pad 1) over dY (length 2*Lin).
  Op convT_dgrad(const std::string& wname, int C, int Lin_fwd) {
    Op op;
    op.name = wname + " (dgrad)";
    op.Lin = 2 * Lin_fwd;
    op.Lout = Lin_fwd;
    op.stride = 2;
    op.pad = 1;
    int lo, hi;
    tap_range(op.Lin, op.Lout, 2, 1, 4, &lo, &hi);
    op.tap_lo = lo;
    op.ntaps = hi - lo + 1;
    const float* w = host(wname + ".weight", (int64_t)C * C * 4);
    std::vector<float> W((size_t)op.ntaps * C * C);
    for (int tp = 0; tp < op.ntaps; ++tp)
      for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < C; ++ci) W[((size_t)tp * C + co) * C + ci] = w[((size_t)ci * C + co) * 4 + lo + tp];
    finish_geom(op, C, 0, C);
    op.w = pack(W, op.ntaps, C, C, nullptr, 0);
    op.flops = 2.0 * C * C * valid_pairs(op.Lin, op.Lout, 2, 1, lo, op.ntaps);
    return op;
  }

  // ---- ResidualTemporalBlock (temporal.py:18-46) -----------------------------------------
  struct RBRec {
    std::string prefix;
    int cin, cout, L;
    bool has_res;
    int y0, st0, y1, st1, g0, g1;
    GnW gn0, gn1;
    RBInfo info;
  };

  RBRec res_block(const std::string& p, BufRef src, int cin, int cout, int L, BufRef dst, bool save) {
    RBRec rec{p, cin, cout, L, cin != cout, NONE, NONE, NONE, NONE, NONE, NONE, {}, {}, {}};
    save = save || train;
    std::string b0 = p + ".blocks.0.block", b1 = p + ".blocks.1.block";
    Op o0 = conv_fwd(b0 + ".0", cin, cout, 5, 1, 2, L, nullptr, 0);
    o0.gn = true;
    o0.gnw = rec.gn0 = gn_weights(b0 + ".2", cout);
    o0.tbias = time_table(p + ".time_mlp.1.weight", p + ".time_mlp.1.bias", cout, dim, 0, true);
    o0.src = src;
    int a0 = net->new_buf((size_t)L * cout);
    o0.dst = {a0, cout, 0};
    std::string rname = p + ".residual_conv";
    Op o1 = conv_fwd(b1 + ".0", cout, cout, 5, 1, 2, L, rec.has_res ? &rname : nullptr, cin);
    o1.gn = true;
    o1.gnw = rec.gn1 = gn_weights(b1 + ".2", cout);
    o1.src = {a0, cout, 0};
    if (rec.has_res) o1.src2 = src; else o1.res = src;
    o1.dst = dst;
    if (save) {
      o0.save_y = rec.y0 = net->new_buf((size_t)L * cout);
      o0.save_st = rec.st0 = net->new_buf(2 * GN_GROUPS);
      o1.save_y = rec.y1 = net->new_buf((size_t)L * cout);
      o1.save_st = rec.st1 = net->new_buf(2 * GN_GROUPS);
      if (!train) {                  // mish'(z)*gamma feeds the fused backward prologue; training recomputes it
        o0.save_g = rec.g0 = net->new_buf((size_t)L * cout);
        o1.save_g = rec.g1 = net->new_buf((size_t)L * cout);
      }
    }
    net->fwd.push_back(o0);
    net->fwd.push_back(o1);
    RBInfo& f = rec.info;
    f.prefix = p; f.cin = cin; f.cout = cout; f.L = L; f.has_res = rec.has_res; f.src = src; f.dst = dst; f.a0 = a0;
    f.y0 = rec.y0; f.st0 = rec.st0; f.y1 = rec.y1; f.st1 = rec.st1; f.gn0 = rec.gn0; f.gn1 = rec.gn1;
    return rec;
  }

  // backward-to-input of a block: g = dL/d(out) [L, cout] -> dst = dL/d(in) [L, cin]
  void res_block_bwd(const RBRec& r, BufRef g, BufRef dst) {
    std::string b0 = r.prefix + ".blocks.0.block", b1 = r.prefix + ".blocks.1.block";
    Op kb = conv_dgrad(b1 + ".0", r.cout, r.cout, 5, r.L, nullptr);
    kb.src = g;
    kb.pro = true;
    kb.pro_gn = r.gn1;
    kb.pro_y = r.y1;
    kb.pro_g = r.g1;
    kb.pro_st = r.st1;
    int da0 = net->new_buf((size_t)r.L * r.cout);
    kb.dst = {da0, r.cout, 0};
    std::string rname = r.prefix + ".residual_conv";
    Op kc = conv_dgrad(b0 + ".0", r.cin, r.cout, 5, r.L, r.has_res ? &rname : nullptr);
    kc.src = {da0, r.cout, 0};
    kc.pro = true;
    kc.pro_gn = r.gn0;
    kc.pro_y = r.y0;
    kc.pro_g = r.g0;
    kc.pro_st = r.st0;
    if (r.has_res) kc.src2 = g; else kc.res = g;
    kc.dst = dst;
    net->bwd.push_back(kb);
    net->bwd.push_back(kc);
  }
};

void build_unet_train_bwd(Builder& B, const ArchPlan& a);     // trainer.inl

void build_unet(Builder& B, const ArchPlan& a) {
  trebi_net* net = B.net;
  const trebi_net_desc& d = net->desc;
  int n = d.n_mults;
  auto S = [](int i) { return std::to_string(i); };
  std::vector<int> C(n), L(n), cat(n, NONE);
  for (int i = 0; i < n; ++i) {
    C[i] = a.downs[i].cout;
    L[i] = a.downs[i].L;
    if (i >= 1) cat[i] = net->new_buf((size_t)L[i] * 2 * C[i]);
  }
  UnetRec& rec = net->rec;
  BufRef cur{EXT_X, 0, 0};
  for (int i = 0; i < n; ++i) {
    int rA = net->new_buf((size_t)L[i] * C[i]);
    rec.down0.push_back(B.res_block("downs." + S(i) + ".0", cur, a.downs[i].cin, C[i], L[i], {rA, C[i], 0}, false).info);
    BufRef skip = i >= 1 ? BufRef{cat[i], 2 * C[i], C[i]} : BufRef{net->new_buf((size_t)L[i] * C[i]), C[i], 0};
    rec.down1.push_back(B.res_block("downs." + S(i) + ".1", {rA, C[i], 0}, C[i], C[i], L[i], skip, false).info);
    cur = skip;
    rec.downs.push_back({});
    if (a.downs[i].down) {
      Op dn = B.conv_fwd("downs." + S(i) + ".3.conv", C[i], C[i], 3, 2, 1, L[i], nullptr, 0);
      dn.src = cur;
      int db = net->new_buf((size_t)dn.Lout * C[i]);
      dn.dst = {db, C[i], 0};
      net->fwd.push_back(dn);
      rec.downs.back() = {"downs." + S(i) + ".3.conv", C[i], L[i], cur, dn.dst};
      cur = {db, C[i], 0};
    }
  }
  int Lm = a.mid_L, Cm = a.mid_c;
  int m1 = net->new_buf((size_t)Lm * Cm);
  rec.mid1 = B.res_block("mid_block1", cur, Cm, Cm, Lm, {m1, Cm, 0}, false).info;
  if (n == 1) {
    int m2 = net->new_buf((size_t)Lm * Cm);
    rec.mid2 = B.res_block("mid_block2", {m1, Cm, 0}, Cm, Cm, Lm, {m2, Cm, 0}, false).info;
    cur = {m2, Cm, 0};
  } else {
    rec.mid2 = B.res_block("mid_block2", {m1, Cm, 0}, Cm, Cm, Lm, {cat[n - 1], 2 * C[n - 1], 0}, false).info;
  }
  for (int k = 0; k < n - 1; ++k) {
    int j = n - 1 - k;
    const auto& u = a.ups[k];
    int Lj = u.L;
    if (Lj != L[j]) unsupported("TemporalUnet skip length mismatch at level %d (%d vs %d)", j, Lj, L[j]);
    int u0 = net->new_buf((size_t)Lj * u.cout);
    rec.up0.push_back(B.res_block("ups." + S(k) + ".0", {cat[j], 2 * C[j], 0}, u.cin, u.cout, Lj, {u0, u.cout, 0}, false).info);
    rec.catC.push_back(C[j]);
    rec.ups.push_back({});
    int Lnext = u.up ? 2 * Lj : Lj;
    if (j - 1 >= 0 && Lnext != L[j - 1])
      unsupported("TemporalUnet up-path length %d does not match skip length %d", Lnext, L[j - 1]);
    BufRef nextdst = (j - 1 >= 1) ? BufRef{cat[j - 1], 2 * C[j - 1], 0}
                                  : BufRef{net->new_buf((size_t)Lnext * u.cout), u.cout, 0};
    if (u.up) {
      int u1 = net->new_buf((size_t)Lj * u.cout);
      rec.up1.push_back(B.res_block("ups." + S(k) + ".1", {u0, u.cout, 0}, u.cout, u.cout, Lj, {u1, u.cout, 0}, false).info);
      Op up = B.convT_fwd("ups." + S(k) + ".3.conv", u.cout, Lj);
      up.src = {u1, u.cout, 0};
      up.dst = nextdst;
      net->fwd.push_back(up);
      rec.ups.back() = {"ups." + S(k) + ".3.conv", u.cout, Lj, up.src, nextdst};
    } else {
      rec.up1.push_back(B.res_block("ups." + S(k) + ".1", {u0, u.cout, 0}, u.cout, u.cout, Lj, nextdst, false).info);
    }
    cur = nextdst;
  }
  // final_conv: Conv1dBlock(dim, dim, 5) then Conv1d(dim, out, 1)   (temporal.py:115-118)
  int Lf = a.final_L, dim = d.dim;
  Op f0 = B.conv_fwd("final_conv.0.block.0", dim, dim, 5, 1, 2, Lf, nullptr, 0);
  f0.gn = true;
  f0.gnw = B.gn_weights("final_conv.0.block.2", dim);
  f0.src = cur;
  int fb = net->new_buf((size_t)Lf * dim);
  f0.dst = {fb, dim, 0};
  rec.final_src = cur; rec.final_fb = fb; rec.final_L = Lf; rec.final_gn = f0.gnw;
  if (B.train) {
    f0.save_y = rec.final_y = net->new_buf((size_t)Lf * dim);
    f0.save_st = rec.final_st = net->new_buf(2 * GN_GROUPS);
  }
  net->fwd.push_back(f0);
  Op f1 = B.conv_fwd("final_conv.1", dim, d.transition_dim, 1, 1, 0, Lf, nullptr, 0);
  f1.src = {fb, dim, 0};
  f1.dst = {EXT_OUT, 0, 0};
  net->fwd.push_back(f1);
  if (Lf != d.horizon) unsupported("TemporalUnet output length %d != horizon %d", Lf, d.horizon);
  if (B.train) build_unet_train_bwd(B, a);
}

void build_value(Builder& B, const ArchPlan& a) {
  trebi_net* net = B.net;
  const trebi_net_desc& d = net->desc;
  int n = d.n_mults;
  auto S = [](int i) { return std::to_string(i); };
  if (a.vfinal_L != 1 || a.fc_dim != a.c3)
    unsupported("ValueFunction(horizon=%d) ends at length %d with fc_dim %d != %d: the reference's own head "
                "Linear does not fit this shape (temporal.py:190-205)", d.horizon, a.vfinal_L, a.fc_dim, a.c3);
  struct Stage {
    Builder::RBRec rb[2];
    int nrb;
    std::string down;
    int C, L;
    BufRef in, rb0_out, rb_out, down_out;
  };
  std::vector<Stage> st;
  BufRef cur{EXT_X, 0, 0};
  auto stage = [&](const std::string& p0, const std::string* p1, const std::string& down, int cin, int cout, int L) {
    Stage s;
    s.nrb = p1 ? 2 : 1;
    s.C = cout;
    s.L = L;
    s.in = cur;
    s.down = down;
    int r0 = net->new_buf((size_t)L * cout);
    s.rb0_out = {r0, cout, 0};
    s.rb[0] = B.res_block(p0, cur, cin, cout, L, s.rb0_out, true);
    s.rb_out = s.rb0_out;
    if (p1) {
      int r1 = net->new_buf((size_t)L * cout);
      s.rb_out = {r1, cout, 0};
      s.rb[1] = B.res_block(*p1, s.rb0_out, cout, cout, L, s.rb_out, true);
    }
    Op dn = B.conv_fwd(down + ".conv", cout, cout, 3, 2, 1, L, nullptr, 0);
    dn.src = s.rb_out;
    int db = net->new_buf((size_t)dn.Lout * cout);
    dn.dst = s.down_out = {db, cout, 0};
    net->fwd.push_back(dn);
    cur = s.down_out;
    st.push_back(s);
  };
  for (int i = 0; i < n; ++i) {
    std::string p0 = "blocks." + S(i) + ".0", p1 = "blocks." + S(i) + ".1";
    stage(p0, &p1, "blocks." + S(i) + ".2", a.blocks[i].cin, a.blocks[i].cout, a.blocks[i].L);
  }
  stage("mid_block1", nullptr, "mid_down1", a.mid1.cin, a.mid1.cout, a.mid1.L);
  stage("mid_block2", nullptr, "mid_down2", a.mid2.cin, a.mid2.cout, a.mid2.L);

  // head (temporal.py:207-211, 237-238)
  int fc = a.fc_dim, hd = fc / 2, dim = d.dim;
  net->fc = fc;
  net->hd = hd;
  const float* w1 = B.host("final_block.0.weight", (int64_t)hd * (fc + dim));
  std::vector<float> w1a((size_t)hd * fc);
  for (int j = 0; j < hd; ++j)
    for (int i = 0; i < fc; ++i) w1a[(size_t)j * fc + i] = w1[(size_t)j * (fc + dim) + i];
  net->head_w1 = B.upload(w1a);
  net->head_tb = B.time_table("final_block.0.weight", "final_block.0.bias", hd, fc + dim, fc, false);
  net->head_w2 = B.upload(B.host("final_block.2.weight", hd), hd);
  net->head_b2 = B.host("final_block.2.bias", 1)[0];
  net->head_b2_dev = B.upload(std::vector<float>((size_t)hd, net->head_b2));
  Op head;
  head.kind = 1;
  head.name = "final_block";
  head.v_buf = cur.id;
  head.dv_buf = net->new_buf(fc);
  // (as a chain stage -- plan_chain -- it is a layer from [1, fc] to [1, fc]: the flattened mid_down2 output in, dL/dv out)
  head.Lin = head.Lout = 1; head.Cin = head.Cin_p = fc; head.N = head.Npad = fc;
  head.src = cur; head.dst = {head.dv_buf, fc, 0};
  head.flops = 2.0 * hd * fc + 2.0 * hd;
  net->fwd.push_back(head);

  // backward chain
  BufRef g{head.dv_buf, fc, 0};
  for (int si = (int)st.size() - 1; si >= 0; --si) {
    Stage& s = st[si];
    Op dd = B.down_dgrad(s.down + ".conv", s.C, s.L);
    dd.src = g;
    int gb = net->new_buf((size_t)s.L * s.C);
    dd.dst = {gb, s.C, 0};
    net->bwd.push_back(dd);
    g = {gb, s.C, 0};
    for (int r = s.nrb - 1; r >= 0; --r) {
      const auto& rec = s.rb[r];
      BufRef dst;
      if (si == 0 && r == 0) dst = {EXT_GRAD, 0, 0};
      else dst = {net->new_buf((size_t)rec.L * rec.cin), rec.cin, 0};
      B.res_block_bwd(rec, g, dst);
      g = dst;
    }
  }
}

// ------------------------------------------------------------------------------------
// tile selection + op resolution
// ------------------------------------------------------------------------------------
int gcd(int a, int b) { return b ? gcd(b, a % b) : a; }

// Pick the tile of one layer for a batch of B candidates in nz nets per launch.  The kernel needs ~200
// VGPRs, so one 512-thread workgroup is resident per compute unit: a launch takes ceil(workgroups / 256)
// rounds of one workgroup lifetime each.  Enumerate the shapes the kernel is instantiated for and take the
// cheapest by a small cost model (fixed per-workgroup latency + matrix-core time per SIMD + staging).
Tile choose_tile(const Op& op, int B, int nz = 1, bool bf = false) {
  int Lout = op.Lout;
  if (Lout > 128) unsupported("sequence length %d > 128 is not tiled", Lout);
  auto mi_for = [&](int s) {
    int rows = s * Lout, m = (rows + 15) / 16;
    return m <= 1 ? 1 : m <= 2 ? 2 : m <= 4 ? 4 : 8;
  };
  const int units = op.Npad / 16;
  int tnu = 1;                                      // narrowest tile in 16-column units
  if (op.gn) {
    int l = op.gnw.gch / gcd(op.gnw.gch, 16) * 16;  // lcm(gch, 16): whole GroupNorm groups per tile
    tnu = l / 16;
    if (op.N % GN_GROUPS || op.Npad != op.N || units % tnu || op.gnw.gch % 4)
      unsupported("GroupNorm(8) over %d channels does not tile (channels must be a multiple of 32)", op.N);
  }
  if (op.osplit && op.osplit % (16 * tnu)) unsupported("phase split %d not a multiple of the tile", op.osplit);
  const int KK = (op.ntaps * op.Cin_p + op.Cin2_p) / 16;
  ConvP q{};
  q.Lout = op.Lout; q.stride = op.stride; q.ntaps = op.ntaps; q.Cin_p = op.Cin_p;
  // stand-in sources for the LDS carve (conv_make_geo asks conv_vec_ok): workspace buffers are float4-addressable, a
  // caller tensor (row stride = transition_dim) is taken not to be -- the carve then never assumes the W5-only late staging
  q.x = (const float*)16; q.ldx = op.src.id >= 0 ? 4 : 1; q.Cin = op.Cin;
  q.x2 = op.src2.id != NONE ? (const float*)16 : nullptr;
  q.ldx2 = op.src2.id >= 0 ? 4 : 1; q.Cin2 = op.Cin2;
  q.Cin2_p = op.Cin2_p; q.gn = op.gn; q.gch = op.gn ? op.gnw.gch : 1;
  q.Lin = op.Lin; q.pro = op.pro ? 1 : 0; q.tbias = op.tbias;
  q.res = op.res.id != NONE ? (const float*)1 : nullptr;
  q.tap_lo = op.tap_lo; q.pad = op.pad; q.B = B;
  const int R = (op.Lout - 1) * op.stride + op.ntaps;
  Tile best{};
  double best_cost = -1;
  std::vector<std::pair<double, Tile>> ranked;
  static const bool pm_on = !(getenv("TREBI_PM") && atoi(getenv("TREBI_PM")) == 0);
  static const int pm_min_b = getenv("TREBI_PM_MINB") ? atoi(getenv("TREBI_PM_MINB")) : 512;          // diagnostics: position-major tiles off
  const int S0 = Lout >= 16 ? 1 : 16 / Lout;
  for (int S = S0; S * Lout <= 128 && (S == S0 || S <= B); S *= 2) {
    const int MI = mi_for(S);
    for (int NI = 1; NI <= 2; ++NI) {
      if (MI == 8 && NI == 2) continue;             // not instantiated
      for (int NG = tnu; NG <= 8; NG += tnu) {
        const int tu = NI * NG;                     // tile width in 16-column units
        if (units % tu || tu % tnu) continue;
        if (op.osplit && op.osplit % (16 * tu)) continue;
        const int KS0 = std::max(1, std::min(8 / NG, std::max(1, KK / 4)));
        // smaller workgroups (less split-K) leave room for several per CU: at large batches the narrow layers' lifetime
        // is all latency, which co-resident workgroups overlap (~200 VGPRs: 8 waves per CU)
        static const bool occ_on = !(getenv("TREBI_OCC") && atoi(getenv("TREBI_OCC")) == 0);
        static const int occ_min_b = getenv("TREBI_OCC_MINB") ? atoi(getenv("TREBI_OCC_MINB")) : 512;
        for (int KS = KS0; KS >= 1; KS = (KS > 1 && occ_on && B >= occ_min_b) ? KS / 2 : 0)
        for (int pm = 0; pm <= 1; ++pm)
        for (int resg = 0; resg <= ((op.res.id != NONE && B >= 512) ? 1 : 0); ++resg) {
          Tile t{MI, NI, NG, KS, S};
          t.pm = pm;
          t.resg = resg;
          t.bf = bf;                                   // the LDS carve depends on it (conv_make_geo: x2late)
          double frac = 1.0;                          // share of (position, tap) pairs that touch real input rows
          int rows = R;
          static const char* dbg = getenv("TREBI_CHOOSER_DEBUG");          // diagnostics: layer-name substring
          const bool say = dbg && op.name.find(dbg) != std::string::npos;
          if (pm) {
            if (say) fprintf(stderr, "[chooser %s B=%d] pm candidate S=%d MI=%d NI=%d NG=%d KS=%d resg=%d Lout=%d\n", op.name.c_str(), B, S, MI, NI, NG, KS, resg, Lout);
            if (!pm_on || B < pm_min_b || S % 16 || S * Lout != 16 * MI) continue;
            const ConvGeo g0 = conv_make_geo(q, Tile{MI, NI, NG, KS, S});
            if (g0.hr == 0) continue;                 // no zero padding to skip
            int valid = 0;
            for (int lo = 0; lo < Lout; ++lo)
              for (int tau = 0; tau < op.ntaps; ++tau) {
                const int j = lo * op.stride + tau;
                valid += (j >= g0.fh && j < g0.fh + g0.nv);
              }
            const double k1 = (double)op.ntaps * op.Cin_p, k2 = op.Cin2_p;
            frac = (k1 * valid / (double)(Lout * op.ntaps) + k2) / (k1 + k2);
            rows = g0.nv;
          }
          const size_t lds = conv_lds_bytes(q, t);
          if (say) fprintf(stderr, "[chooser %s] S=%d MI=%d NI=%d NG=%d KS=%d pm=%d resg=%d lds=%zu\n", op.name.c_str(), S, MI, NI, NG, KS, pm, resg, lds);
          if (lds > 160 * 1024) continue;
          const long wgs = (long)((B + S - 1) / S) * (units / tu) * nz;
          // waves a CU holds: 2 per SIMD for the ~200-VGPR instantiations, 3 for the ones under 168 VGPRs
          // (hipcc -Rpass-analysis=kernel-resource-usage: (MI, NI) = (4,1), (1,2), (2,2) at 159-164)
          // (measured: +1-2 % at 8192 candidates, -1-2 % at 512-2048, where the launches do not fill the slots anyway)
          static const int occ3_min_b = getenv("TREBI_OCC3_MINB") ? atoi(getenv("TREBI_OCC3_MINB")) : 4096;
          // (the bf16x3 instantiations of the same shapes need 170-175: two waves per SIMD)
          const bool small_regs = !bf && B >= occ3_min_b && ((MI == 4 && NI == 1) || (MI <= 2 && NI == 2));
          const int cu_waves = small_regs ? 12 : 8;
          const int occ = (occ_on && B >= occ_min_b) ? std::max(1, std::min(cu_waves / (NG * KS), (int)(160 * 1024 / std::max<size_t>(lds, 1)))) : 1;
          const long rounds = (wgs + 256 * occ - 1) / (256 * occ);
          const int chunk = (KK + KS - 1) / KS;
          // position-major tiles issue each tile's 4 MFMAs back to back behind a branch: with one column tile per
          // wave (NI == 1) they form a dependent chain (40 instead of 32 cycles each) -- measured, not modelled finer
          // (the channel-block-major loop of conv_w5_ok has neither the chain nor the branches; what it pays for is its
          // weight stream, so the eight-tile one-column waves, with half the requests per MFMA, come out ~10 % ahead of
          // the four-tile ones, which run as fast as the tap-major loop did: measured per layer at 8192 candidates)
          const bool w5 = pm && conv_w5_ok(q, t) && conv_vec_ok(q);
          const double pmk = pm ? (NI == 1 ? (w5 ? (MI == 8 ? 0.92 : 1.05) : 1.25) : 1.05) : 1.0;
          const double mfma = 32.0 * 4 * MI * NI * chunk * frac * pmk;     // cycles of one wave's matrix work
          const double simd = mfma * std::max(1.0, occ * NG * KS / 4.0);   // waves of a SIMD share its matrix core
          const double nthr = 64.0 * NG * KS;
          const double stage = 30.0 * ((double)S * rows * op.Cin_p / 4 * (op.pro ? 3 : 1) +
                                       (double)S * Lout * op.Cin2_p / 4) / nthr;
          const double epi = 60.0 * (16.0 * MI * 16 * tu / 4) / nthr;
          const double wload = 8.0 * chunk * NI;                           // issue + latency share of the weight stream
          // the three-waves-per-SIMD instantiations come out 15-25 % ahead of what the terms above say on the narrow layers
          // (<= 64 output columns; per-layer times of the model's first four choices at 8192 candidates, tools/layer_profile.py
          // + TREBI_TILE_RANK; configs[3] 392.5 -> 384.3 ms, flat for a factor between 0.6 and 0.9; on the wide layers and
          // at 4096 candidates the same factor loses)
          static const double resg_cost = getenv("TREBI_RESG_COST") ? atof(getenv("TREBI_RESG_COST")) : 500.0;
          static const bool k3_all = getenv("TREBI_OCC3_ALL") && atoi(getenv("TREBI_OCC3_ALL")) == 1;
          static const double k3 = getenv("TREBI_OCC3_GAIN") ? atof(getenv("TREBI_OCC3_GAIN")) : 0.8;
          const double cost = rounds * (11000.0 + 1.4 * simd + stage + epi + wload + (resg ? resg_cost : 0.0)) * (small_regs && op.Npad <= 64 && B >= 8192 && (k3_all || units == tu) ? k3 : 1.0);
          if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = t; }
          ranked.push_back({cost, t});
        }
      }
    }
  }
  if (best_cost < 0) unsupported("layer %s needs more than 160 KiB of LDS", op.name.c_str());
  static const int rank = getenv("TREBI_TILE_RANK") ? atoi(getenv("TREBI_TILE_RANK")) : 0;   // diagnostics: r-th best tile
  if (rank > 0) {
    std::stable_sort(ranked.begin(), ranked.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    return ranked[std::min<size_t>(rank, ranked.size() - 1)].second;
  }
  return best;
}

// A layer runs with bf16x3 operands when its net is in that mode, has the copy, and the batch is large enough.
inline bool use_bf(const trebi_net* n, const Op& op, int B) {
  return n && n->prec == 1 && op.w16 != nullptr && B >= n->bf_min_b;
}

struct Resolver {
  const trebi_net* net;
  std::vector<size_t> off;
  Binding b;
  int B;
  const float* ptr(const BufRef& r, int* ld) const {
    if (r.id == NONE) { *ld = 0; return nullptr; }
    if (r.id == EXT_X) { *ld = b.ldx; return b.x; }
    if (r.id == EXT_OUT) { *ld = b.ldo; return b.out; }
    if (r.id == EXT_GRAD) { *ld = b.ldg; return b.grad; }
    *ld = r.ld;
    return b.ws + off[r.id] + r.coloff;
  }
  float* buf(int id) const { return id == NONE ? nullptr : b.ws + off[id]; }
  ConvP conv(const Op& op) const {
    ConvP p{};
    p.x = ptr(op.src, &p.ldx);
    p.Cin = op.Cin; p.Cin_p = op.Cin_p;
    p.Lin = op.Lin; p.Lout = op.Lout; p.stride = op.stride; p.pad = op.pad;
    p.tap_lo = op.tap_lo; p.ntaps = op.ntaps;
    p.x2 = ptr(op.src2, &p.ldx2);
    p.Cin2 = op.Cin2; p.Cin2_p = op.Cin2_p;
    p.w = op.w; p.bias = op.bias; p.bias2 = op.bias2;
    p.N = op.N; p.Npad = op.Npad;
    p.out = const_cast<float*>(ptr(op.dst, &p.ldo));
    p.osplit = op.osplit;
    p.gn = op.gn ? 1 : 0; p.gch = op.gn ? op.gnw.gch : 1;
    p.gamma = op.gnw.gamma; p.beta = op.gnw.beta;
    p.save_y = buf(op.save_y); p.save_g = buf(op.save_g); p.save_stats = buf(op.save_st);
    p.tbias = op.tbias; p.t = b.t; p.t_uniform = b.t_uniform; p.T = net->T;
    p.res = ptr(op.res, &p.ldres);
    p.pro = op.pro ? 1 : 0; p.pro_gch = op.pro ? op.pro_gn.gch : 1;
    p.pro_y = buf(op.pro_y); p.pro_g = buf(op.pro_g); p.pro_stats = buf(op.pro_st);
    p.B = B;
    return p;
  }
  HeadP head(const Op& op, bool want_grad) const {
    HeadP h{};
    h.v = buf(op.v_buf);
    h.w1 = net->head_w1; h.tb = net->head_tb; h.w2 = net->head_w2; h.b2 = net->head_b2;
    h.t = b.t; h.T = net->T; h.fc = net->fc; h.hd = net->hd; h.B = B;
    h.y = b.y;
    h.dv = want_grad ? buf(op.dv_buf) : nullptr;
    return h;
  }
};

// Profiling hook: when set, each GEMM launch is bracketed by HIP events on the launch stream.
struct GemmProfile {
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<double> flops;
  std::vector<std::string> names;  // layer + tile of each timed launch (TREBI_PROFILE_DUMP diagnostics)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> null_ev;   // empty brackets: the cost of the bracketing itself
  size_t used = 0;
  void reset() { used = 0; flops.clear(); names.clear(); }
  ~GemmProfile() {
    for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : null_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
};

// Debug aid: TREBI_POISON=1 fills every caller workspace with NaN before each network pass, so a
// kernel that reads an activation no earlier kernel of the pass wrote shows up as NaN in the result instead of silently
// re-using the previous call's data (tests/test_gpu_paths.py runs the golden loops that way).  Read once per process.
static bool poison_on() {
  static const bool on = getenv("TREBI_POISON") && atoi(getenv("TREBI_POISON")) != 0;
  return on;
}
static int poison_ws(float* ws, size_t floats, hipStream_t st) {
  for (size_t o = 0; o < floats; o += (size_t)1 << 30) {
    const size_t n = std::min(floats - o, (size_t)1 << 30);
    HIPCHK(launch_fill_f(ws + o, (int)n, std::nanf(""), st));
  }
  return 0;
}

// Run a list of ops for one net, or for two structurally identical nets in the same launches.
int run_ops(const std::vector<Op>& ops, const trebi_net* n0, const Binding& b0, const trebi_net* n1,
            const Binding* b1, const std::vector<Op>* ops1, int B, bool want_grad, hipStream_t st,
            GemmProfile* prof) {
  Resolver r0{n0, n0->offsets(B), b0, B};
  Resolver r1{n1 ? n1 : n0, n1 ? n1->offsets(B) : std::vector<size_t>(), n1 ? *b1 : b0, B};
  int nz = n1 ? 2 : 1;
  if (poison_on() && &ops == &n0->fwd) {           // once per pass: the backward reads what the forward saved
    if (int e = poison_ws(b0.ws, n0->ws_floats(B), st)) return e;
    if (n1)
      if (int e = poison_ws(b1->ws, n1->ws_floats(B), st)) return e;
  }
  for (size_t i = 0; i < ops.size(); ++i) {
    const Op& op = ops[i];
#ifdef TREBI_STAMPS
    if (const char* lim = getenv("TREBI_STOP_AFTER"))
      if ((int)i > atoi(lim)) break;
#endif
    if (op.kind == 1) {
      HeadLaunch L{};
      L.p[0] = r0.head(op, want_grad);
      if (n1) L.p[1] = r1.head((*ops1)[i], want_grad);
      HIPCHK(launch_head(L, nz, st));
      continue;
    }
    ConvLaunch L{};
    L.p[0] = r0.conv(op);
    if (n1) L.p[1] = r1.conv((*ops1)[i]);
    Tile tl;
    const bool bf = use_bf(n0, op, B) && conv_vec_ok(L.p[0]) && (!n1 || (use_bf(n1, (*ops1)[i], B) && conv_vec_ok(L.p[1])));
    try {
      tl = choose_tile(op, B, nz, bf);
    } catch (const Unsupported& u) {
      return fail(TREBI_UNSUPPORTED_CONFIG, "%s", u.msg.c_str());
    }
    // shapes the kernel and its grid assume, checked on the host before every launch
    if (tl.S * op.Lout > 16 * tl.MI || op.Npad % (16 * tl.NI * tl.NG) || 64 * tl.NG * tl.KS > 512 ||
        (op.pro && (op.stride != 1 || op.Lin != op.Lout)) || !L.p[0].x || !L.p[0].out)
      return fail(TREBI_UNSUPPORTED_CONFIG, "internal tiling error at layer %s", op.name.c_str());
    tl.bf = 0;
    if (bf) {
      tl.bf = 1;
      L.p[0].w = op.w16;
      if (n1) L.p[1].w = (*ops1)[i].w16;
    }
    if (prof && prof->enabled) {
      if (prof->used == prof->ev.size()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        prof->ev.push_back({a, b});
      }
      HIPCHK(hipEventRecord(prof->ev[prof->used].first, st));
      HIPCHK(launch_conv(L, nz, tl, st));
      HIPCHK(hipEventRecord(prof->ev[prof->used].second, st));
      prof->flops.push_back(op.flops * B * nz);
      prof->used++;
    } else {
      HIPCHK(launch_conv(L, nz, tl, st));
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------
// launch plans: the op list of one network pass resolved to launches for a fixed batch and fixed
// buffers.  Runs of layers that keep all output columns in one workgroup are folded into ONE chained
// launch (conv_chain_kernel); everything else is a single-layer launch.
// ------------------------------------------------------------------------------------
struct PlanStep {
  int kind = 0;                    // 0 single conv, 1 value head, 2 chain
  ConvLaunch L{};
  Tile tl{};
  HeadLaunch H{};
  int nz = 1;
  int chain_off = 0, chain_n = 0, nblocks = 0, chain_direct = 0, ny = 1;
  size_t chain_lds = 0;
  size_t par_off = 0;              // direct chains: float offset of the parameter table in the sampler's table arena
  int parN = 0, parT = 0, oTab = 0, has_head = 0, feat = 0;
  double flops = 0;                // algorithmic, whole batch, all nets of the launch
  std::string name;
};
typedef std::vector<PlanStep> Plan;

// Per-column parameter tables of the direct chains of a set of plans (common.h, launch_chain): what to fill and where.
struct ParTab {
  struct Fill { ConvP p; size_t off; int parN; size_t t_stride; };
  std::vector<Fill> fills;
  size_t floats = 0;
  // reserve [nz][T][nstages][5][parN] for a chain and queue the fills of its stages
  size_t add(const std::vector<RStage>& run, int nz, int parN, int T) {
    const size_t base = floats, n = run.size(), t_stride = n * 5 * (size_t)parN;
    for (int z = 0; z < nz; ++z)
      for (size_t k = 0; k < n; ++k)
        fills.push_back({run[k].p[z], base + ((size_t)z * T * n + k) * 5 * parN, parN, t_stride});
    floats += (size_t)nz * T * t_stride;
    return base;
  }
};

int pow2ceil(int v) { int r = 1; while (r < v) r *= 2; return r; }

// Tile of a resident-chain stage: S candidates per workgroup, ALL output columns in the workgroup.
// Preferred: the direct form (no split-K; a wave owns 16 rows x 16*NI columns and finishes them in
// registers), possible when a wave's 16 rows lie in one candidate and the whole tile fits 8 waves.
bool chain_tile(const Op& op, int S, int max_n, Tile* out, bool* direct) {
  if (op.kind != 0 || op.Npad > max_n || op.Npad % 16) return false;
  const int units = op.Npad / 16;
  const int NI = units > 8 ? 2 : 1;
  if (units % NI || units / NI > 8) return false;
  const int NG = units / NI;
  const int KK = (op.ntaps * op.Cin_p + op.Cin2_p) / 16;
  if (op.osplit && op.osplit % 16) return false;
  static const bool allow_direct = !(getenv("TREBI_CHAIN_DIRECT") && atoi(getenv("TREBI_CHAIN_DIRECT")) == 0);
  auto pro_ok = [&](int nteams16) {
    if (!op.pro) return true;
    if (op.stride != 1 || op.Lin != op.Lout || op.pro_gn.gch % 4 || op.Cin % 16) return false;
    const int ts = (S * GN_GROUPS * 4 <= nteams16) ? 64 : 16;
    return op.Lin * (op.pro_gn.gch / 4) <= 2 * ts && S * GN_GROUPS <= 16 * nteams16 / ts;
  };
  const int L = op.Lout;
  const bool lpow2 = L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || L == 32;
  if (allow_direct && lpow2 && NI == 1) {
    const int MS = (S * L + 15) / 16;
    bool ok = MS * NG <= 8 && (L >= 16 || S * L <= 16 || (S * L) % 16 == 0);
    if (op.gn) {
      const int gch = op.gnw.gch;
      const bool two = L == 32, twoc = gch == 32;
      ok = ok && op.Npad == op.N && op.N % GN_GROUPS == 0 && (gch == 4 || gch == 8 || gch == 16 || gch == 32) &&
           !(twoc && (two || L < 4 || NG % 2));
    }
    if (op.pro) {
      const int ts = (S * GN_GROUPS * 4 <= 4 * NG * MS) ? 64 : 16;
      ok = ok && op.stride == 1 && op.Lin == op.Lout && op.pro_gn.gch % 4 == 0 && op.Cin % 16 == 0 &&
           op.Lin * (op.pro_gn.gch / 4) <= 2 * ts;
    }
    if (ok) {
      *out = Tile{MS, NI, NG, MS, S};                 // MI = KS = row tiles: TM = 16*MS, 64*NG*MS threads
      *direct = true;
      return true;
    }
  }
  *direct = false;
  const int MI = pow2ceil((S * op.Lout + 15) / 16);
  if (!chain_variant_ok(MI, NI)) return false;
  const int KS = std::max(1, std::min(8 / NG, std::max(1, KK / 4)));
  const int nteams16 = 4 * NG * KS;
  if (op.gn) {
    if (op.Npad != op.N || op.gnw.gch % 4 || op.N % GN_GROUPS) return false;
    const int ts = (S * GN_GROUPS * 4 <= nteams16) ? 64 : 16;       // all groups sit in the one column tile
    if (op.Lout * (op.gnw.gch / 4) > 2 * ts) return false;          // a team keeps its group in registers
  }
  if (!pro_ok(nteams16)) return false;
  *out = Tile{MI, NI, NG, KS, S};
  return true;
}

// Try to fold ops [i, ...) into one resident-chain launch; returns the number of ops folded (0: none).
size_t plan_chain(const std::vector<const Op*>& o0, const std::vector<const Op*>& o1, size_t i, const Resolver& r0,
                  const Resolver* r1, int B, int S, int max_n, bool want_grad, std::vector<RStage>& tab, ParTab& par,
                  Plan& plan) {
  const int nz = r1 ? 2 : 1;
  std::vector<Tile> tiles;
  std::vector<char> directs;
  size_t j = i;
  static const size_t max_len = getenv("TREBI_CHAIN_MAXLEN") ? (size_t)atoi(getenv("TREBI_CHAIN_MAXLEN")) : 1000;
  // The value head as a stage of the chain it follows (chain.hip head_stage): one launch and one boundary fewer per value
  // pass (112 launches per planning call at the headline), and the tail of the forward runs into the head of the backward.
  // Time-neutral at 64 candidates (five interleaved pairs of processes: median 18.75 ms with, 18.74 ms without; the
  // process-to-process spread is +-0.3 ms): the kernel instantiation that carries the head's code runs its ordinary stages a
  // little slower, which is what the saved boundaries return.  TREBI_CHAIN_HEAD=0: the head as a kernel of its own.
  static const bool head_in_chain = !(getenv("TREBI_CHAIN_HEAD") && atoi(getenv("TREBI_CHAIN_HEAD")) == 0);
  while (j < o0.size() && j - i < max_len) {
    Tile tl;
    bool direct = false;
    if (o0[j]->kind == 1) {          // the value head: a stage of a direct chain (never its first: the first stage brings the table in)
      if (!head_in_chain || directs.empty() || !directs[0] || S > 8 || o0[j]->Cin % 4 || r0.net->hd > 64) break;
      tiles.push_back(Tile{1, 1, 0, 0, S});
      directs.push_back(1);
      ++j;
      continue;
    }
    if (!chain_tile(*o0[j], S, max_n, &tl, &direct)) break;
    if (!directs.empty() && (bool)directs[0] != direct) break;      // one stage kind per launch
    tiles.push_back(tl);
    directs.push_back(direct);
    ++j;
  }
  struct Key {
    int id, coloff, C;
    bool operator==(const Key& k) const { return id == k.id && coloff == k.coloff && C == k.C; }
  };
  auto key_of = [](const BufRef& r, int C) { return Key{r.id, r.coloff, C}; };
  auto reads = [&](const Op& op, const Key& k) {      // op reads exactly this logical tensor
    if (op.kind == 1) return key_of(op.src, op.Cin) == k;
    if (op.kind != 0) return false;
    return key_of(op.src, op.Cin) == k || (op.src2.id != NONE && key_of(op.src2, op.Cin2) == k) ||
           (op.res.id != NONE && key_of(op.res, op.N) == k);
  };
  auto touches = [&](const Op& op, int id) {          // op reads workspace buffer `id` through any view
    if (op.kind == 1) return op.v_buf == id;
    return op.src.id == id || op.src2.id == id || op.res.id == id || op.pro_y == id || op.pro_g == id ||
           op.pro_st == id;
  };
  for (; j >= i + 2; --j) {                            // shrink until the chain fits in LDS
    const size_t n = j - i;
    std::vector<RStage> run(n);
    struct Live { Key k; int region; size_t last; int L; };
    std::vector<Live> live;
    constexpr int MAXREG = 4;
    size_t regsz = 0, slabsz = 0, slab2sz = 0, restsz = 0, redsz = 0, red2sz = 0, colsz = 0, tbsz = 0, headsz = 0;
    int nreg = 0;
    std::vector<int> written;                          // workspace ids stored to global by earlier stages
    double fl = 0;
    for (size_t k = 0; k < n; ++k) {
      const Op& op = *o0[i + k];
      RStage& st = run[k];
      const Tile& tl = tiles[k];
      st.MI = tl.MI; st.NI = tl.NI;
      st.direct = directs[k];
      if (op.kind == 1) {              // the value head (RStage::head, chain.hip head_stage)
        auto head_p = [&](const Resolver& r, const Op& o) {
          const HeadP h = r.head(o, want_grad);
          ConvP p{};
          p.x = h.v; p.ldx = h.fc; p.Cin = p.Cin_p = h.fc; p.Lin = p.Lout = 1; p.stride = 1; p.ntaps = 1;
          p.w = h.w1; p.tbias = h.tb; p.gn = 1; p.gch = 1; p.gamma = h.w2; p.N = p.Npad = h.hd;
          p.bias2 = r.net->head_b2_dev;                  // b2 replicated over hd columns: the table's last row
          p.out = h.y; p.save_y = h.dv; p.T = h.T; p.t_uniform = -1; p.B = B;
          return p;
        };
        st.p[0] = head_p(r0, op);
        if (r1) st.p[1] = head_p(*r1, *o1[i + k]);
        st.head = 1;
        st.vec = 1;
        st.g = ConvGeo{};
        st.g.S = S; st.g.ldA = op.Cin + 8;
        headsz = std::max(headsz, (size_t)r0.net->hd * (op.Cin + 4) + (size_t)S * r0.net->hd + (size_t)S * op.Cin);
      } else {
      st.p[0] = r0.conv(op);
      if (r1) st.p[1] = r1->conv(*o1[i + k]);
      st.vec = conv_vec_ok(st.p[0]) && (!r1 || conv_vec_ok(st.p[1]));
      st.g = conv_make_geo(st.p[0], tl);
      }
      const ConvGeo& G = st.g;
      auto find_live = [&](const Key& kk) -> Live* {
        for (auto& l : live) if (l.k == kk) return &l;
        return nullptr;
      };
      // sources
      Live* ls = op.src.id >= 0 ? find_live(key_of(op.src, op.Cin)) : nullptr;
      st.proRes = ls != nullptr && op.pro;            // the prologue copies: the K loop reads the stage slab
      st.srcRes = ls != nullptr && !op.pro;
      st.Rb = st.srcRes ? op.Lin + 4 : G.R;
      st.a0 = st.srcRes ? 2 + op.tap_lo - op.pad : 0;
      st.pr0 = G.fh;
      Live* l2 = (op.src2.id >= 0) ? find_live(key_of(op.src2, op.Cin2)) : nullptr;
      st.x2Res = l2 != nullptr;
      st.Rb2 = l2 ? op.Lout + 4 : op.Lout;
      st.h2 = l2 ? 2 : 0;
      Live* lr = (op.res.id >= 0) ? find_live(key_of(op.res, op.N)) : nullptr;
      st.resRes = lr != nullptr;
      st.RbR = lr ? op.Lout + 4 : op.Lout;
      st.hR = lr ? 2 : 0;
      st.ldR = lr ? op.N + 8 : G.TNp;
      const int reg_s = ls ? ls->region : -1, reg_2 = l2 ? l2->region : -1, reg_r = lr ? lr->region : -1;
      const bool have_s = ls != nullptr, have_2 = l2 != nullptr, have_r = lr != nullptr;
      if (st.a0 < 0 || (!st.vec && op.pro)) { run.clear(); break; }
      // global sources written earlier in this chain need the stores drained first
      auto was_written = [&](int id) { return id >= 0 && std::find(written.begin(), written.end(), id) != written.end(); };
      st.drain = (!have_s && was_written(op.src.id)) || (op.src2.id != NONE && !have_2 && was_written(op.src2.id)) ||
                 (op.res.id != NONE && !have_r && was_written(op.res.id)) ||
                 (op.pro && (was_written(op.pro_y) || was_written(op.pro_g) || was_written(op.pro_st)));
      // destination
      const int Cd = op.osplit ? op.osplit : op.N, Ld = op.osplit ? 2 * op.Lout : op.Lout;
      const Key kd = key_of(op.dst, Cd);
      size_t last = 0;
      for (size_t m = k + 1; m < n; ++m)
        if (reads(*o0[i + m], kd)) last = m;
      bool need_global = op.dst.id < 0;
      for (size_t m = 0; m < o0.size() && !need_global; ++m) {
        if (m >= i && m < j) {
          if (m > i + k && touches(*o0[m], op.dst.id) && !reads(*o0[m], kd)) need_global = true;
          continue;
        }
        if (m > i + k && touches(*o0[m], op.dst.id)) need_global = true;
      }
      st.dstRes = 0;
      static const bool no_resident = getenv("TREBI_CHAIN_RESIDENT") && atoi(getenv("TREBI_CHAIN_RESIDENT")) == 0;
      if (last > 0 && op.dst.id >= 0 && op.N == op.Npad && Cd % 16 == 0 && !no_resident) {
        bool busy[MAXREG] = {false, false, false, false};
        for (auto& l : live)
          if (l.last >= k) busy[l.region] = true;       // still read by this or a later stage
        int reg = -1;
        for (int q = 0; q < MAXREG; ++q)
          if (!busy[q]) { reg = q; break; }
        if (reg >= 0) {
          live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.region == reg; }),
                     live.end());
          live.push_back({kd, reg, last, Ld});
          st.dstRes = 1;
          st.oD = reg;                                   // region index for now, offset below
          st.RbD = Ld + 4; st.ldD = Cd + 8; st.Ld = Ld;   // row stride as conv_make_geo's ldA (bank rule there)
          st.mLd4 = st.ldD / 4 > 1 ? 0xFFFFFFFFu / (unsigned)(st.ldD / 4) + 1u : 0u;
          nreg = std::max(nreg, reg + 1);
          regsz = std::max(regsz, (size_t)S * (Ld + 4) * (Cd + 8));
        } else {
          need_global = true;
        }
      } else if (last > 0) {
        need_global = true;
      }
      if (op.kind == 1) {               // the head's `out` is y; its tensor destination (dL/dv) is save_y
        if (!need_global) { st.p[0].save_y = nullptr; st.p[1].save_y = nullptr; }
      } else if (!need_global) { st.p[0].out = nullptr; st.p[1].out = nullptr; }
      if (need_global && op.dst.id >= 0) written.push_back(op.dst.id);
      for (int id : {op.save_y, op.save_g, op.save_st}) if (id >= 0) written.push_back(id);
      // remember region indices of the sources (offsets are assigned once the carve is known)
      st.oSrc = st.srcRes ? reg_s : -1;
      st.oPro = st.proRes ? reg_s : -1;
      st.o2 = reg_2;
      st.oR = reg_r;
      if (!st.srcRes) slabsz = std::max(slabsz, (size_t)S * G.R * G.ldA);
      if (op.Cin2_p && !have_2) slab2sz = std::max(slab2sz, (size_t)S * op.Lout * G.ld2);
      if (op.res.id != NONE && !have_r) restsz = std::max(restsz, (size_t)G.TM * G.TNp);
      if (st.direct) {
        redsz = std::max(redsz, (size_t)8 * 64 * 2);           // (mean, M2) exchange between partner waves
      } else {
        redsz = std::max(redsz, (size_t)G.red1);
        if (op.Cin2_p) red2sz = std::max(red2sz, (size_t)G.red1);
        colsz = std::max(colsz, (size_t)4 * G.TN);
        if (op.tbias) tbsz = std::max(tbsz, (size_t)S * G.TN);
      }
      fl += op.flops * B * nz;
    }
    if (run.empty()) continue;
    auto r4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
    regsz = r4(regsz);
    const size_t oSlab = nreg * regsz, oSlab2 = oSlab + r4(slabsz), oRest = oSlab2 + r4(slab2sz),
                 oRed = oRest + r4(restsz), oRed2 = oRed + r4(redsz), oCol = oRed2 + r4(red2sz),
                 oTb = oCol + r4(colsz), oHead = oTb + r4(tbsz), total = oHead + r4(headsz);
    // direct chains: LDS copy of the per-column parameter table behind everything else
    int parN = 16;
    for (size_t k = 0; k < n; ++k) parN = std::max(parN, o0[i + k]->Npad);
    const size_t tabsz = directs[0] ? n * 5 * (size_t)parN : 0;
    if (tabsz > 8192) continue;                                    // what the kernel's registers carry (chain.hip TABV)
    if ((total + tabsz) * sizeof(float) > 160 * 1024) continue;
    for (auto& st : run) {
      st.oSrc = st.srcRes ? (int)(st.oSrc * regsz) : (int)oSlab;
      st.oPro = st.proRes ? (int)(st.oPro * regsz) : 0;
      st.o2 = st.x2Res ? (int)(st.o2 * regsz) : (int)oSlab2;
      st.oR = st.resRes ? (int)(st.oR * regsz) : (int)oRest;
      st.oD = st.dstRes ? (int)(st.oD * regsz) : 0;
      st.oRed = (int)oRed; st.oRed2 = (int)oRed2; st.oColp = (int)oCol; st.oTb = (int)oTb; st.oHead = (int)oHead;
    }
    PlanStep ps;
    ps.kind = 2; ps.nz = nz;
    ps.chain_off = (int)tab.size(); ps.chain_n = (int)n; ps.chain_direct = directs[0];
    ps.nblocks = (B + S - 1) / S;
    ps.chain_lds = (total + tabsz) * sizeof(float); ps.flops = fl;
    if (directs[0]) {
      ps.parN = parN; ps.parT = run[0].p[0].T; ps.oTab = (int)total;
      ps.par_off = par.add(run, nz, parN, ps.parT);
      for (auto& st : run) {
        ps.has_head |= st.head;
        if (!st.head) ps.feat |= (st.p[0].pro ? 1 : 0) | ((st.p[0].save_y || st.p[0].save_g || st.p[0].save_stats) ? 2 : 0);
      }
    }
    ps.name = std::string(directs[0] ? "direct " : "") + "chain[" + std::to_string(n) + "] S" + std::to_string(S) + " " + o0[i]->name + " .. " + o0[j - 1]->name;
    tab.insert(tab.end(), run.begin(), run.end());
    plan.push_back(ps);
    return n;
  }
  return 0;
}

// A single layer as a one-stage direct launch: workgroups own MS row tiles x NG column tiles of 16x16,
// one wave per tile over the whole reduction, grid = (candidate groups, column tiles, nets).
bool plan_direct_single(const Op& op, const Op* op1, const Resolver& r0, const Resolver* r1, int B,
                        std::vector<RStage>& tab, ParTab& par, Plan& plan) {
  static const bool allow = getenv("TREBI_DIRECT_SINGLE") && atoi(getenv("TREBI_DIRECT_SINGLE")) != 0;   // experimental: the split-K kernel is still faster for wide layers
  static const int max_b = getenv("TREBI_DIRECT_MAXB") ? atoi(getenv("TREBI_DIRECT_MAXB")) : 256;
  if (!allow || B > max_b || op.kind != 0 || op.Npad % 16) return false;
  const int nz = r1 ? 2 : 1;
  const int L = op.Lout, units = op.Npad / 16;
  if (!(L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || L == 32)) return false;
  if (op.osplit && op.osplit % 16) return false;
  const int gch = op.gn ? op.gnw.gch : 0;
  const bool two = L == 32, twoc = gch == 32;
  if (op.gn && (op.Npad != op.N || op.N % GN_GROUPS || !(gch == 4 || gch == 8 || gch == 16 || gch == 32))) return false;
  if (twoc && (two || L < 4 || units % 2)) return false;
  const int S = L >= 16 ? 1 : 16 / L;
  const int MS = (S * L + 15) / 16;
  int NG = 1;
  for (int c : {2, 4, 8})
    if (units % c == 0 && c * MS <= 8 && c <= 4) NG = c;
  if (twoc && NG % 2) return false;
  if (op.pro) {
    const int ts = (S * GN_GROUPS * 4 <= 4 * NG * MS) ? 64 : 16;
    if (op.stride != 1 || op.Lin != op.Lout || op.pro_gn.gch % 4 || op.Cin % 16 || op.Lin * (op.pro_gn.gch / 4) > 2 * ts)
      return false;
  }
  RStage st{};
  st.p[0] = r0.conv(op);
  if (r1) st.p[1] = r1->conv(*op1);
  if (!st.p[0].x || !st.p[0].out) return false;
  const Tile tl{MS, 1, NG, MS, S};
  st.MI = MS; st.NI = 1; st.direct = 1;
  st.vec = conv_vec_ok(st.p[0]) && (!r1 || conv_vec_ok(st.p[1]));
  if (op.pro && !st.vec) return false;
  st.g = conv_make_geo(st.p[0], tl);
  const ConvGeo& G = st.g;
  auto r4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
  const size_t slab = r4((size_t)S * G.R * G.ldA), slab2 = op.Cin2_p ? r4((size_t)S * op.Lout * G.ld2) : 0,
               rest = op.res.id != NONE ? r4((size_t)G.TM * G.TNp) : 0, xch = 8 * 64 * 2;
  st.srcRes = 0; st.oSrc = 0; st.Rb = G.R; st.a0 = 0; st.pr0 = G.fh;
  st.x2Res = 0; st.o2 = (int)slab; st.Rb2 = op.Lout; st.h2 = 0;
  st.resRes = 0; st.oR = (int)(slab + slab2); st.RbR = op.Lout; st.hR = 0; st.ldR = G.TNp;
  st.dstRes = 0; st.oD = 0; st.RbD = 1; st.ldD = 4; st.Ld = 0; st.mLd4 = 0;
  st.oRed = (int)(slab + slab2 + rest); st.oRed2 = st.oRed; st.oColp = st.oRed; st.oTb = st.oRed;
  const size_t total = slab + slab2 + rest + xch;
  const int parN = op.Npad;
  const size_t tabsz = 5 * (size_t)parN;
  if ((total + tabsz) * sizeof(float) > 160 * 1024 || tabsz > 8192) return false;
  PlanStep ps;
  ps.kind = 2; ps.nz = nz;
  ps.chain_off = (int)tab.size(); ps.chain_n = 1; ps.chain_direct = 1;
  ps.nblocks = (B + S - 1) / S;
  ps.ny = op.Npad / (16 * NG);
  ps.chain_lds = (total + tabsz) * sizeof(float);
  ps.parN = parN; ps.parT = st.p[0].T; ps.oTab = (int)total;
  ps.par_off = par.add(std::vector<RStage>{st}, nz, parN, ps.parT);
  ps.feat = (st.p[0].pro ? 1 : 0) | ((st.p[0].save_y || st.p[0].save_g || st.p[0].save_stats) ? 2 : 0);
  ps.flops = op.flops * B * nz;
  char tb[96];
  snprintf(tb, sizeof tb, " direct MS%d NG%d S%d grid %dx%d", MS, NG, S, ps.nblocks, ps.ny);
  ps.name = op.name + (op.pro ? " (bwd)" : "") + tb;
  tab.push_back(st);
  plan.push_back(ps);
  return true;
}

int build_plan(const std::vector<const std::vector<Op>*>& lists0, const std::vector<const std::vector<Op>*>& lists1,
               const Resolver& r0, const Resolver* r1, int B, bool want_grad, int chain_S, int chain_max_n,
               std::vector<RStage>& tab, ParTab& par, Plan& plan) {
  const int nz = r1 ? 2 : 1;
  std::vector<const Op*> o0, o1;
  for (auto* l : lists0) for (auto& op : *l) o0.push_back(&op);
  if (r1) for (auto* l : lists1) for (auto& op : *l) o1.push_back(&op);
  size_t i = 0;
  while (i < o0.size()) {
    const Op& op = *o0[i];
    if (op.kind == 1) {                // (a direct chain takes the head as a stage when it follows one of its stages: plan_chain)
      PlanStep st;
      st.kind = 1; st.nz = nz;
      st.H.p[0] = r0.head(op, want_grad);
      if (r1) st.H.p[1] = r1->head(*o1[i], want_grad);
      st.flops = op.flops * B * nz;
      plan.push_back(st);
      ++i;
      continue;
    }
    if (chain_S > 0) {
      size_t n = plan_chain(o0, o1, i, r0, r1, B, chain_S, chain_max_n, want_grad, tab, par, plan);
      if (n) { i += n; continue; }
    }
    if (plan_direct_single(op, r1 ? o1[i] : nullptr, r0, r1, B, tab, par, plan)) { ++i; continue; }
    PlanStep st;
    st.kind = 0; st.nz = nz;
    st.L.p[0] = r0.conv(op);
    if (r1) st.L.p[1] = r1->conv(*o1[i]);
    const bool bf = use_bf(r0.net, op, B) && conv_vec_ok(st.L.p[0]) && (!r1 || (use_bf(r1->net, *o1[i], B) && conv_vec_ok(st.L.p[1])));
    try {
      st.tl = choose_tile(op, B, nz, bf);
    } catch (const Unsupported& u) {
      return fail(TREBI_UNSUPPORTED_CONFIG, "%s", u.msg.c_str());
    }
    if (st.tl.S * op.Lout > 16 * st.tl.MI || op.Npad % (16 * st.tl.NI * st.tl.NG) || 64 * st.tl.NG * st.tl.KS > 512 ||
        (op.pro && (op.stride != 1 || op.Lin != op.Lout)) || !st.L.p[0].x || !st.L.p[0].out)
      return fail(TREBI_UNSUPPORTED_CONFIG, "internal tiling error at layer %s", op.name.c_str());
    st.flops = op.flops * B * nz;
    st.tl.bf = 0;
    if (bf) {
      st.tl.bf = 1;
      st.L.p[0].w = op.w16;
      if (r1) st.L.p[1].w = o1[i]->w16;
    }
    {
      char tb[96];
      snprintf(tb, sizeof tb, " MI%d NI%d NG%d KS%d S%d%s%s%s grid %dx%d", st.tl.MI, st.tl.NI, st.tl.NG, st.tl.KS, st.tl.S, st.tl.pm ? " pm" : "", st.tl.resg ? " resg" : "", st.tl.bf ? " bf16x3" : "",
               (B + st.tl.S - 1) / st.tl.S, op.Npad / (16 * st.tl.NI * st.tl.NG));
      st.name = op.name + (op.pro ? " (bwd)" : "") + tb;
      if (!st.tl.bf && narrow_ok(st.L, nz))        // launch_conv routes it to narrow.hip: the tile above is not what runs
        st.name = op.name + (op.pro ? " (bwd)" : "") + " narrow N" + std::to_string(op.Npad) + " L" + std::to_string(op.Lout) + " persistent";
    }
    plan.push_back(st);
    ++i;
  }
  return 0;
}

int exec_plan(const Plan& plan, const RStage* dev_tab, const float* dev_par, int t_uniform, const long long* tptr,
              hipStream_t st, GemmProfile* prof) {
  for (const PlanStep& ps : plan) {
    if (ps.kind == 1) {
      HeadLaunch H = ps.H;
      for (int z = 0; z < ps.nz; ++z) H.p[z].t = tptr;
      HIPCHK(launch_head(H, ps.nz, st));
      continue;
    }
    const bool timed = prof && prof->enabled;
    if (timed) {
      if (prof->used == prof->ev.size()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        prof->ev.push_back({a, b});
      }
      HIPCHK(hipEventRecord(prof->ev[prof->used].first, st));
    }
    if (ps.kind == 2) {
      HIPCHK(launch_chain(dev_tab + ps.chain_off, ps.chain_n, ps.chain_direct, ps.nz, ps.nblocks, ps.ny, ps.chain_lds, t_uniform, tptr,
                          dev_par + ps.par_off, ps.parN, ps.parT, ps.oTab, ps.has_head, ps.feat, st));
    } else {
      ConvLaunch L = ps.L;
      for (int z = 0; z < ps.nz; ++z) { L.p[z].t = tptr; L.p[z].t_uniform = t_uniform; }
      HIPCHK(launch_conv(L, ps.nz, ps.tl, st));
    }
    if (timed) {
      HIPCHK(hipEventRecord(prof->ev[prof->used].second, st));
      prof->flops.push_back(ps.flops);
      prof->names.push_back(ps.name);
      prof->used++;
    }
  }
  return 0;
}

bool same_arch(const trebi_net* a, const trebi_net* b) {
  const trebi_net_desc &x = a->desc, &y = b->desc;
  if (x.kind != y.kind || x.horizon != y.horizon || x.transition_dim != y.transition_dim || x.dim != y.dim ||
      x.n_mults != y.n_mults || a->T != b->T)
    return false;
  for (int i = 0; i < x.n_mults; ++i)
    if (x.dim_mults[i] != y.dim_mults[i]) return false;
  return a->fwd.size() == b->fwd.size() && a->bwd.size() == b->bwd.size();
}

}  // namespace

// ------------------------------------------------------------------------------------
// extern "C": basics
// ------------------------------------------------------------------------------------
extern "C" const char* trebi_last_error(void) { return g_err.c_str(); }
extern "C" int trebi_version(void) { return 1; }

extern "C" int trebi_ctx_create(int device, trebi_ctx** out) {
  if (!out) return fail(TREBI_BAD_ARG, "out is null");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(TREBI_BAD_ARG, "device %d out of range (%d visible)", device, n);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(TREBI_UNSUPPORTED_CONFIG, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                prop.gcnArchName);
  HIPCHK(conv_init_device());
  HIPCHK(chain_init_device());
  if (device < MAX_DEVICES) {                      // once per device, that device current; contexts may be created concurrently
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
      void* p = nullptr;
      HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)));
      g_scratch_ptr[device] = p;
    }
  }
  *out = new trebi_ctx{device};
  return 0;
}
extern "C" void trebi_ctx_destroy(trebi_ctx* ctx) { delete ctx; }

extern "C" int trebi_net_param_count(const trebi_net_desc* desc) {
  if (int e = check_desc(desc)) return e;
  return (int)param_specs(*desc).size();
}
extern "C" int trebi_net_param_info(const trebi_net_desc* desc, int i, char* name_out, size_t name_cap,
                                    int64_t* numel_out) {
  if (int e = check_desc(desc)) return e;
  auto s = param_specs(*desc);
  if (i < 0 || i >= (int)s.size()) return fail(TREBI_BAD_ARG, "parameter index %d out of range", i);
  if (name_out && name_cap) {
    strncpy(name_out, s[i].name.c_str(), name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  if (numel_out) *numel_out = s[i].numel;
  return 0;
}

static int net_create_impl(trebi_ctx* ctx, const trebi_net_desc* desc, const char* const* names,
                           const float* const* tensors, const int64_t* numels, int n_tensors, bool train,
                           trebi_net** out) {
  if (!ctx || !out || !names || !tensors || !numels) return fail(TREBI_BAD_ARG, "null argument");
  if (int e = check_desc(desc)) return e;
  HIPCHK(hipSetDevice(ctx->device));
  auto specs = param_specs(*desc);
  std::unique_ptr<trebi_net> net(new trebi_net);
  net->ctx = ctx;
  net->desc = *desc;
  net->T = desc->max_timesteps;
  net->train = train;
  Builder B;
  B.net = net.get();
  B.dim = desc->dim;
  B.T = net->T;
  B.train = train;
  for (int i = 0; i < n_tensors; ++i) {
    if (!names[i] || !tensors[i]) return fail(TREBI_BAD_ARG, "tensor table entry %d is null", i);
    B.sd[names[i]] = {tensors[i], numels[i]};
  }
  for (auto& s : specs) {
    auto it = B.sd.find(s.name);
    if (it == B.sd.end()) return fail(TREBI_BAD_SHAPE, "state_dict is missing '%s'", s.name.c_str());
    if (it->second.second != s.numel)
      return fail(TREBI_BAD_SHAPE, "'%s' has %lld elements, the architecture needs %lld", s.name.c_str(),
                  (long long)it->second.second, (long long)s.numel);
  }
  if ((size_t)n_tensors != specs.size())
    return fail(TREBI_BAD_SHAPE, "state_dict has %d tensors, the architecture has %zu", n_tensors, specs.size());
  try {
    int dim = desc->dim;
    const float* w1 = B.upload(B.host("time_mlp.1.weight", (int64_t)4 * dim * dim), (size_t)4 * dim * dim);
    const float* b1 = B.upload(B.host("time_mlp.1.bias", 4 * dim), 4 * dim);
    const float* w2 = B.upload(B.host("time_mlp.3.weight", (int64_t)4 * dim * dim), (size_t)4 * dim * dim);
    const float* b2 = B.upload(B.host("time_mlp.3.bias", dim), dim);
    float* temb = B.dalloc((size_t)net->T * dim);
    if (train) {                     // the backward of time_mlp needs the embedding and the pre-activation too
      net->tt_e = B.dalloc((size_t)net->T * dim);
      net->tt_a1 = B.dalloc((size_t)net->T * 4 * dim);
      net->tt_temb = temb;
      HIPCHK(launch_time_embed_train(w1, b1, w2, b2, dim, net->T, net->tt_e, net->tt_a1, temb, 0));
      net->time_jobs.push_back({0, w1, b1, w2, b2, 0, 0, 0, 0, temb});
    } else {
      HIPCHK(launch_time_embed(w1, b1, w2, b2, dim, net->T, temb, 0));
    }
    B.temb = temb;
    ArchPlan a = make_arch(*desc);
    if (desc->kind == 0) build_unet(B, a); else build_value(B, a);
    // every layer must tile for at least a batch of one
    for (auto* ops : {&net->fwd, &net->bwd})
      for (auto& op : *ops)
        if (op.kind == 0) (void)choose_tile(op, 1);
  } catch (const Unsupported& u) {
    return fail(TREBI_UNSUPPORTED_CONFIG, "%s", u.msg.c_str());
  } catch (const BadShape& b) {
    return fail(TREBI_BAD_SHAPE, "%s", b.msg.c_str());
  }
  for (auto& op : net->fwd) net->flops_fwd += op.flops;
  for (auto& op : net->bwd) net->flops_bwd += op.flops;
  HIPCHK(hipDeviceSynchronize());
  *out = net.release();
  return 0;
}
extern "C" int trebi_net_create(trebi_ctx* ctx, const trebi_net_desc* desc, const char* const* names,
                                const float* const* tensors, const int64_t* numels, int n_tensors,
                                trebi_net** out) {
  return net_create_impl(ctx, desc, names, tensors, numels, n_tensors, false, out);
}
extern "C" void trebi_net_destroy(trebi_net* net) { delete net; }

// (Re)build the bf16x3 copies of every packed weight block from the fp32 ones (after a re-pack while training).
static int net_refresh_bf16(trebi_net* net, hipStream_t st) {
  for (auto& j : net->bf_jobs) HIPCHK(launch_pack_bf16x3(j.w32, j.w16, j.steps16, j.Npad, st));
  return 0;
}

extern "C" int trebi_net_set_precision(trebi_net* net, int mode, int min_batch) {
  if (!net) return fail(TREBI_BAD_ARG, "null net");
  if (mode != 0 && mode != 1) return fail(TREBI_BAD_ARG, "precision mode must be 0 (f32) or 1 (bf16x3)");
  HIPCHK(hipSetDevice(net->ctx->device));
  if (mode == 1 && !net->bf_arena) {
    std::vector<Op*> ops;
    for (auto& op : net->fwd) ops.push_back(&op);
    for (auto& op : net->bwd) ops.push_back(&op);
    for (auto& t : net->tops)
      if (t.kind == 0) ops.push_back(&t.conv);
    size_t total = 0;
    std::vector<size_t> offs;
    for (Op* op : ops) {
      const bool ok = op->kind == 0 && !op->pro && op->w && (op->Cin_p % 32) == 0 && (op->Cin2_p % 32) == 0;
      offs.push_back(ok ? total : (size_t)-1);
      if (ok) total += ((size_t)((op->ntaps * op->Cin_p + op->Cin2_p) / 16 + 8) * 16 * op->Npad * sizeof(float) + 255) / 256 * 256;
    }
    HIPCHK(hipMalloc(&net->bf_arena, std::max<size_t>(total, 256)));
    for (size_t i = 0; i < ops.size(); ++i) {
      if (offs[i] == (size_t)-1) continue;
      Op* op = ops[i];
      float* w16 = (float*)((char*)net->bf_arena + offs[i]);
      op->w16 = w16;
      net->bf_jobs.push_back({op->w, w16, (op->ntaps * op->Cin_p + op->Cin2_p) / 16 + 8, op->Npad});
    }
    if (int e = net_refresh_bf16(net, 0)) return e;
    HIPCHK(hipDeviceSynchronize());
  }
  net->prec = mode;
  if (min_batch > 0) net->bf_min_b = min_batch;
  return 0;
}

extern "C" size_t trebi_unet_workspace_bytes(const trebi_net* net, int B) {
  return net && B > 0 ? net->ws_floats(B) * sizeof(float) : 0;
}
extern "C" size_t trebi_value_workspace_bytes(const trebi_net* net, int B) {
  return net && B > 0 ? net->ws_floats(B) * sizeof(float) : 0;
}

// Where the output of a named ResidualTemporalBlock sits in the workspace after trebi_unet_forward (parity taps).
extern "C" int trebi_unet_tap(const trebi_net* net, const char* block, int B, size_t* offset_floats, int* L, int* C,
                              int* ld) {
  if (!net || !block || !offset_floats || !L || !C || !ld || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  if (net->desc.kind != 0) return fail(TREBI_BAD_ARG, "not a TemporalUnet");
  const UnetRec& rec = net->rec;
  std::vector<const RBInfo*> all{&rec.mid1, &rec.mid2};
  for (auto* v : {&rec.down0, &rec.down1, &rec.up0, &rec.up1})
    for (auto& r : *v) all.push_back(&r);
  for (const RBInfo* r : all)
    if (r->prefix == block) {
      if (r->dst.id < 0) return fail(TREBI_BAD_ARG, "block '%s' writes to a caller tensor, not the workspace", block);
      *offset_floats = net->offsets(B)[r->dst.id] + r->dst.coloff;
      *L = r->L; *C = r->cout; *ld = r->dst.ld;
      return 0;
    }
  return fail(TREBI_BAD_ARG, "no ResidualTemporalBlock named '%s'", block);
}

// Per-layer parity taps: what every launch of a pass reads and writes, so that a test can hold ONE layer to the oracle's
// Conv1dBlock / Downsample1d / Upsample1d / input-gradient unit on the very tensors the library fed it.
extern "C" int trebi_net_layer_count(const trebi_net* net, int pass) {
  if (!net || pass < 0 || pass > 1) return 0;
  return (int)(pass ? net->bwd.size() : net->fwd.size());
}

extern "C" int trebi_net_layer_info(const trebi_net* net, int pass, int index, int B, trebi_layer_info* out) {
  if (!net || !out || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  if (pass < 0 || pass > 1) return fail(TREBI_BAD_ARG, "pass must be 0 (forward) or 1 (backward-to-input)");
  const std::vector<Op>& ops = pass ? net->bwd : net->fwd;
  if (index < 0 || index >= (int)ops.size()) return fail(TREBI_BAD_ARG, "layer index %d outside 0..%zu", index, ops.size());
  const Op& op = ops[index];
  const std::vector<size_t> off = net->offsets(B);
  memset(out, 0, sizeof *out);
  snprintf(out->name, sizeof out->name, "%s", op.name.c_str());
  out->kind = op.kind; out->Lin = op.Lin; out->Lout = op.Lout; out->Cin = op.Cin; out->Cin2 = op.Cin2; out->N = op.N;
  out->stride = op.stride; out->pad = op.pad; out->phase_split = op.osplit;
  out->group_norm = op.gn ? 1 : 0; out->time_bias = op.tbias ? 1 : 0; out->prologue = op.pro ? 1 : 0;
  auto ref = [&](const BufRef& r, int L, int C, trebi_tensor_ref* o) {
    o->L = L; o->C = C;
    if (r.id == NONE) return;
    if (r.id >= 0) { o->where = 1; o->offset_floats = off[r.id] + r.coloff; o->ld = r.ld; }
    else o->where = r.id == EXT_X ? 2 : r.id == EXT_OUT ? 3 : 4;
  };
  auto whole = [&](int id, int L, int C, trebi_tensor_ref* o) { ref(id == NONE ? BufRef{} : BufRef{id, C, 0}, L, C, o); };
  ref(op.src, op.Lin, op.Cin, &out->src);
  ref(op.src2, op.Lout, op.Cin2, &out->src2);
  ref(op.res, op.Lout, op.N, &out->res);
  if (op.osplit) ref(op.dst, 2 * op.Lout, op.osplit, &out->dst);
  else ref(op.dst, op.Lout, op.N, &out->dst);
  whole(op.save_y, op.Lout, op.N, &out->saved_xhat);
  whole(op.save_g, op.Lout, op.N, &out->saved_gain);
  whole(op.save_st, GN_GROUPS, 2, &out->saved_stats);
  whole(op.pro_y, op.Lin, op.Cin, &out->pro_xhat);
  whole(op.pro_g, op.Lin, op.Cin, &out->pro_gain);
  whole(op.pro_st, GN_GROUPS, 2, &out->pro_stats);
  return 0;
}

extern "C" int trebi_unet_forward(trebi_net* net, const float* x, int ldx, const int64_t* t, float* out, int ldo,
                                  int B, void* ws, size_t ws_bytes, void* stream) {
  if (!net || !x || !t || !out || !ws) return fail(TREBI_BAD_ARG, "null argument");
  if (net->desc.kind != 0) return fail(TREBI_BAD_ARG, "not a TemporalUnet");
  if (B < 1 || ldx < net->desc.transition_dim || ldo < net->desc.transition_dim)
    return fail(TREBI_BAD_SHAPE, "B=%d ldx=%d ldo=%d do not fit transition_dim=%d", B, ldx, ldo,
                net->desc.transition_dim);
  if (ws_bytes < trebi_unet_workspace_bytes(net, B))
    return fail(TREBI_WORKSPACE_TOO_SMALL, "workspace %zu < %zu bytes", ws_bytes, trebi_unet_workspace_bytes(net, B));
  Binding b;
  b.x = x; b.ldx = ldx; b.out = out; b.ldo = ldo; b.t = (const long long*)t; b.ws = (float*)ws;
  return run_ops(net->fwd, net, b, nullptr, nullptr, nullptr, B, false, (hipStream_t)stream, nullptr);
}

// The no-allocation variant: same kernels, the workspace is this GPU's static scratch array.  Nothing in the
// workspace has to start at zero (every buffer is fully written before it is read; trebi_unet_forward runs on
// uninitialised caller memory too), so the previous call's contents need no clearing.
extern "C" int trebi_unet_forward_static(trebi_net* net, const float* x, int ldx, const int64_t* t, float* out,
                                         int ldo, int B, void* stream) {
  if (!net || !x || !t || !out) return fail(TREBI_BAD_ARG, "null argument");
  if (net->desc.kind != 0) return fail(TREBI_BAD_ARG, "not a TemporalUnet");
  if (B < 1 || ldx < net->desc.transition_dim || ldo < net->desc.transition_dim)
    return fail(TREBI_BAD_SHAPE, "B=%d ldx=%d ldo=%d do not fit transition_dim=%d", B, ldx, ldo,
                net->desc.transition_dim);
  const int dev = net->ctx->device;
  void* ws = dev < MAX_DEVICES ? g_scratch_ptr[dev] : nullptr;
  if (!ws) return fail(TREBI_HIP_ERROR, "no static scratch on device %d (context not created through trebi_ctx_create)", dev);
  const size_t need = trebi_unet_workspace_bytes(net, B);
  if (need > (size_t)TREBI_STATIC_SCRATCH_BYTES)
    return fail(TREBI_WORKSPACE_TOO_SMALL, "B=%d needs %zu bytes of scratch, the static array holds %u: use "
                "trebi_unet_forward with a caller workspace", B, need, (unsigned)TREBI_STATIC_SCRATCH_BYTES);
  Binding b;
  b.x = x; b.ldx = ldx; b.out = out; b.ldo = ldo; b.t = (const long long*)t; b.ws = (float*)ws;
  return run_ops(net->fwd, net, b, nullptr, nullptr, nullptr, B, false, (hipStream_t)stream, nullptr);
}

static int value_run(trebi_net* n0, const Binding& b0, trebi_net* n1, const Binding* b1, int B, bool grad,
                     hipStream_t st, GemmProfile* prof) {
  if (int e = run_ops(n0->fwd, n0, b0, n1, b1, n1 ? &n1->fwd : nullptr, B, grad, st, prof)) return e;
  if (grad)
    if (int e = run_ops(n0->bwd, n0, b0, n1, b1, n1 ? &n1->bwd : nullptr, B, true, st, prof)) return e;
  return 0;
}

extern "C" int trebi_value_gradients(trebi_net* net, const float* x, int ldx, const int64_t* t, float* y,
                                     float* grad, int ldg, int B, void* ws, size_t ws_bytes, void* stream) {
  if (!net || !x || !t || !y || !ws) return fail(TREBI_BAD_ARG, "null argument");
  if (net->desc.kind != 1) return fail(TREBI_BAD_ARG, "not a ValueFunction");
  int D = net->desc.transition_dim;
  if (B < 1 || ldx < D || (grad && ldg < D))
    return fail(TREBI_BAD_SHAPE, "B=%d ldx=%d ldg=%d do not fit transition_dim=%d", B, ldx, ldg, D);
  if (ws_bytes < trebi_value_workspace_bytes(net, B))
    return fail(TREBI_WORKSPACE_TOO_SMALL, "workspace %zu < %zu bytes", ws_bytes,
                trebi_value_workspace_bytes(net, B));
  hipStream_t st = (hipStream_t)stream;
  if (grad && ldg > D) HIPCHK(hipMemsetAsync(grad, 0, (size_t)B * net->desc.horizon * ldg * sizeof(float), st));
  Binding b;
  b.x = x; b.ldx = ldx; b.grad = grad; b.ldg = ldg; b.t = (const long long*)t; b.ws = (float*)ws; b.y = y;
  return value_run(net, b, nullptr, nullptr, B, grad != nullptr, st, nullptr);
}

// ------------------------------------------------------------------------------------
// elementwise steps + selection
// ------------------------------------------------------------------------------------
static int make_step(const trebi_step_desc* d, const float* const* cond, StepP* s) {
  if (!d) return fail(TREBI_BAD_ARG, "null step descriptor");
  if (d->n_cond < 0 || d->n_cond > 4) return fail(TREBI_UNSUPPORTED_CONFIG, "at most 4 conditioned timesteps");
  if (d->transition_dim < d->action_dim + d->observation_dim || d->horizon < 1 || d->n_timesteps < 1)
    return fail(TREBI_BAD_SHAPE, "inconsistent step descriptor");
  s->H = d->horizon; s->D = d->transition_dim; s->act = d->action_dim; s->obs = d->observation_dim;
  s->T = d->n_timesteps; s->scale = d->scale; s->t_stopgrad = d->t_stopgrad;
  s->scale_by_std = d->scale_grad_by_std; s->state_mask = d->state_grad_mask; s->cgw = d->cost_grad_weight;
  s->guide_type = d->guide_type; s->n_cond = d->n_cond;
  for (int k = 0; k < d->n_cond; ++k) {
    if (!cond || !cond[k]) return fail(TREBI_BAD_ARG, "cond[%d] is null", k);
    if (d->cond_index[k] < 0 || d->cond_index[k] >= d->horizon)
      return fail(TREBI_BAD_SHAPE, "conditioned index %d outside the horizon", d->cond_index[k]);
    s->cond_index[k] = d->cond_index[k];
    s->cond[k] = cond[k];
  }
  return 0;
}

extern "C" int trebi_guide_update(trebi_ctx* ctx, const trebi_step_desc* d, float* x, const float* r_grad,
                                  const float* cost_grad, const float* cost_y, const float* thr, const int64_t* t,
                                  const float* plv, const float* const* cond, int B, void* stream) {
  if (!ctx || !x || !r_grad || !t || !plv || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  if (cost_grad && (!cost_y || !thr)) return fail(TREBI_BAD_ARG, "cost_grad needs cost_y and cost_threshold");
  if (d && d->guide_type != 0 && d->guide_type != 1) return fail(TREBI_BAD_ARG, "guide_type must be 0 or 1");
  StepP s{};
  if (int e = make_step(d, cond, &s)) return e;
  HIPCHK(launch_guide_update(s, x, r_grad, cost_grad, cost_y, thr, (const long long*)t, plv, B, (hipStream_t)stream));
  return 0;
}

extern "C" int trebi_ddpm_update(trebi_ctx* ctx, const trebi_step_desc* d, float* x, const float* x0,
                                 const float* noise, const int64_t* t, const float* plv, const float* c1,
                                 const float* c2, const float* const* cond, int B, void* stream) {
  if (!ctx || !x || !x0 || !noise || !t || !plv || !c1 || !c2 || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  StepP s{};
  if (int e = make_step(d, cond, &s)) return e;
  HIPCHK(launch_ddpm_update(s, x, x0, noise, (const long long*)t, plv, c1, c2, B, nullptr, (hipStream_t)stream));
  return 0;
}

extern "C" int trebi_feasibility_rule(trebi_ctx* ctx, const float* r_y, const float* cost_y, const float* thr,
                                      int n, float* out, void* stream) {
  if (!ctx || !r_y || !cost_y || !thr || !out || n < 1) return fail(TREBI_BAD_ARG, "null argument");
  HIPCHK(launch_feasibility(r_y, cost_y, thr, n, out, (hipStream_t)stream));
  return 0;
}
extern "C" int trebi_argsort_desc(trebi_ctx* ctx, const float* values, int n, int64_t* order, void* stream) {
  if (!ctx || !values || !order || n < 1) return fail(TREBI_BAD_ARG, "null argument");
  HIPCHK(launch_argsort_desc(values, n, (long long*)order, (hipStream_t)stream));
  return 0;
}
extern "C" int trebi_group_argmax(trebi_ctx* ctx, const float* values, int n_groups, int per, int64_t* best,
                                  void* stream) {
  if (!ctx || !values || !best || n_groups < 1 || per < 1) return fail(TREBI_BAD_ARG, "null argument");
  HIPCHK(launch_group_argmax(values, n_groups, per, (long long*)best, (hipStream_t)stream));
  return 0;
}

// ------------------------------------------------------------------------------------
// GuidedPolicy glue on the device (policies.py:25-45, 70-82)
// ------------------------------------------------------------------------------------
extern "C" int trebi_policy_prepare(trebi_ctx* ctx, const float* obs, int n_obs, int obs_dim, const float* mean,
                                    const float* stdv, int B, float* cond_out, const float* thr, int n_thr, float* thr_out,
                                    void* stream) {
  if (!ctx || !obs || !cond_out || n_obs < 1 || obs_dim < 1 || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  if ((mean == nullptr) != (stdv == nullptr)) return fail(TREBI_BAD_ARG, "mean and std go together");
  if (B % n_obs) return fail(TREBI_BAD_SHAPE, "batch %d is not a multiple of the %d observations", B, n_obs);
  if (thr_out && (!thr || n_thr < 1 || B % n_thr)) return fail(TREBI_BAD_SHAPE, "threshold list does not tile the batch");
  HIPCHK(launch_policy_prepare(obs, n_obs, obs_dim, mean, stdv, B, cond_out, thr, n_thr, thr_out, (hipStream_t)stream));
  return 0;
}
extern "C" int trebi_policy_first_action(trebi_ctx* ctx, const float* x, int B, int horizon, int transition_dim,
                                         int action_dim, const int64_t* index, int n, const float* mean, const float* stdv,
                                         float* action_out, void* stream) {
  if (!ctx || !x || !index || !action_out || n < 1 || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  if ((mean == nullptr) != (stdv == nullptr)) return fail(TREBI_BAD_ARG, "mean and std go together");
  if (action_dim < 1 || action_dim > transition_dim || horizon < 1) return fail(TREBI_BAD_SHAPE, "inconsistent dimensions");
  HIPCHK(launch_policy_action(x, horizon, transition_dim, action_dim, (const long long*)index, n, B, mean, stdv, action_out,
                              (hipStream_t)stream));
  return 0;
}

// ------------------------------------------------------------------------------------
// the captured sampling loop
// ------------------------------------------------------------------------------------
struct trebi_sampler {
  trebi_ctx* ctx = nullptr;
  trebi_net *unet = nullptr, *reward = nullptr, *cost = nullptr;
  trebi_sampler_desc d{};
  int B = 0;
  bool dual = false;
  void* arena = nullptr;            // one device allocation: tables + workspaces + state
  float *plv = nullptr, *c1 = nullptr, *c2 = nullptr;
  long long* tbuf = nullptr;        // [T, B] one timestep vector per step
  float *ws_u = nullptr, *ws_r = nullptr, *ws_c = nullptr;
  float *x0 = nullptr, *rgrad = nullptr, *cgrad = nullptr;
  // graph cache, keyed by the pointers baked into it
  hipGraphExec_t exec = nullptr;
  const void* key[12] = {};
  GemmProfile prof;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  bool have_loop_time = false;
  // launch plans (resolved once per set of caller pointers) and the device table of chained stages
  std::vector<Plan> vf_grad, vf_nograd;   // one entry (both nets in one launch / a single net) or two
  Plan unet_plan;
  RStage* dev_tab = nullptr;
  size_t dev_tab_cap = 0;
  float* dev_par = nullptr;         // per-column parameter tables of the direct chains (ParTab)
  size_t dev_par_cap = 0;
  const void* plan_key[3] = {};
  bool plans_valid = false;
  bool bound = false;               // trebi_sampler_bind succeeded for the pointers in `key`
  ~trebi_sampler() {
    if (dev_tab) (void)hipFree(dev_tab);
    if (dev_par) (void)hipFree(dev_par);
    if (exec) (void)hipGraphExecDestroy(exec);
    if (arena) (void)hipFree(arena);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_b) (void)hipEventDestroy(ev_b);
  }
};

extern "C" int trebi_sampler_create(trebi_ctx* ctx, trebi_net* unet, trebi_net* reward, trebi_net* cost,
                                    const trebi_sampler_desc* desc, const float* plv, const float* c1,
                                    const float* c2, int B, trebi_sampler** out) {
  if (!ctx || !unet || !desc || !plv || !c1 || !c2 || !out || B < 1) return fail(TREBI_BAD_ARG, "null argument");
  const trebi_step_desc& sd = desc->step;
  if (desc->kind < 0 || desc->kind > 2) return fail(TREBI_BAD_ARG, "sampler kind must be 0, 1 or 2");
  if (desc->kind >= 1 && !reward) return fail(TREBI_BAD_ARG, "guided sampling needs a reward ValueFunction");
  if (desc->kind == 2 && !cost) return fail(TREBI_BAD_ARG, "constrained sampling needs a cost ValueFunction");
  if (desc->kind >= 1 && desc->n_guide_steps < 1)
    return fail(TREBI_UNSUPPORTED_CONFIG, "n_guide_steps must be >= 1 (the reference leaves y undefined at 0)");
  if (desc->kind == 2 && sd.guide_type != 0 && sd.guide_type != 1) return fail(TREBI_BAD_ARG, "guide_type must be 0 or 1");
  if (unet->desc.kind != 0 || (reward && reward->desc.kind != 1) || (cost && cost->desc.kind != 1))
    return fail(TREBI_BAD_ARG, "network kinds do not match their roles");
  int D = sd.transition_dim, H = sd.horizon, T = sd.n_timesteps;
  if (unet->desc.transition_dim != D || unet->desc.horizon != H)
    return fail(TREBI_BAD_SHAPE, "TemporalUnet is [%d,%d], the sampler wants [%d,%d]", unet->desc.horizon,
                unet->desc.transition_dim, H, D);
  for (trebi_net* v : {reward, cost})
    if (v && (v->desc.transition_dim > D || v->desc.horizon != H))
      return fail(TREBI_BAD_SHAPE, "ValueFunction shape does not fit the trajectories");
  for (trebi_net* v : {unet, reward, cost})
    if (v && v->T < T) return fail(TREBI_BAD_SHAPE, "network time table (%d) shorter than n_timesteps (%d)", v->T, T);
  StepP tmp{};
  const float* dummy[4] = {plv, plv, plv, plv};
  if (int e = make_step(&sd, dummy, &tmp)) return e;
  HIPCHK(hipSetDevice(ctx->device));
  std::unique_ptr<trebi_sampler> s(new trebi_sampler);
  s->ctx = ctx; s->unet = unet; s->d = *desc; s->B = B;
  s->reward = desc->kind >= 1 ? reward : nullptr;
  s->cost = desc->kind == 2 ? cost : nullptr;
  s->dual = s->reward && s->cost && same_arch(s->reward, s->cost);
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  size_t per = (size_t)B * H * D;
  size_t n_tab = al(3 * (size_t)T), n_t = al((size_t)T * B * 2), n_u = al(unet->ws_floats(B)),
         n_r = s->reward ? al(s->reward->ws_floats(B)) : 0, n_c = s->cost ? al(s->cost->ws_floats(B)) : 0,
         n_x = al(per);
  size_t total = n_tab + n_t + n_u + n_r + n_c + 3 * n_x;
  HIPCHK(hipMalloc(&s->arena, total * sizeof(float)));
  float* p = (float*)s->arena;
  s->plv = p; s->c1 = p + T; s->c2 = p + 2 * T; p += n_tab;
  s->tbuf = (long long*)p; p += n_t;
  s->ws_u = p; p += n_u;
  s->ws_r = p; p += n_r;
  s->ws_c = p; p += n_c;
  s->x0 = p; p += n_x;
  s->rgrad = p; p += n_x;
  s->cgrad = p; p += n_x;
  HIPCHK(hipMemcpy(s->plv, plv, T * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->c1, c1, T * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->c2, c2, T * sizeof(float), hipMemcpyHostToDevice));
  for (int i = 0; i < T; ++i) HIPCHK(launch_fill_t(s->tbuf + (size_t)i * B, B, i, 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventCreate(&s->ev_a));
  HIPCHK(hipEventCreate(&s->ev_b));
  *out = s.release();
  return 0;
}
extern "C" void trebi_sampler_destroy(trebi_sampler* s) { delete s; }

// Resolve every network pass of the loop to launches for the caller's x / values / costs pointers.
static int ensure_plans(trebi_sampler* s, float* x, float* values, float* costs) {
  const void* key[3] = {x, values, costs};
  if (s->plans_valid && memcmp(key, s->plan_key, sizeof key) == 0) return 0;
  HIPCHK(hipDeviceSynchronize());          // a replay using the old table may still be in flight
  const trebi_sampler_desc& d = s->d;
  const int B = s->B, D = d.step.transition_dim;
  const char* e_chain = getenv("TREBI_CHAIN");
  const char* e_maxn = getenv("TREBI_CHAIN_MAXN");
  // Chains pay off while a launch is latency-bound; one workgroup per S candidates cannot fill the chip at
  // larger batches.  Measured (ms per planning call, chains vs none): 64 candidates 21.6 / 25.9, 128: 24.2 / 26.9,
  // 256: 29.7 (S=2) / 33.2, 512: 45.9 (S=2) / 48.0; wider layers (N > 64) and larger S lose.
  const char* e_maxb = getenv("TREBI_CHAIN_MAXB");
  const char* e_cs = getenv("TREBI_CHAIN_S");
  const int chain_max_b = e_maxb ? atoi(e_maxb) : 512;
  const int chain_S = (e_chain && atoi(e_chain) == 0) || B > chain_max_b ? 0 : (e_cs ? atoi(e_cs) : (B <= 160 ? 1 : B <= 448 ? 2 : 4));
  const int chain_max_n = e_maxn ? atoi(e_maxn) : 64;
  std::vector<RStage> tab;
  ParTab par;
  s->vf_grad.clear(); s->vf_nograd.clear(); s->unet_plan.clear();
  Binding bu;
  bu.x = x; bu.ldx = D; bu.out = s->x0; bu.ldo = D; bu.ws = s->ws_u;
  Resolver ru{s->unet, s->unet->offsets(B), bu, B};
  if (int e = build_plan({&s->unet->fwd}, {}, ru, nullptr, B, false, chain_S, chain_max_n, tab, par, s->unet_plan)) return e;
  if (s->reward) {
    Binding br, bc;
    br.x = x; br.ldx = D; br.grad = s->rgrad; br.ldg = D; br.ws = s->ws_r; br.y = values;
    bc = br; bc.grad = s->cgrad; bc.ws = s->ws_c; bc.y = costs;
    Resolver rr{s->reward, s->reward->offsets(B), br, B};
    if (s->cost) {
      Resolver rc{s->cost, s->cost->offsets(B), bc, B};
      if (s->dual) {
        s->vf_grad.resize(1); s->vf_nograd.resize(1);
        if (int e = build_plan({&s->cost->fwd, &s->cost->bwd}, {&s->reward->fwd, &s->reward->bwd}, rc, &rr, B, true,
                               chain_S, chain_max_n, tab, par, s->vf_grad[0])) return e;
        if (int e = build_plan({&s->cost->fwd}, {&s->reward->fwd}, rc, &rr, B, false, chain_S, chain_max_n, tab, par,
                               s->vf_nograd[0])) return e;
      } else {
        s->vf_grad.resize(2); s->vf_nograd.resize(2);
        if (int e = build_plan({&s->cost->fwd, &s->cost->bwd}, {}, rc, nullptr, B, true, chain_S, chain_max_n, tab, par, s->vf_grad[0])) return e;
        if (int e = build_plan({&s->cost->fwd}, {}, rc, nullptr, B, false, chain_S, chain_max_n, tab, par, s->vf_nograd[0])) return e;
        if (int e = build_plan({&s->reward->fwd, &s->reward->bwd}, {}, rr, nullptr, B, true, chain_S, chain_max_n, tab, par, s->vf_grad[1])) return e;
        if (int e = build_plan({&s->reward->fwd}, {}, rr, nullptr, B, false, chain_S, chain_max_n, tab, par, s->vf_nograd[1])) return e;
      }
    } else {
      s->vf_grad.resize(1); s->vf_nograd.resize(1);
      if (int e = build_plan({&s->reward->fwd, &s->reward->bwd}, {}, rr, nullptr, B, true, chain_S, chain_max_n, tab, par, s->vf_grad[0])) return e;
      if (int e = build_plan({&s->reward->fwd}, {}, rr, nullptr, B, false, chain_S, chain_max_n, tab, par, s->vf_nograd[0])) return e;
    }
  }
  if (tab.size() > s->dev_tab_cap) {
    if (s->dev_tab) (void)hipFree(s->dev_tab);
    s->dev_tab = nullptr;
    s->dev_tab_cap = tab.size() + 64;
    HIPCHK(hipMalloc((void**)&s->dev_tab, s->dev_tab_cap * sizeof(RStage)));
  }
  if (!tab.empty()) HIPCHK(hipMemcpy(s->dev_tab, tab.data(), tab.size() * sizeof(RStage), hipMemcpyHostToDevice));
  if (par.floats > s->dev_par_cap) {
    if (s->dev_par) (void)hipFree(s->dev_par);
    s->dev_par = nullptr;
    s->dev_par_cap = par.floats + 1024;
    HIPCHK(hipMalloc((void**)&s->dev_par, s->dev_par_cap * sizeof(float)));
  }
  for (const auto& f : par.fills) HIPCHK(launch_chain_params(f.p, s->dev_par + f.off, f.parN, f.p.T, f.t_stride, 0));
  HIPCHK(hipDeviceSynchronize());
  memcpy(s->plan_key, key, sizeof key);
  s->plans_valid = true;
  return 0;
}

// Enqueue the whole loop on `st` (diffusion.py:192-204 with the chosen sample function inlined).
static int enqueue_loop(trebi_sampler* s, const float* noise, const float* const* cond, const float* thr,
                        float* x, float* values, float* costs, float* chain, hipStream_t st, GemmProfile* prof) {
  const trebi_sampler_desc& d = s->d;
  const int B = s->B, T = d.step.n_timesteps, H = d.step.horizon, D = d.step.transition_dim;
  const size_t per = (size_t)B * H * D;
  StepP sp{};
  if (int e = make_step(&d.step, cond, &sp)) return e;
  HIPCHK(launch_init_x(sp, x, noise, B, chain, st));
  // (a fill kernel, not a memset node: tiny memset nodes did not replay reliably inside the captured graph)
  if (d.kind == 0 || d.kind == 1) HIPCHK(launch_fill_f(costs, B, 0.f, st));
  if (d.kind == 0) HIPCHK(launch_fill_f(values, B, 0.f, st));
  const int Dv = s->reward ? s->reward->desc.transition_dim : D;
  if (s->reward && Dv < D) {      // guides.py:13 reads a column slice; the other columns get zero gradient
    HIPCHK(launch_fill_f(s->rgrad, (int)per, 0.f, st));
    if (s->cost) HIPCHK(launch_fill_f(s->cgrad, (int)per, 0.f, st));
  }
  for (int i = T - 1; i >= 0; --i) {
    const long long* t = s->tbuf + (size_t)i * B;
    if (poison_on()) {                             // every pass of this step starts from NaN-filled workspaces
      if (int e = poison_ws(s->ws_u, s->unet->ws_floats(B), st)) return e;
      if (s->reward)
        if (int e = poison_ws(s->ws_r, s->reward->ws_floats(B), st)) return e;
      if (s->cost)
        if (int e = poison_ws(s->ws_c, s->cost->ws_floats(B), st)) return e;
    }
    if (d.kind >= 1) {
      const bool stopped = i < d.step.t_stopgrad;          // gradient is zeroed: x cannot change (functions.py:116)
      const bool dead = stopped && i != 0;                 // ... and y is only returned from the last step
      if (!(dead && d.skip_dead_evals)) {
        for (int k = 0; k < d.n_guide_steps; ++k) {
          const bool grad = !stopped;
          for (const Plan& pl : (grad ? s->vf_grad : s->vf_nograd))
            if (int e = exec_plan(pl, s->dev_tab, s->dev_par, i, t, st, prof)) return e;
          if (grad)
            HIPCHK(launch_guide_update(sp, x, s->rgrad, s->cost ? s->cgrad : nullptr, costs, thr, t, s->plv, B, st));
          else if (d.skip_dead_evals)
            break;                                         // same x => same y on every further iteration
        }
      }
    }
    if (int e = exec_plan(s->unet_plan, s->dev_tab, s->dev_par, i, t, st, prof)) return e;
    float* ch = chain ? chain + (size_t)(T - i) * per : nullptr;
    HIPCHK(launch_ddpm_update(sp, x, s->x0, noise + (size_t)(T - i) * per, t, s->plv, s->c1, s->c2, B, ch, st));
  }
  return 0;
}

// Set-up for a set of caller buffers: resolves every network pass to launches for these pointers (allocates the
// device table of chained stages, synchronises) and, for a graph sampler, captures + instantiates the loop.
extern "C" int trebi_sampler_bind(trebi_sampler* s, const float* noise, const float* const* cond, const float* thr,
                                  float* x, float* values, float* costs, float* chain) {
  if (!s || !noise || !x || !values || !costs) return fail(TREBI_BAD_ARG, "null argument");
  if (s->d.kind == 2 && !thr) return fail(TREBI_BAD_ARG, "constrained sampling needs cost_threshold");
  if (s->d.step.n_cond > 0 && !cond) return fail(TREBI_BAD_ARG, "cond is null");
  HIPCHK(hipSetDevice(s->ctx->device));
  s->bound = false;
  if (int e = ensure_plans(s, x, values, costs)) return e;
  const void* key[12] = {noise, thr, x, values, costs, chain};
  for (int k = 0; k < s->d.step.n_cond; ++k) key[6 + k] = cond[k];
  if (s->d.use_graph && (!s->exec || memcmp(key, s->key, sizeof key) != 0)) {
    if (s->exec) {
      HIPCHK(hipDeviceSynchronize());              // a launch of the old executable graph may still be in flight
      (void)hipGraphExecDestroy(s->exec);
      s->exec = nullptr;
    }
    hipStream_t cap;
    HIPCHK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    hipError_t e0 = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    int rc = 0;
    if (e0 == hipSuccess) rc = enqueue_loop(s, noise, cond, thr, x, values, costs, chain, cap, nullptr);
    hipError_t e1 = e0 == hipSuccess ? hipStreamEndCapture(cap, &graph) : e0;
    (void)hipStreamDestroy(cap);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e1 != hipSuccess) return fail(TREBI_HIP_ERROR, "graph capture failed: %s", hipGetErrorString(e1));
    hipError_t e2 = hipGraphInstantiate(&s->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) { s->exec = nullptr; return fail(TREBI_HIP_ERROR, "hipGraphInstantiate: %s", hipGetErrorString(e2)); }
  }
  memcpy(s->key, key, sizeof key);
  s->bound = true;
  return 0;
}

// Only enqueues (one hipGraphLaunch, or the loop's plain launches for use_graph = 0): no allocation, no
// synchronisation -- a use_graph = 0 sampler can be captured into the caller's own graph.
extern "C" int trebi_sampler_run(trebi_sampler* s, const float* noise, const float* const* cond, const float* thr,
                                 float* x, float* values, float* costs, float* chain, void* stream) {
  if (!s || !noise || !x || !values || !costs) return fail(TREBI_BAD_ARG, "null argument");
  if (s->d.step.n_cond > 0 && !cond) return fail(TREBI_BAD_ARG, "cond is null");
  const void* key[12] = {noise, thr, x, values, costs, chain};
  for (int k = 0; k < s->d.step.n_cond; ++k) key[6 + k] = cond[k];
  if (!s->bound || memcmp(key, s->key, sizeof key) != 0)
    return fail(TREBI_BAD_ARG, "the sampler is not bound to these buffers: call trebi_sampler_bind first "
                "(it resolves the launch plans and captures the loop for a given set of pointers)");
  hipStream_t st = (hipStream_t)stream;
  if (s->prof.enabled) {          // profiled run (diagnostics; creates timing events): plain launches bracketed by events
    s->prof.reset();
    HIPCHK(hipEventRecord(s->ev_a, st));
    if (int e = enqueue_loop(s, noise, cond, thr, x, values, costs, chain, st, &s->prof)) return e;
    HIPCHK(hipEventRecord(s->ev_b, st));
    if (s->prof.null_ev.empty())
      for (int i = 0; i < 16; ++i) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        s->prof.null_ev.push_back({a, b});
      }
    for (auto& e : s->prof.null_ev) {          // empty brackets, subtracted from every launch bracket
      HIPCHK(hipEventRecord(e.first, st));
      HIPCHK(hipEventRecord(e.second, st));
    }
    s->have_loop_time = true;
    return 0;
  }
  if (!s->d.use_graph) return enqueue_loop(s, noise, cond, thr, x, values, costs, chain, st, nullptr);
  HIPCHK(hipGraphLaunch(s->exec, st));
  return 0;
}

extern "C" int trebi_sampler_set_profiling(trebi_sampler* s, int enabled) {
  if (!s) return fail(TREBI_BAD_ARG, "null sampler");
  s->prof.enabled = enabled != 0;
  return 0;
}

extern "C" int trebi_sampler_get_profile(trebi_sampler* s, double* gemm_us_total, int64_t* gemm_launches,
                                         double* gemm_flops_total, double* loop_us_total) {
  if (!s) return fail(TREBI_BAD_ARG, "null sampler");
  if (!s->have_loop_time) return fail(TREBI_BAD_ARG, "no profiled run recorded");
  HIPCHK(hipEventSynchronize(s->ev_b));
  double us = 0, fl = 0, null_us = 0;
  {
    std::vector<double> nu;
    for (auto& e : s->prof.null_ev) {
      HIPCHK(hipEventSynchronize(e.second));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
      nu.push_back(1e3 * ms);
    }
    std::sort(nu.begin(), nu.end());
    // an empty bracket exposes both marker packets; around a kernel one of them overlaps the kernel's own
    // dispatch, so half is charged (calibrated: rocprofv3 kernel durations then agree within a few %)
    if (!nu.empty()) null_us = 0.5 * nu[nu.size() / 2];
  }
  for (size_t i = 0; i < s->prof.used; ++i) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->prof.ev[i].first, s->prof.ev[i].second));
    us += std::max(0.0, 1e3 * ms - null_us);
    fl += s->prof.flops[i];
  }
  if (const char* path = getenv("TREBI_PROFILE_DUMP")) {   // diagnostics: per-layer totals of the profiled run
    std::map<std::string, std::array<double, 3>> agg;
    for (size_t i = 0; i < s->prof.used && i < s->prof.names.size(); ++i) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, s->prof.ev[i].first, s->prof.ev[i].second));
      auto& a = agg[s->prof.names[i]];
      a[0] += 1; a[1] += 1e3 * ms; a[2] += s->prof.flops[i];
    }
    if (FILE* f = fopen(path, "w")) {
      fprintf(f, "launches\ttotal_us\tavg_us\tTFLOPs\tlayer\n");
      for (auto& kv : agg)
        fprintf(f, "%d\t%.1f\t%.2f\t%.2f\t%s\n", (int)kv.second[0], kv.second[1], kv.second[1] / kv.second[0],
                kv.second[2] / (kv.second[1] * 1e-6) / 1e12, kv.first.c_str());
      fclose(f);
    }
  }
  float lms = 0;
  HIPCHK(hipEventElapsedTime(&lms, s->ev_a, s->ev_b));
  if (gemm_us_total) *gemm_us_total = us;
  if (gemm_launches) *gemm_launches = (int64_t)s->prof.used;
  if (gemm_flops_total) *gemm_flops_total = fl;
  if (loop_us_total) *loop_us_total = 1e3 * lms;
  return 0;
}

// FLOP accounting hook used by bench.py / DESIGN.md (algorithmic, zero-padding taps excluded).
extern "C" double trebi_net_flops(const trebi_net* net, int backward) {
  return net ? (backward ? net->flops_bwd : net->flops_fwd) : 0.0;
}

// ------------------------------------------------------------------------------------
// the training step: host side
// ------------------------------------------------------------------------------------
#ifdef TREBI_STAMPS
extern "C" int trebi_debug_read_wstamps(unsigned long long* out);
#endif
#include "trainer.inl"
