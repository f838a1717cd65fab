// Device helpers shared by the kernel files.
#pragma once
#include "common.h"

namespace trebi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every pointer the kernels receive (inside by-value structs or descriptor tables) points to global
// memory, but the compiler only sees generic pointers and would emit FLAT loads / stores, which count
// against lgkmcnt as well as vmcnt -- every wait for LDS data would then also wait for all outstanding
// global traffic.  These accessors cast to the global address space so that global_load / global_store
// are emitted.
#define TREBI_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float gld(const float* p) { return *(const TREBI_GLOBAL float*)p; }
__device__ __forceinline__ long long gldll(const long long* p) { return *(const TREBI_GLOBAL long long*)p; }
__device__ __forceinline__ f32x4 gld4(const float* p) { return *(const TREBI_GLOBAL f32x4*)p; }
__device__ __forceinline__ void gst(float* p, float v) { *(TREBI_GLOBAL float*)p = v; }
__device__ __forceinline__ void gst4(float* p, f32x4 v) { *(TREBI_GLOBAL f32x4*)p = v; }

#ifdef TREBI_STAMPS          // diagnostic build only: per-phase cycle stamps of workgroup (0,0,0), wave 0
extern __device__ unsigned long long g_stamps[32];
extern __device__ unsigned long long g_wave_stamps[8][16];     // the same per wave of workgroup (0,0,0)
#define STAMP(i)                                                                                  \
  do {                                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {              \
      g_stamps[i] = __builtin_amdgcn_s_memtime();                                                 \
      g_stamps[16 + i] = __builtin_amdgcn_s_memrealtime();                                        \
    }                                                                                             \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0 && threadIdx.x < 512)  \
      g_wave_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime();                          \
  } while (0)
#else
#define STAMP(i)
#endif

// Mish and its derivative from ONE exponential: with e = exp(x), n = e*(e+2):
//   tanh(softplus(x)) = n/(n+2),   sigmoid(x) = e/(1+e).
// torch's softplus switches to the identity above 20 (tanh(20) == 1 in f32), so x > 20 returns x / 1: the exponent is
// clamped there (no overflow) and the result selected, no branch.  The quotients use v_rcp_f32 (1 ulp) times the numerator:
// `a / b` and __fdividef both compile to the IEEE division sequence (div_scale / rcp / 4 fma / div_fmas / div_fixup, ~10
// instructions each), which made Mish ~40 instructions per element and the GroupNorm epilogue as expensive as the K loop on
// the 32- and 64-channel layers.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float mish_f(float x) {
  float e = __expf(fminf(x, 20.f));
  float n = e * (e + 2.f);
  float m = x * (n * fast_rcp(n + 2.f));
  return x > 20.f ? x : m;
}
__device__ __forceinline__ float mish_grad_f(float x) {
  float e = __expf(fminf(x, 20.f));
  float n = e * (e + 2.f);
  float th = n * fast_rcp(n + 2.f);
  float sg = e * fast_rcp(1.f + e);
  float d = th + x * sg * (1.f - th * th);
  return x > 20.f ? 1.f : d;
}
__device__ __forceinline__ void mish_both(float x, float* m, float* dm) {
  float e = __expf(fminf(x, 20.f));
  float n = e * (e + 2.f);
  float th = n * fast_rcp(n + 2.f);
  float sg = e * fast_rcp(1.f + e);
  const bool big = x > 20.f;
  *m = big ? x : x * th;
  *dm = big ? 1.f : th + x * sg * (1.f - th * th);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- bf16x3: fp32 operands split into two bf16 halves, x = hi + lo (+ O(2^-17 |x|)) ---------------------------
// A product is then hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): 3 instructions of 16 cycles
// for K = 32 instead of 8 of 32 cycles on the fp32 matrix pipe, at a relative error of ~2^-16 per product.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned bf16_rne(float x) {          // float -> bf16 bits, round to nearest even
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split_bf16(float x, unsigned* hi, unsigned* lo) {
  const unsigned h = bf16_rne(x);
  *hi = h;
  *lo = bf16_rne(x - __uint_as_float(h << 16));
}
// four consecutive channels of one row -> the row's hi plane (bf16 [Cp]) and lo plane (bf16 [Cp] behind it)
__device__ __forceinline__ void store4_bf16x2(float* row, int c, int Cp, f32x4 v) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_bf16(v[q], &h[q], &l[q]);
  unsigned short* b = (unsigned short*)row;
  *(uint2*)(b + c) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  *(uint2*)(b + Cp + c) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}
__device__ __forceinline__ bf16x8 as_bf16x8(f32x4 v) {
  union { f32x4 f; bf16x8 b; } c;
  c.f = v;
  return c.b;
}

// n / d with a host-computed multiplier (ConvGeo): one mul-hi.
struct FastDiv {
  unsigned d, m;
  __device__ FastDiv(unsigned dd, unsigned mm) : d(dd), m(mm) {}
  __device__ __forceinline__ unsigned div(unsigned n) const { return m == 0 ? n : __umulhi(n, m); }
};

// Keep a wave-uniform value in scalar registers: without this the compiler re-reads kernel-argument
// fields with an s_load + s_waitcnt in front of every use (hundreds of stall cycles per use).
#define PIN(x) asm volatile("" : "+s"(x))

// apply_conditioning (helpers.py:142-145): is element (h, d) of candidate b overwritten by a condition?
__device__ __forceinline__ bool conditioned(const StepP& s, int h, int d, int b, float* val) {
  if (d < s.act || d >= s.act + s.obs) return false;
  for (int k = 0; k < s.n_cond; ++k)
    if (s.cond_index[k] == h) {
      *val = s.cond[k][(size_t)b * s.obs + (d - s.act)];
      return true;
    }
  return false;
}

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
__device__ __forceinline__ float hsum(f32x4 v) { return (v.x + v.y) + (v.z + v.w); }

// float4 (VEC) or 4 guarded scalars from a row with `left` valid floats starting at sp
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* sp, int left) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (left > 0) v = gld4(sp);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < left) v[e] = gld(sp + e);
  }
  return v;
}


// A launch / stage descriptor spread over the lanes of 4 VGPRs (dword i in lane i & 63 of register i >> 6);
// a field is then ONE v_readlane_b32 with a constant lane instead of a scalar load + wait at every place
// the compiler decides to (re)read it.
struct DV { unsigned v[4]; };
template <int IDX>
__device__ __forceinline__ unsigned dget(const DV& d) {
  return (unsigned)__builtin_amdgcn_readlane((int)d.v[IDX >> 6], IDX & 63);
}
template <typename T, int IDX>
__device__ __forceinline__ T dfield(const DV& d) {
  if constexpr (sizeof(T) == 4) {
    union { unsigned u; T t; } c;
    c.u = dget<IDX>(d);
    return c.t;
  } else {
    static_assert(sizeof(T) == 8, "descriptor fields are 4 or 8 bytes");
    union { unsigned long long u; T t; } c;
    c.u = (unsigned long long)dget<IDX>(d) | ((unsigned long long)dget<IDX + 1>(d) << 32);
    return c.t;
  }
}

}  // namespace trebi
