// Device helpers shared by the kernel files.
#pragma once
#include "common.h"

namespace trebi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef TREBI_STAMPS          // diagnostic build only: per-phase cycle stamps of workgroup (0,0,0), wave 0
extern __device__ unsigned long long g_stamps[32];
#define STAMP(i)                                                                                  \
  do {                                                                                            \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {              \
      g_stamps[i] = __builtin_amdgcn_s_memtime();                                                 \
      g_stamps[16 + i] = __builtin_amdgcn_s_memrealtime();                                        \
    }                                                                                             \
  } while (0)
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


}  // namespace trebi
