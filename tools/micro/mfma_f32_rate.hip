// Diagnostic microbenchmark: what v_mfma_f32_16x16x4_f32 sustains on gfx950 with 1 / 2 waves per SIMD, bare and with the
// operand patterns of conv_gemm_kernel's K loop.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GL __attribute__((address_space(1)))
// 0: bare; 1: A from LDS per step (read, wait, multiply); 2: A one step ahead; 3: as 1 + weights streamed from global in
// half blocks of 4 steps (8 float4 per lane per half block, consumed one block later), like the large-batch loop
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* wts, int iters, int rnd) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // argv[2] = 1: operands with random mantissas (the power drawn by the matrix pipe depends on the data: zeros and
  // near-constants toggle few bits)
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    lds[i] = rnd ? (float)(int)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f : 1.0f + i * 1e-6f;
  }
  __syncthreads();
  f32x4 acc[4][2];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 w[8][2];
  for (int u = 0; u < 8; ++u) for (int n = 0; n < 2; ++n) w[u][n] = (f32x4){1.f, 2.f, 3.f, 4.f} * (float)(lane + u + n) * (rnd ? 1.37e-3f : 1.f);
  const f32x4* base = (const f32x4*)lds + lane;
  const GL f32x4* wp = (const GL f32x4*)wts + wave * 32 + lane;      // 256 columns x float4 per step
  f32x4 a[4], an[4];
  for (int m = 0; m < 4; ++m) { a[m] = (f32x4){1.f, 1.f, 1.f, 1.f} * (float)m; an[m] = a[m]; }
  if (MODE == 2) for (int m = 0; m < 4; ++m) an[m] = base[m * 64];
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int off = (((it + u) & 15) * 256);
      if (MODE >= 5) {      // a step the scheduler cannot reorder across (the real loop branches per step): read, wait, multiply
        __builtin_amdgcn_sched_barrier(0);
        for (int m = 0; m < 4; ++m) a[m] = base[off + m * 64];
      }
      if (MODE == 1 || MODE == 3 || MODE == 4) { for (int m = 0; m < 4; ++m) a[m] = base[off + m * 64]; }
      if (MODE == 2) { for (int m = 0; m < 4; ++m) a[m] = an[m]; for (int m = 0; m < 4; ++m) an[m] = base[off + m * 64]; }
      if (MODE == 4) {      // tile-major order of the position-major tiles: an accumulator comes back after ONE other MFMA
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], w[u][n][j], acc[m][n], 0, 0, 0);
      } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], w[u][n][j], acc[m][n], 0, 0, 0);
      }
      if (MODE == 6) {      // + every accumulator copied after the step (what the register allocator leaves at the joins)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("v_mov_b32 %0, %0" : "+v"(acc[m][n][e]));
      }
      if (MODE >= 5) __builtin_amdgcn_sched_barrier(0);
      if ((MODE == 3 || MODE == 4 || MODE >= 5) && (u & 3) == 3) {
        const int h = u >> 2;
        const int step = (it + 8 + 4 * h) % 320;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int n = 0; n < 2; ++n) w[4 * h + q][n] = wp[(size_t)(step + q) * 1024 + n * 16];
      }
    }
  }
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
static int g_rnd = 0;
static int g_iters = 4000;        // argv[1]: longer runs show the SUSTAINED rate (clocks under power management)
template <int MODE>
void run(const char* name, int threads, float* d, const float* w) {
  const int iters = g_iters, grid = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, d, w, 96, g_rnd);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, d, w, iters, g_rnd);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * (threads / 64) * iters * 32.0 * 2048.0;
  printf("%-44s %d waves/SIMD: %7.1f TFLOP/s (%.1f %% of 157.3) in %.1f ms\n", name, threads / 256, fl / ms / 1e9,
         fl / ms / 1e9 / 157.3 * 100, ms);
}
int main(int argc, char** argv) {
  if (argc > 1) g_iters = atoi(argv[1]) & ~7;
  if (argc > 2) g_rnd = atoi(argv[2]);
  float *d, *w; (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&w, (size_t)330 * 1024 * 16); (void)hipMemset(w, 0, (size_t)330 * 1024 * 16);
  if (g_rnd) {
    std::vector<float> hw((size_t)330 * 1024 * 4);
    unsigned h = 12345u;
    for (auto& v : hw) { h = h * 1664525u + 1013904223u; v = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  }
  run<0>("bare", 256, d, w); run<0>("bare", 512, d, w);
  run<1>("A from LDS, read-wait-multiply", 256, d, w); run<1>("A from LDS, read-wait-multiply", 512, d, w);
  run<2>("A from LDS, one step ahead", 256, d, w); run<2>("A from LDS, one step ahead", 512, d, w);
  run<3>("A from LDS + weights from L2 (half blocks)", 256, d, w); run<3>("A from LDS + weights from L2 (half blocks)", 512, d, w);
  run<4>("same, tile-major MFMA order (PM tiles)", 256, d, w); run<4>("same, tile-major MFMA order (PM tiles)", 512, d, w);
  run<5>("fenced steps (read, wait, multiply) + L2 weights", 256, d, w); run<5>("fenced steps (read, wait, multiply) + L2 weights", 512, d, w);
  run<6>("same + all accumulators copied per step", 256, d, w); run<6>("same + all accumulators copied per step", 512, d, w);
  (void)hipDeviceSynchronize();
  return 0;
}
