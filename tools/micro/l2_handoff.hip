// Diagnostic microbenchmark: the price of handing a candidate tile from one layer to the next INSIDE a launch (the column-tile
// workgroups of one candidate tile write their slices, signal a per-candidate-tile arrival counter, wait for their mates and read
// the whole tile back) against the same hand-off done by a kernel boundary.  This is the step the "one launch per
// ResidualTemporalBlock" design for the 128/256-channel layers would put where a launch boundary is today.  Not part of the product.
//
//   l2_handoff [column tiles = 8] [floats per slice = 512] [rounds = 200]
//
// grid = (column tiles, candidate tiles) with x fastest, as conv_gemm_kernel launches: the mates of one candidate tile sit on
// different XCDs, so the slices cross L2s (release: write back; acquire: invalidate).  Every spin is bounded; a round that gives up
// sets a flag and the kernel runs on to its end (results are then reported as failed, nothing hangs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int SPIN_LIMIT = 1 << 16;

__device__ inline float slice_value(int round, int ytile, int xt, int i) { return (float)((round * 131 + ytile * 17 + xt * 5 + i) & 1023); }

// MODE 0: plain stores and loads around agent-scope release / acquire fences (L2 write-back and invalidate per workgroup).
// MODE 1: no fence: the slices are written and read with agent-scope relaxed atomics (write-through stores, L2-bypassing loads),
//         ordered by s_waitcnt vmcnt(0) in front of a relaxed counter increment.
// one launch, `rounds` hand-offs inside it
template <int MODE>
__global__ __launch_bounds__(512) void handoff_inside(float* buf, unsigned* counters, int rounds, int nslice, float* sink,
                                                      unsigned long long* cyc, int* failed) {
  const int xt = blockIdx.x, yt = blockIdx.y, nx = gridDim.x, ny = gridDim.y;
  const size_t tile = (size_t)nx * nslice;
  float acc = 0.f;
  int bad = 0;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    float* dst = buf + ((size_t)(r & 1) * ny + yt) * tile;
    for (int i = threadIdx.x; i < nslice; i += blockDim.x) {
      if (MODE == 0) dst[(size_t)xt * nslice + i] = slice_value(r, yt, xt, i);
      else __hip_atomic_store(dst + (size_t)xt * nslice + i, slice_value(r, yt, xt, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 0) __threadfence();                    // release at agent scope: the slice leaves this XCD's L2
    else __builtin_amdgcn_s_waitcnt(0);                // the write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* c = counters + (size_t)r * ny + yt;
      if (MODE == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int it = 0;
      if (MODE == 0) while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nx && ++it < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
      else while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nx && ++it < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
      if (it >= SPIN_LIMIT) *failed = 1;
    }
    __syncthreads();
    if (MODE == 0) __threadfence();                    // acquire: stale lines of the mates' slices are dropped
    for (size_t i = threadIdx.x; i < tile; i += blockDim.x) {
      const float v = MODE == 0 ? dst[i] : __hip_atomic_load(dst + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int x2 = (int)(i / nslice), i2 = (int)(i % nslice);
      bad |= v != slice_value(r, yt, x2, i2);
      acc += v;
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (bad) *failed = 2;
  if (threadIdx.x == 0) {
    sink[blockIdx.y * nx + xt] = acc;
    cyc[blockIdx.y * nx + xt] = t1 - t0;
  }
}

// the same write / read split over launches: the kernel boundary is the hand-off
__global__ __launch_bounds__(512) void handoff_boundary(float* buf, int r, int nslice, float* sink, int* failed) {
  const int xt = blockIdx.x, yt = blockIdx.y, nx = gridDim.x, ny = gridDim.y;
  const size_t tile = (size_t)nx * nslice;
  float acc = 0.f;
  int bad = 0;
  if (r > 0) {                                         // read what the previous launch wrote
    const float* src = buf + ((size_t)((r - 1) & 1) * ny + yt) * tile;
    for (size_t i = threadIdx.x; i < tile; i += blockDim.x) {
      const float v = src[i];
      bad |= v != slice_value(r - 1, yt, (int)(i / nslice), (int)(i % nslice));
      acc += v;
    }
  }
  float* dst = buf + ((size_t)(r & 1) * ny + yt) * tile;
  for (int i = threadIdx.x; i < nslice; i += blockDim.x) dst[(size_t)xt * nslice + i] = slice_value(r, yt, xt, i);
  if (bad) *failed = 2;
  if (threadIdx.x == 0) sink[blockIdx.y * nx + xt] = acc;
}

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 8, nslice = argc > 2 ? atoi(argv[2]) : 512, rounds = argc > 3 ? atoi(argv[3]) : 200;
  if (nx < 1 || nx > 16 || nslice < 64 || nslice > 16384 || rounds < 1 || rounds > 2000) { printf("bad arguments\n"); return 1; }
  for (int ny : {16, 32 / (nx > 8 ? 2 : 1)}) {         // 128 and 256 workgroups at 8 column tiles: every workgroup is resident (1 per CU)
    if (nx * ny > 256) continue;
    float *buf, *sink; unsigned* counters; unsigned long long* cyc; int* failed;
    CHECK(hipMalloc(&buf, 2ull * ny * nx * nslice * sizeof(float)));
    CHECK(hipMalloc(&sink, nx * ny * sizeof(float)));
    CHECK(hipMalloc(&counters, (size_t)rounds * ny * sizeof(unsigned)));
    CHECK(hipMalloc(&cyc, nx * ny * sizeof(unsigned long long)));
    CHECK(hipMalloc(&failed, sizeof(int)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms_in[2] = {0.f, 0.f}, ms_b = 0.f, ms_g = 0.f;
    int f_in[2] = {0, 0}, f_b = 0;
    unsigned long long cmax[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {              // the last repetition is the one reported
        CHECK(hipMemset(counters, 0, (size_t)rounds * ny * sizeof(unsigned)));
        CHECK(hipMemset(failed, 0, sizeof(int)));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(handoff_inside<0>, dim3(nx, ny), dim3(512), 0, 0, buf, counters, rounds, nslice, sink, cyc, failed);
        else hipLaunchKernelGGL(handoff_inside<1>, dim3(nx, ny), dim3(512), 0, 0, buf, counters, rounds, nslice, sink, cyc, failed);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms_in[mode], e0, e1));
        CHECK(hipMemcpy(&f_in[mode], failed, sizeof(int), hipMemcpyDeviceToHost));
        if (f_in[mode]) break;
      }
      std::vector<unsigned long long> hc(nx * ny);
      CHECK(hipMemcpy(hc.data(), cyc, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      for (auto c : hc) cmax[mode] = c > cmax[mode] ? c : cmax[mode];
    }
    // boundary version: plain launches, then the same launches replayed as a graph (what the product's loop does)
    CHECK(hipMemset(failed, 0, sizeof(int)));
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(handoff_boundary, dim3(nx, ny), dim3(512), 0, 0, buf, r, nslice, sink, failed);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventElapsedTime(&ms_b, e0, e1));
    }
    CHECK(hipMemcpy(&f_b, failed, sizeof(int), hipMemcpyDeviceToHost));
    hipStream_t st; hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamCreate(&st));
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(handoff_boundary, dim3(nx, ny), dim3(512), 0, st, buf, r, nslice, sink, failed);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventRecord(e0, st));
      CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventElapsedTime(&ms_g, e0, e1));
    }
    printf("%d column tiles x %d candidate tiles (%d workgroups), %d floats per slice (tile %.1f KB), %d rounds\n", nx, ny, nx * ny, nslice,
           nx * nslice * 4 / 1024.0, rounds);
    for (int mode = 0; mode < 2; ++mode)
      printf("  inside one launch, %s : %7.3f us per hand-off (events)  %7.3f us by the slowest workgroup's clock  %s\n",
             mode ? "write-through slices, no fence" : "agent-scope release / acquire ", 1e3 * ms_in[mode] / rounds, cmax[mode] / 100.0 / rounds,
             f_in[mode] == 0 ? "ok" : f_in[mode] == 1 ? "SPIN LIMIT HIT" : "STALE DATA READ");
    printf("  kernel boundaries : %7.3f us per launch (stream)   %7.3f us per launch (graph replay)  %s\n", 1e3 * ms_b / rounds, 1e3 * ms_g / rounds,
           f_b ? "WRONG DATA" : "ok");
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
    (void)hipFree(buf); (void)hipFree(sink); (void)hipFree(counters); (void)hipFree(cyc); (void)hipFree(failed);
  }
  return 0;
}
