// Micro-benchmark: plain float4 read rate vs buffer size (Infinity-Cache resident ... HBM only),
// launches from a graph (times per launch include the ~1.6 us boundary).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(512) void chunk_read(const f32x4* __restrict__ p, size_t n4, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t b = (size_t)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
  size_t i = b + threadIdx.x;
  for (; i + (UNROLL - 1) * 512 < e; i += UNROLL * 512) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 512];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < e; i += 512) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  float* o; hipMalloc(&o, 64);
  for (size_t mb : {40, 82, 164, 256, 512, 1024, 3200}) {
    const size_t bytes = mb << 20;
    float* a; if (hipMalloc(&a, bytes) != hipSuccess) { printf("alloc %zu MB failed\n", mb); continue; }
    hipMemset(a, 0, bytes);
    const size_t n4 = bytes / 16;
    for (int grid : {512, 2048}) {
      hipGraph_t gr; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      const int reps = mb <= 256 ? 40 : 8;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chunk_read<8>, dim3(grid), dim3(512), 0, st, (const f32x4*)a, n4, o);
      hipStreamEndCapture(st, &gr); hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
      for (int i = 0; i < 2; ++i) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
      hipStreamSynchronize(st);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5 * reps);
      printf("%5zu MB, %4d WGs: %8.2f us per read -> %5.2f TB/s\n", mb, grid, us, bytes / us / 1e6);
      hipGraphExecDestroy(ge); hipGraphDestroy(gr);
    }
    hipFree(a);
  }
  return 0;
}
