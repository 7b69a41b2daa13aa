// Micro-benchmark: achievable read rate for an 80 MB fp32 buffer (c2's X32 or Xt32), alternating
// between two such buffers (as the sweep does) so that both must stay resident in the Infinity Cache.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ p, size_t n4, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

int main() {
  const size_t bytes = (size_t)10048 * 2048 * 4;   // 82 MB
  float *a, *b, *o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t n4 = bytes / 16;
  for (int grid : {256, 512, 1024, 2048, 4096, 8192}) {
    for (int two = 0; two < 2; ++two) {
      auto run = [&](int reps) {
        for (int r = 0; r < reps; ++r) {
          hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, st, (const f32x4*)a, n4, o);
          hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, st, (const f32x4*)(two ? b : a), n4, o);
        }
      };
      run(20); CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::steady_clock::now();
      run(200); CK(hipStreamSynchronize(st));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 400;
      printf("grid %5d  %s: %6.2f us per 82 MB read = %5.2f TB/s (incl. ~2.6 us eager launch gap)\n", grid,
             two ? "alternating 2 buffers" : "same buffer         ", us, bytes / us / 1e6);
    }
  }
  return 0;
}
