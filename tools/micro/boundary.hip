// Micro-benchmark: cost of a dependent kernel boundary (eager vs hipGraph) for trivial kernels of
// several grid sizes, and of a kernel that only reads a few KB that every block shares.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1025) p[0] = 1; }
__global__ __launch_bounds__(256) void shared_read_kernel(const double* __restrict__ m, double* out, int n) {
  __shared__ double s[1024];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += m[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[255];
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int* d; CK(hipMalloc(&d, 4096));
  double *m, *o; CK(hipMalloc(&m, 1 << 20)); CK(hipMalloc(&o, 1 << 20)); CK(hipMemset(m, 0, 1 << 20));
  const int N = 2000;
  for (int grid : {1, 256, 625, 2048}) {
    for (int mode = 0; mode < 3; ++mode) {   // 0 empty eager, 1 empty graph, 2 shared-read graph
      auto enqueue = [&](int reps) {
        for (int i = 0; i < reps; ++i) {
          if (mode == 2) hipLaunchKernelGGL(shared_read_kernel, dim3(grid), dim3(256), 0, st, m, o, 1024);
          else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d);
        }
      };
      hipGraphExec_t exec = nullptr;
      if (mode >= 1) {
        hipGraph_t g; CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(100);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
      }
      auto run = [&]() { if (mode >= 1) { for (int i = 0; i < N / 100; ++i) hipGraphLaunch(exec, st); } else enqueue(N); };
      run(); CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::steady_clock::now();
      run(); CK(hipStreamSynchronize(st));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("grid %5d  %-18s %7.2f us per kernel\n", grid, mode == 0 ? "empty eager" : mode == 1 ? "empty graph" : "shared-read graph", us / N);
      if (exec) CK(hipGraphExecDestroy(exec));
    }
  }
  return 0;
}
