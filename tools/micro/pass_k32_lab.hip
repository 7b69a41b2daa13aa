// Micro-benchmark of the k > 16 streaming pass at the c4 / c5 shapes (pass_body_wide of the product, included as it is):
// tiles per workgroup, splits, residency.   hipcc --offload-arch=gfx950 -O3 -I../../resnmtf_amd/csrc -I../../include
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "resnmtf_hip.h"
#include "resnmtf_kernels.hip.inc"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// Measured with the first version of this file (one launch each, c5 = 50000 x 8000 k = 64, c4 = 20000 x 4000 k = 32):
//   one 64-column tile per workgroup, B pieces from L2 per wave:  c5 X.G 439 us, Xt.F 369 us; c4 71-76 / 66-72 us
//   wide (8 tiles per workgroup, B block staged once in LDS):     c5 X.G 343 us, Xt.F 321 us; c4 59 / 62 us
//   wide with 294 workgroups (1.15 rounds of 256 CUs): 468 us; two-piece products: 295 / 268 us (not f32-grade)
template <int NT, int MINW>
__global__ __launch_bounds__(512, MINW) void lab_wide(const float* __restrict__ A, size_t tile_stride, int ntiles, int tw,
                                                      const unsigned short* __restrict__ Bk, float* __restrict__ P, int cols_pad,
                                                      int rows_pad, int rps) {
  extern __shared__ __attribute__((aligned(16))) u32x4_t lds[];
  const int ntg = (ntiles + tw - 1) / tw, tg = blockIdx.x % ntg, split = blockIdx.x / ntg;
  const int r_begin = split * rps;
  if (tw == 8) pass_body_wide<NT, 1>(A, tile_stride, ntiles, tg * 8, Bk, r_begin, min(r_begin + rps, rows_pad), lds, P + (size_t)split * cols_pad * 16 * NT);
  else pass_body_wide<NT, 2>(A, tile_stride, ntiles, tg * 4, Bk, r_begin, min(r_begin + rps, rows_pad), lds, P + (size_t)split * cols_pad * 16 * NT);
}

__global__ void fill_kernel(float* x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned int h = (unsigned int)(i * 2654435761u) ^ (unsigned int)(i >> 13);
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    x[i] = 1e-4f * (1.0f + (float)(h & 0xFFFF) / 65536.0f);
  }
}
__global__ void fill_bk_kernel(unsigned short* bk, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    bk[i] = (unsigned short)(0x3a00u + ((i * 40503u) & 0xFF));
}

// groups of 10 launches back to back: the first groups of a process run while the clocks ramp, a long run shows what the
// chip sustains (LAB_GROUPS, default 3; the last group is reported, all are printed with LAB_VERBOSE)
template <typename F>
double time_us(F&& launch, hipStream_t st) {
  const int groups = getenv("LAB_GROUPS") ? atoi(getenv("LAB_GROUPS")) : 3;
  double us = 0.0;
  for (int g = 0; g < groups; ++g) {
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; ++r) launch();
    hipStreamSynchronize(st);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
    if (getenv("LAB_VERBOSE")) printf("   group %d: %.1f us\n", g, us);
  }
  return us;
}

template <int NT, int MINW>
int run_wide(const char* tag, const float* A, size_t ts, int ntiles, int rows_pad, const unsigned short* Bk, float* P, int tw, int rps, hipStream_t st) {
  const int cols_pad = ntiles * 64, ns = (rows_pad + rps - 1) / rps, ntg = (ntiles + tw - 1) / tw;
  const size_t smem = wide_smem_bytes(NT);
  auto fn = lab_wide<NT, MINW>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const double us = time_us([&] { hipLaunchKernelGGL(fn, dim3(ntg * ns), dim3(512), smem, st, A, ts, ntiles, tw, Bk, P, cols_pad, rows_pad, rps); }, st);
  printf("%s wide NT %d tw %d minw %d rps %5d (%5d WGs): %8.1f us -> %5.2f TB/s\n", tag, NT, tw, MINW, rps, ntg * ns, us,
         ((double)rows_pad * cols_pad * 4 + 4.0 * (rows_pad + cols_pad) * 16 * NT) / us / 1e6);
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *A, *P; unsigned short* Bk;
  const size_t afl = (size_t)50048 * 8064 + (size_t)800 * 64;          // either orientation of c5 incl. tile pad rows
  CK(hipMalloc(&A, afl * 4));
  CK(hipMalloc(&Bk, (size_t)50048 * 64 * 6));
  CK(hipMalloc(&P, (size_t)16 * 50048 * 64 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, A, afl);
  hipLaunchKernelGGL(fill_bk_kernel, dim3(1024), dim3(256), 0, st, Bk, (size_t)50048 * 64 * 3);
  CK(hipStreamSynchronize(st));
  // c5 (k = 64): X.G  A = Xt32: 782 tiles of [8000 (+1)][64];  Xt.F  A = X32: 125 tiles of [50048 (+1)][64]
  run_wide<4, 2>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 8, 1600, st);
  run_wide<4, 2>("c5 xtf", A, (size_t)50049 * 64, 125, 50048, Bk, P, 8, 3136, st);
  run_wide<4, 2>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 8, 1600, st);
  run_wide<4, 2>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 8, 640, st);
  run_wide<4, 2>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 4, 2688, st);
  run_wide<4, 2>("c5 xtf", A, (size_t)50049 * 64, 125, 50048, Bk, P, 8, 3136, st);
  run_wide<4, 2>("c5 xtf", A, (size_t)50049 * 64, 125, 50048, Bk, P, 4, 6272, st);
  // c4 (k = 32): X.G  313 tiles of [4032 (+1)][64];  Xt.F  63 tiles of [20032 (+1)][64]
  run_wide<2, 2>("c4 xg ", A, (size_t)4033 * 64, 313, 4032, Bk, P, 8, 672, st);
  run_wide<2, 4>("c4 xg ", A, (size_t)4033 * 64, 313, 4032, Bk, P, 8, 320, st);
  run_wide<2, 2>("c4 xg ", A, (size_t)4033 * 64, 313, 4032, Bk, P, 4, 1344, st);
  run_wide<2, 2>("c4 xtf", A, (size_t)20033 * 64, 63, 20032, Bk, P, 4, 1280, st);
  run_wide<2, 4>("c4 xtf", A, (size_t)20033 * 64, 63, 20032, Bk, P, 4, 640, st);
  run_wide<2, 2>("c4 xtf", A, (size_t)20033 * 64, 63, 20032, Bk, P, 8, 640, st);
  return 0;
}
