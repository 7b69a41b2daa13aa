// Micro-benchmark: streaming pass at HBM-resident sizes (c4: 20000x4000 k=32, c5: 50000x8000 k=64).
// Variants of waves per workgroup, unroll depth and non-temporal loads; full body (loads, B operand,
// MFMA, tree reduction, partial store).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int NW, int UNROLL, bool NTL>
__global__ __launch_bounds__(64 * NW) void pass_k(const float* __restrict__ A, int lda, int ntiles, const float* __restrict__ B,
                                                  float* __restrict__ P, int cols_pad, int rows_pad, int rows_per_split) {
  constexpr int KP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int r_begin = split * rows_per_split;
  const int r_end = min(r_begin + rows_per_split, rows_pad);
  const int nsteps = (r_end - r_begin) / (4 * NW);
  f32x4 acc[4][NT];
  for (int jj = 0; jj < 4; ++jj) for (int nt = 0; nt < NT; ++nt) acc[jj][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row0 = r_begin + 4 * wave + q;
  const float* b_ptr = B + (size_t)row0 * 64 + p;
  constexpr size_t b_step = (size_t)4 * NW * 64;
  const float* a_ptr = A + (size_t)row0 * lda + (size_t)tile * 64 + 4 * p;
  const size_t a_step = (size_t)4 * NW * lda;
  for (int i = 0; i < nsteps; i += UNROLL) {
    f32x4 av[UNROLL]; float bv[UNROLL][NT];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const bool in = i + u < nsteps;
      const f32x4* ap = reinterpret_cast<const f32x4*>(a_ptr + (size_t)(i + u) * a_step);
      av[u] = in ? (NTL ? __builtin_nontemporal_load(ap) : *ap) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[u][nt] = in ? b_ptr[(size_t)(i + u) * b_step + 16 * nt] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], bv[u][nt], acc[jj][nt], 0, 0, 0);
  }
#pragma unroll
  for (int s = NW / 2; s >= 1; s >>= 1) {
    if (wave >= s && wave < 2 * s) {
      float* slot = red + (size_t)(wave - s) * 64 * KP;
      for (int jj = 0; jj < 4; ++jj) for (int nt = 0; nt < NT; ++nt) for (int t = 0; t < 4; ++t) slot[((jj * NT + nt) * 4 + t) * 64 + lane] = acc[jj][nt][t];
    }
    __syncthreads();
    if (wave < s) {
      const float* slot = red + (size_t)wave * 64 * KP;
      for (int jj = 0; jj < 4; ++jj) for (int nt = 0; nt < NT; ++nt) for (int t = 0; t < 4; ++t) acc[jj][nt][t] += slot[((jj * NT + nt) * 4 + t) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0)
    for (int jj = 0; jj < 4; ++jj) for (int nt = 0; nt < NT; ++nt) for (int t = 0; t < 4; ++t) red[(16 * q + 4 * t + jj) * KP + 16 * nt + p] = acc[jj][nt][t];
  __syncthreads();
  float* out = P + ((size_t)split * cols_pad + (size_t)tile * 64) * KP;
  for (int e = threadIdx.x * 4; e < 64 * KP; e += 256 * NW)
    *reinterpret_cast<f32x4*>(out + e) = *reinterpret_cast<const f32x4*>(&red[e]);
}

template <int NT, int NW, int UNROLL, bool NTL>
int bench(const char* tag, const float* A, int lda, int rows_pad, int cols_pad, const float* B, float* P, int rps, hipStream_t st) {
  const int ntiles = cols_pad / 64, ns = (rows_pad + rps - 1) / rps;
  const size_t smem = sizeof(float) * (NW / 2 > 0 ? NW / 2 : 1) * 64 * 16 * NT;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_k<NT, NW, UNROLL, NTL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto run = [&](int reps) { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((pass_k<NT, NW, UNROLL, NTL>), dim3(ntiles * ns), dim3(64 * NW), smem, st, A, lda, ntiles, B, P, cols_pad, rows_pad, rps); };
  run(3); CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  run(20); CK(hipStreamSynchronize(st));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
  printf("%s NT %d NW %2d UNROLL %2d nt %d rps %5d (%5d WGs, %3d steps/wave): %8.1f us -> %5.2f TB/s\n", tag, NT, NW, UNROLL, (int)NTL,
         rps, ntiles * ns, rps / (4 * NW), us, (double)rows_pad * cols_pad * 4 / us / 1e6);
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *A, *B, *P;
  const size_t abytes = (size_t)50048 * 8000 * 4;
  CK(hipMalloc(&A, abytes)); CK(hipMemset(A, 0, abytes));
  CK(hipMalloc(&B, (size_t)50048 * 64 * 4)); CK(hipMemset(B, 0, (size_t)50048 * 64 * 4));
  CK(hipMalloc(&P, (size_t)32 * 50048 * 64 * 4));
  // c4: Xt.F  A = [20032][4032],  X.G  A = [4032][20032]   (k = 32 -> NT 2)
  bench<2, 8, 4, false>("c4 xtf", A, 4032, 20032, 4032, B, P, 1280, st);
  bench<2, 8, 8, false>("c4 xtf", A, 4032, 20032, 4032, B, P, 1280, st);
  bench<2, 8, 8, true>("c4 xtf", A, 4032, 20032, 4032, B, P, 1280, st);
  bench<2, 4, 8, false>("c4 xtf", A, 4032, 20032, 4032, B, P, 1280, st);
  bench<2, 4, 8, true>("c4 xtf", A, 4032, 20032, 4032, B, P, 640, st);
  bench<2, 8, 8, true>("c4 xtf", A, 4032, 20032, 4032, B, P, 2560, st);
  bench<2, 8, 4, false>("c4 xg ", A, 20032, 4032, 20032, B, P, 512, st);
  bench<2, 8, 8, false>("c4 xg ", A, 20032, 4032, 20032, B, P, 512, st);
  bench<2, 8, 8, true>("c4 xg ", A, 20032, 4032, 20032, B, P, 512, st);
  bench<2, 4, 8, true>("c4 xg ", A, 20032, 4032, 20032, B, P, 512, st);
  bench<2, 8, 8, true>("c4 xg ", A, 20032, 4032, 20032, B, P, 1024, st);
  // c5: Xt.F  A = [50048][8000],  X.G  A = [8000][50048]   (k = 64 -> NT 4)
  bench<4, 8, 4, false>("c5 xtf", A, 8000, 50048, 8000, B, P, 3328, st);
  bench<4, 8, 8, false>("c5 xtf", A, 8000, 50048, 8000, B, P, 3328, st);
  bench<4, 8, 8, true>("c5 xtf", A, 8000, 50048, 8000, B, P, 3328, st);
  bench<4, 4, 8, true>("c5 xtf", A, 8000, 50048, 8000, B, P, 3328, st);
  bench<4, 4, 8, true>("c5 xtf", A, 8000, 50048, 8000, B, P, 1664, st);
  bench<4, 8, 4, false>("c5 xg ", A, 50048, 8000, 50048, B, P, 512, st);
  bench<4, 8, 8, true>("c5 xg ", A, 50048, 8000, 50048, B, P, 512, st);
  bench<4, 4, 8, true>("c5 xg ", A, 50048, 8000, 50048, B, P, 512, st);
  bench<4, 4, 8, true>("c5 xg ", A, 50048, 8000, 50048, B, P, 1024, st);
  return 0;
}
