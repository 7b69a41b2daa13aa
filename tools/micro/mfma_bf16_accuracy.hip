// How does v_mfma_f32_16x16x32_bf16 round?  D = A B + C with bf16 operands (products exact in f32) against an fp64
// evaluation, for C = 0 and for C much larger than the 32-product sum (the situation of a long accumulation):
// signed error in units of ulp(D).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
__global__ void k(const u16* A /*[16][32]*/, const u16* B /*[32][16]*/, const float* C /*[16][16]*/, float* D, int reps) {
  const int lane = threadIdx.x, p = lane & 15, q = lane >> 4;
  u16 a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = A[p * 32 + 8 * q + j]; b[j] = B[(8 * q + j) * 16 + p]; }
  bf16x8_t av, bv;
  __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
  f32x4 c;
  for (int t = 0; t < 4; ++t) c[t] = C[(4 * q + t) * 16 + p];
  for (int r = 0; r < reps; ++r) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
  for (int t = 0; t < 4; ++t) D[(4 * q + t) * 16 + p] = c[t];
}
static float bf(u16 h) { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  std::vector<u16> A(512), B(512); std::vector<float> C(256), D(256);
  u16 *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
  srand(1);
  for (double cscale : {0.0, 1.0, 64.0, 4096.0}) {
    for (int reps : {1, 16}) {
      double sum_err = 0, sum_abs = 0, max_abs = 0; int cnt = 0;
      for (int trial = 0; trial < 200; ++trial) {
        for (auto& x : A) x = (u16)(0x3f80 + (rand() & 0x7f));          // [1, 2): all positive, 8 significant bits
        for (auto& x : B) x = (u16)(0x3f80 + (rand() & 0x7f));
        for (auto& x : C) x = (float)(cscale * 64.0 * (1.0 + (rand() % 1000) / 1000.0));
        hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, reps);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
          double s = 0; for (int l = 0; l < 32; ++l) s += (double)bf(A[i * 32 + l]) * (double)bf(B[l * 16 + j]);
          const double exact = (double)C[i * 16 + j] + reps * s;
          const double ulp = ldexp(1.0, ilogb(exact) - 23);
          const double e = ((double)D[i * 16 + j] - exact) / ulp;
          sum_err += e; sum_abs += fabs(e); max_abs = fmax(max_abs, fabs(e)); ++cnt;
        }
      }
      printf("C ~ %6.0f x one product sum, %2d chained MFMAs: mean signed error %+8.3f ulp, mean |error| %7.3f ulp, max %7.2f ulp\n", cscale, reps,
             sum_err / cnt, sum_abs / cnt, max_abs);
    }
  }
  return 0;
}
