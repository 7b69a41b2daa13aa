// Prints the accumulator layout of v_mfma_f64_16x16x4_f64: which D[i][j] register t of lane l holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x, p = l & 15, q = l >> 4;
  const double a = (q == 0) ? (double)(p + 1) : 0.0;          // A[i = p][k = q]
  const double b = (q == 0) ? (double)(p + 1) * 100.0 : 0.0;  // B[k = q][j = p]
  f64x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
  for (int t = 0; t < 4; ++t) out[l * 4 + t] = d[t];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8); probe<<<1, 64>>>(d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 32, 48, 63})
    for (int t = 0; t < 4; ++t) {
      const int v = (int)(h[l * 4 + t] / 100.0 + 0.5);   // (i + 1) * (j + 1)
      printf("lane %2d reg %d: value %6.0f  -> (i+1)*(j+1) = %d\n", l, t, h[l * 4 + t], v);
    }
  return 0;
}
