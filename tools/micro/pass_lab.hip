// Micro-benchmark: the library's streaming-pass body (resnmtf_kernels.hip.inc, pass_body) on c2's two
// operand copies, launched back to back from a graph the way a sweep alternates them:
//   A: Xt.F geometry on X32, X.G geometry on Xt32  (two 82-85 MB buffers alternate)
//   B: Xt.F geometry on X32 twice                  (one buffer, Infinity-Cache resident)
// for a list of (splits_xtf, splits_xg).  Reports us per launch pair and the implied read rate.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "../../resnmtf_amd/csrc/resnmtf_kernels.hip.inc"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NW>
__global__ __launch_bounds__(64 * NW, 2 * NW / 4) void lab_pass(const float* A, int lda, int ntiles, const float* B, float* P,
                                                               int cols_pad, int rows_pad, int rps) {
  extern __shared__ __attribute__((aligned(16))) float red_lab[];
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int r_begin = split * rps;
  pass_body<1, NW, 8>(A + (size_t)tile * 64, lda, B, 64, r_begin, min(r_begin + rps, rows_pad), red_lab,
                      P + ((size_t)split * cols_pad + (size_t)tile * 64) * 16);
}

int main() {
  const int n_pad = 10048, m_pad = 2048, ldx = 2112, ldxt = 10048;
  float *X, *Xt, *F, *G, *P;
  CK(hipMalloc(&X, (size_t)n_pad * ldx * 4)); CK(hipMalloc(&Xt, (size_t)m_pad * ldxt * 4));
  CK(hipMalloc(&F, (size_t)n_pad * 64 * 4)); CK(hipMalloc(&G, (size_t)m_pad * 64 * 4));
  CK(hipMalloc(&P, (size_t)64 * n_pad * 16 * 4));
  CK(hipMemset(X, 0, (size_t)n_pad * ldx * 4)); CK(hipMemset(Xt, 0, (size_t)m_pad * ldxt * 4));
  CK(hipMemset(F, 0, (size_t)n_pad * 64 * 4)); CK(hipMemset(G, 0, (size_t)m_pad * 64 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t smem = 4 * 64 * 16 * 4 + 1024;
  const double bytes = 4.0 * 10000 * 2000;
  auto rps_of = [](int rows, int ns) { int r = (rows + ns - 1) / ns; return (r + 63) / 64 * 64; };
  struct Geo { int s_xtf, s_xg; };
  for (Geo g : {Geo{14, 4}, Geo{15, 3}, Geo{15, 4}, Geo{16, 3}, Geo{20, 4}, Geo{24, 5}, Geo{30, 6}, Geo{40, 8}}) {
    for (int mode = 0; mode < 2; ++mode) {
      const int r1 = rps_of(n_pad, g.s_xtf), n1 = (n_pad + r1 - 1) / r1;
      const int r2 = rps_of(m_pad, g.s_xg), n2 = (m_pad + r2 - 1) / r2;
      auto pair = [&]() {
        hipLaunchKernelGGL(lab_pass<8>, dim3(32 * n1), dim3(512), smem, st, X, ldx, 32, F, P, m_pad, n_pad, r1);
        if (mode == 0) hipLaunchKernelGGL(lab_pass<8>, dim3(157 * n2), dim3(512), smem, st, Xt, ldxt, 157, G, P, n_pad, m_pad, r2);
        else hipLaunchKernelGGL(lab_pass<8>, dim3(32 * n1), dim3(512), smem, st, X, ldx, 32, F, P, m_pad, n_pad, r1);
      };
      hipGraph_t gr; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < 20; ++i) pair();
      CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
      printf("xtf %2d splits (%3d WGs) xg %d splits (%3d WGs) %s: %6.2f us per pair -> %5.2f TB/s\n", n1, 32 * n1, n2, 157 * n2,
             mode == 0 ? "X32 / Xt32 alternate" : "X32 twice           ", us, 2 * bytes / us / 1e6);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
    }
  }
  return 0;
}
