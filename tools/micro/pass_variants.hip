// Micro-benchmark: which part of the streaming pass costs time?  Same geometry as the Xt.F pass of
// c2 (A = [10048][2048] f32, 64-column tiles, NW waves x nsplit row splits), variants:
//   0 = A loads only (VALU add), 1 = A loads + MFMA with a register B, 2 = A + B loads + MFMA,
//   3 = variant 2 + tree reduction + partial store (the real kernel body)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW, int UNROLL, int VARIANT>
__global__ __launch_bounds__(64 * NW) void pass_variant(const float* __restrict__ A, int lda, int ntiles,
                                                       const float* __restrict__ B, float* __restrict__ P,
                                                       int cols_pad, int rows_pad, int rows_per_split) {
  constexpr int KP = 16;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int r_begin = split * rows_per_split;
  const int r_end = min(r_begin + rows_per_split, rows_pad);
  const int nsteps = (r_end - r_begin) / (4 * NW);
  f32x4 acc[4];
  for (int jj = 0; jj < 4; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row0 = r_begin + 4 * wave + q;
  const float* b_ptr = B + (size_t)row0 * 64 + p;
  constexpr size_t b_step = (size_t)4 * NW * 64;
  const float* a_ptr = A + (size_t)row0 * lda + (size_t)tile * 64 + 4 * p;
  const size_t a_step = (size_t)4 * NW * lda;
  int i = 0;
  for (; i + UNROLL <= nsteps; i += UNROLL) {
    f32x4 av[UNROLL];
    float bv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      av[u] = *reinterpret_cast<const f32x4*>(a_ptr + (size_t)(i + u) * a_step);
      bv[u] = (VARIANT >= 2) ? b_ptr[(size_t)(i + u) * b_step] : 1.0f + p;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (VARIANT == 0) { acc[0] += av[u]; }
      else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], bv[u], acc[jj], 0, 0, 0);
      }
    }
  }
  if (VARIANT < 3) {
    float s = 0.f;
    for (int jj = 0; jj < 4; ++jj) s += acc[jj][0] + acc[jj][1] + acc[jj][2] + acc[jj][3];
    if (s == 12345.678f) P[0] = s;
    return;
  }
#pragma unroll
  for (int s = NW / 2; s >= 1; s >>= 1) {
    if (wave >= s && wave < 2 * s) {
      float* slot = red + (size_t)(wave - s) * 64 * KP;
      for (int jj = 0; jj < 4; ++jj) for (int t = 0; t < 4; ++t) slot[(jj * 4 + t) * 64 + lane] = acc[jj][t];
    }
    __syncthreads();
    if (wave < s) {
      const float* slot = red + (size_t)wave * 64 * KP;
      for (int jj = 0; jj < 4; ++jj) for (int t = 0; t < 4; ++t) acc[jj][t] += slot[(jj * 4 + t) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0)
    for (int jj = 0; jj < 4; ++jj) for (int t = 0; t < 4; ++t) red[(16 * q + 4 * t + jj) * KP + p] = acc[jj][t];
  __syncthreads();
  float* out = P + ((size_t)split * cols_pad + (size_t)tile * 64) * KP;
  for (int e = threadIdx.x * 4; e < 64 * KP; e += 256 * NW)
    *reinterpret_cast<f32x4*>(out + e) = *reinterpret_cast<const f32x4*>(&red[e]);
}

template <int NW, int VARIANT>
int bench(const float* A, const float* A2, int lda, int ntiles, const float* B, float* P, int cols_pad, int rows_pad,
          int nsplit, hipStream_t st) {
  const int rps = ((rows_pad / nsplit + 63) / 64) * 64;
  const int ns = (rows_pad + rps - 1) / rps;
  const size_t smem = sizeof(float) * (NW / 2 > 0 ? NW / 2 : 1) * 64 * 16;
  auto run = [&](int reps) {
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL((pass_variant<NW, 8, VARIANT>), dim3(ntiles * ns), dim3(64 * NW), smem, st, A, lda, ntiles, B, P,
                         cols_pad, rows_pad, rps);
      hipLaunchKernelGGL((pass_variant<NW, 8, VARIANT>), dim3(ntiles * ns), dim3(64 * NW), smem, st, A2, lda, ntiles, B, P,
                         cols_pad, rows_pad, rps);
    }
  };
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); run(10); CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 400;
  printf("NW %2d nsplit %2d (%4d WGs) variant %d: %6.2f us per launch (incl. 1.6 us boundary) -> %5.2f TB/s\n", NW, ns,
         ntiles * ns, VARIANT, us, (double)rows_pad * lda * 4 / (us - 1.6) / 1e6);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

__global__ void touch_kernel(float* p, int n) {   // stands in for the update kernel: rewrites the B operand
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001f + 1e-9f;
}

template <int NW, int VARIANT>
int bench2(const float* A, int lda, int ntiles, float* B, float* P, int cols_pad, int rows_pad, int nsplit, bool touch,
           hipStream_t st) {
  const int rps = ((rows_pad / nsplit + 63) / 64) * 64;
  const int ns = (rows_pad + rps - 1) / rps;
  const size_t smem = sizeof(float) * (NW / 2 > 0 ? NW / 2 : 1) * 64 * 16;
  auto run = [&](int reps) {
    for (int r = 0; r < reps; ++r) {
      if (touch) hipLaunchKernelGGL(touch_kernel, dim3((rows_pad * 64 + 255) / 256), dim3(256), 0, st, B, rows_pad * 64);
      hipLaunchKernelGGL((pass_variant<NW, 8, VARIANT>), dim3(ntiles * ns), dim3(64 * NW), smem, st, A, lda, ntiles, B, P,
                         cols_pad, rows_pad, rps);
    }
  };
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); run(10); CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
  printf("lda %5d tiles %3d rows %5d NW %2d nsplit %2d (%4d WGs, %3d steps/wave) touch %d: %6.2f us per iteration\n", lda, ntiles,
         rows_pad, NW, ns, ntiles * ns, rps / (4 * NW), (int)touch, us);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *A, *B, *P;
  CK(hipMalloc(&A, (size_t)10048 * 2112 * 4 + (size_t)2048 * 10112 * 4)); CK(hipMemset(A, 0, (size_t)10048 * 2112 * 4 + (size_t)2048 * 10112 * 4));
  CK(hipMalloc(&B, (size_t)10048 * 64 * 4)); CK(hipMemset(B, 0, (size_t)10048 * 64 * 4));
  CK(hipMalloc(&P, (size_t)64 * 10112 * 16 * 4));
  float* Axt = A;                               // Xt.F geometry: [10048][2048]
  float* Axg = A + (size_t)10048 * 2112;        // X.G geometry:  [2048][10048]
  for (int touch = 0; touch < 2; ++touch) {
    bench2<8, 3>(Axt, 2048, 32, B, P, 2048, 10048, 16, touch, st);
    bench2<8, 3>(Axt, 2112, 32, B, P, 2048, 10048, 16, touch, st);
    bench2<8, 3>(Axg, 10048, 157, B, P, 10048, 2048, 4, touch, st);
    bench2<4, 3>(Axg, 10048, 157, B, P, 10048, 2048, 8, touch, st);
    bench2<4, 3>(Axg, 10048, 157, B, P, 10048, 2048, 4, touch, st);
    bench2<2, 3>(Axg, 10048, 157, B, P, 10048, 2048, 8, touch, st);
    bench2<16, 3>(Axg, 10048, 157, B, P, 10048, 2048, 2, touch, st);
    bench2<8, 3>(Axg, 10048, 157, B, P, 10048, 2048, 13, touch, st);
    bench2<8, 0>(Axg, 10048, 157, B, P, 10048, 2048, 4, touch, st);
    bench2<8, 2>(Axg, 10048, 157, B, P, 10048, 2048, 4, touch, st);
  }
  return 0;
}
