// Micro-benchmark: does the width of the contiguous piece a workgroup reads per row matter?
// Xt.F geometry of c2 on X32 [10048][2112]: a workgroup of 8 waves covers TW columns (TW/64 sub-tiles
// of 64, one per wave group) x a row split; per step it reads (8 / (TW/64)) * 4 rows x TW*4 bytes.
// Also a plain linear read of the same buffer for reference.  All launches from a graph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// BMODE: 0 = no B loads (register constant), 1 = B32 pitch 64 floats (library layout), 2 = compact pitch 16
template <int TW, int UNROLL, int BMODE, int OCC, int COMPUTE = 1>
__global__ __launch_bounds__(512, OCC) void strip_pass(const float* __restrict__ A, int lda, int nwide, const float* __restrict__ B,
                                                     float* __restrict__ P, int rows_pad, int rps) {
  constexpr int BP = BMODE == 2 ? 16 : 64;
  constexpr int NSUB = TW / 64, NRG = 8 / NSUB;         // sub-tiles, row groups
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int sub = wave % NSUB, rg = wave / NSUB;
  const int wt = blockIdx.x % nwide, split = blockIdx.x / nwide;
  const int r_begin = split * rps, r_end = min(r_begin + rps, rows_pad);
  const int nsteps = (r_end - r_begin) / (4 * NRG);
  f32x4 acc[4];
  for (int jj = 0; jj < 4; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row0 = r_begin + 4 * rg + q;
  const float* a_ptr = A + (size_t)row0 * lda + (size_t)wt * TW + sub * 64 + 4 * p;
  const float* b_ptr = B + (size_t)row0 * BP + p;
  const size_t a_step = (size_t)4 * NRG * lda, b_step = (size_t)4 * NRG * BP;
  for (int i = 0; i < nsteps; i += UNROLL) {
    f32x4 av[UNROLL]; float bv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const bool in = i + u < nsteps;
      av[u] = in ? *reinterpret_cast<const f32x4*>(a_ptr + (size_t)(i + u) * a_step) : (f32x4){0.f, 0.f, 0.f, 0.f};
      bv[u] = BMODE == 0 ? (float)(i + u) : (in ? b_ptr[(size_t)(i + u) * b_step] : 0.f);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (COMPUTE == 0) { acc[u & 3] += av[u] * bv[u]; }
      else if (COMPUTE == 2) { acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0] + av[u][1] + av[u][2] + av[u][3], bv[u], acc[u & 3], 0, 0, 0); }
      else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], bv[u], acc[jj], 0, 0, 0);
      }
    }
  }
  // every wave stores its own accumulators (stand-in for the reduction + slab store: same bytes)
  float* out = P + ((size_t)blockIdx.x * 8 + wave) * 1024;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(out + (jj * 64 + lane) * 4) = acc[jj];
}

// software-pipelined form: the loads of trip t+1 are issued before the MFMAs of trip t
template <int UNROLL, int OCC>
__global__ __launch_bounds__(512, OCC) void strip_pipe(const float* __restrict__ A, int lda, int nwide, const float* __restrict__ B,
                                                      float* __restrict__ P, int rows_pad, int rps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int wt = blockIdx.x % nwide, split = blockIdx.x / nwide;
  const int r_begin = split * rps, r_end = min(r_begin + rps, rows_pad);
  const int nsteps = (r_end - r_begin) / 32;
  f32x4 acc[4];
  for (int jj = 0; jj < 4; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row0 = r_begin + 4 * wave + q;
  const float* a_ptr = A + (size_t)row0 * lda + (size_t)wt * 64 + 4 * p;
  const float* b_ptr = B + (size_t)row0 * 64 + p;
  const size_t a_step = (size_t)32 * lda, b_step = (size_t)32 * 64;
  f32x4 av[UNROLL], an[UNROLL]; float bv[UNROLL], bn[UNROLL];
  auto load = [&](int i, f32x4 (&a)[UNROLL], float (&b)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const bool in = i + u < nsteps;
      a[u] = in ? *reinterpret_cast<const f32x4*>(a_ptr + (size_t)(i + u) * a_step) : (f32x4){0.f, 0.f, 0.f, 0.f};
      b[u] = in ? b_ptr[(size_t)(i + u) * b_step] : 0.f;
    }
  };
  auto mma = [&](const f32x4 (&a)[UNROLL], const float (&b)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][jj], b[u], acc[jj], 0, 0, 0);
  };
  load(0, av, bv);
  for (int i = 0; i < nsteps; i += 2 * UNROLL) {
    load(i + UNROLL, an, bn);
    mma(av, bv);
    load(i + 2 * UNROLL, av, bv);
    mma(an, bn);
  }
  float* out = P + ((size_t)blockIdx.x * 8 + wave) * 1024;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(out + (jj * 64 + lane) * 4) = acc[jj];
}

// B operand compact ([rows][16]) and staged once per workgroup through LDS
template <int UNROLL, int OCC>
__global__ __launch_bounds__(512, OCC) void strip_ldsb(const float* __restrict__ A, int lda, int nwide, const float* __restrict__ B,
                                                      float* __restrict__ P, int rows_pad, int rps) {
  extern __shared__ __attribute__((aligned(16))) float Bs[];       // [rps][16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int wt = blockIdx.x % nwide, split = blockIdx.x / nwide;
  const int r_begin = split * rps, r_end = min(r_begin + rps, rows_pad);
  const int nsteps = (r_end - r_begin) / 32;
  f32x4 acc[4];
  for (int jj = 0; jj < 4; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row0 = r_begin + 4 * wave + q;
  const float* a_ptr = A + (size_t)row0 * lda + (size_t)wt * 64 + 4 * p;
  const size_t a_step = (size_t)32 * lda;
  // first trip of A in flight, then the B slab
  f32x4 av[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) av[u] = u < nsteps ? *reinterpret_cast<const f32x4*>(a_ptr + (size_t)u * a_step) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4* bsrc = reinterpret_cast<const f32x4*>(B + (size_t)r_begin * 16);
  const int n4 = (r_end - r_begin) * 4;
  for (int e = threadIdx.x; e < n4; e += 512) reinterpret_cast<f32x4*>(Bs)[e] = bsrc[e];
  __syncthreads();
  const float* bs = Bs + (4 * wave + q) * 16 + p;
  for (int i = 0; i < nsteps; i += UNROLL) {
    float bv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) bv[u] = (i + u < nsteps) ? bs[(size_t)(i + u) * 32 * 16] : 0.f;
    f32x4 an[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) an[u] = (i + UNROLL + u < nsteps) ? *reinterpret_cast<const f32x4*>(a_ptr + (size_t)(i + UNROLL + u) * a_step) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], bv[u], acc[jj], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) av[u] = an[u];
  }
  float* out = P + ((size_t)blockIdx.x * 8 + wave) * 1024;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(out + (jj * 64 + lane) * 4) = acc[jj];
}
// same without the prefetch of the next trip (load, wait, MFMA)
template <int UNROLL, int OCC>
__global__ __launch_bounds__(512, OCC) void strip_ldsb_np(const float* __restrict__ A, int lda, int nwide, const float* __restrict__ B,
                                                         float* __restrict__ P, int rows_pad, int rps) {
  extern __shared__ __attribute__((aligned(16))) float Bs[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int wt = blockIdx.x % nwide, split = blockIdx.x / nwide;
  const int r_begin = split * rps, r_end = min(r_begin + rps, rows_pad);
  const int nsteps = (r_end - r_begin) / 32;
  f32x4 acc[4];
  for (int jj = 0; jj < 4; ++jj) acc[jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row0 = r_begin + 4 * wave + q;
  const float* a_ptr = A + (size_t)row0 * lda + (size_t)wt * 64 + 4 * p;
  const size_t a_step = (size_t)32 * lda;
  f32x4 av[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) av[u] = u < nsteps ? *reinterpret_cast<const f32x4*>(a_ptr + (size_t)u * a_step) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4* bsrc = reinterpret_cast<const f32x4*>(B + (size_t)r_begin * 16);
  const int n4 = (r_end - r_begin) * 4;
  for (int e = threadIdx.x; e < n4; e += 512) reinterpret_cast<f32x4*>(Bs)[e] = bsrc[e];
  __syncthreads();
  const float* bs = Bs + (4 * wave + q) * 16 + p;
  for (int i = 0; i < nsteps; i += UNROLL) {
    if (i > 0) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) av[u] = (i + u < nsteps) ? *reinterpret_cast<const f32x4*>(a_ptr + (size_t)(i + u) * a_step) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float bv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) bv[u] = (i + u < nsteps) ? bs[(size_t)(i + u) * 32 * 16] : 0.f;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], bv[u], acc[jj], 0, 0, 0);
  }
  float* out = P + ((size_t)blockIdx.x * 8 + wave) * 1024;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4*>(out + (jj * 64 + lane) * 4) = acc[jj];
}

template <int UNROLL>
__global__ __launch_bounds__(512) void linear_read(const f32x4* __restrict__ p, size_t n4, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 512;
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
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
// contiguous chunk per workgroup (each WG reads one contiguous range, UNROLL x 8 KB in flight)
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

template <typename L>
double time_graph(hipStream_t st, L&& launch) {
  hipGraph_t gr; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 40; ++i) launch();
  hipStreamEndCapture(st, &gr); hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 400;
  hipGraphExecDestroy(ge); hipGraphDestroy(gr);
  return us;
}

int main() {
  const int n_pad = 10048, m_pad = 2048;
  float *X, *F, *P, *o;
  const size_t xbytes_alloc = (size_t)n_pad * 2112 * 4;
  CK(hipMalloc(&X, xbytes_alloc)); CK(hipMalloc(&F, (size_t)n_pad * 64 * 4)); CK(hipMalloc(&P, (size_t)4096 * 8 * 1024 * 4)); CK(hipMalloc(&o, 64));
  CK(hipMemset(X, 0, xbytes_alloc)); CK(hipMemset(F, 0, (size_t)n_pad * 64 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  const double bytes = 4.0 * n_pad * m_pad;
  const int lda = 2112;
  for (int wgs : {256, 480}) {
#define RUN(TWV, UN, BM, OCC, label)                                                                                    \
  {                                                                                                                     \
    const int nwide = m_pad / TWV, ns = wgs / nwide;                                                                    \
    int rps = (n_pad + ns - 1) / ns; rps = (rps + 31) / 32 * 32;                                                        \
    const int ns2 = (n_pad + rps - 1) / rps;                                                                            \
    const double us = time_graph(st, [&]() { hipLaunchKernelGGL((strip_pass<TWV, UN, BM, OCC>), dim3(nwide * ns2), dim3(512), 0, st, X, lda, nwide, F, P, n_pad, rps); }); \
    printf("%-34s %4d WGs (%3d splits of %4d rows): %6.2f us -> %5.2f TB/s\n", label, nwide * ns2, ns2, rps, us, bytes / us / 1e6); \
  }
#define RUNC(UN, BM, CP, label)                                                                                         \
  {                                                                                                                     \
    const int ns = wgs / 32;                                                                                            \
    int rps = (n_pad + ns - 1) / ns; rps = (rps + 31) / 32 * 32;                                                        \
    const int ns2 = (n_pad + rps - 1) / rps;                                                                            \
    const double us = time_graph(st, [&]() { hipLaunchKernelGGL((strip_pass<64, UN, BM, 4, CP>), dim3(32 * ns2), dim3(512), 0, st, X, lda, 32, F, P, n_pad, rps); }); \
    printf("%-40s %4d WGs (%3d splits of %4d rows): %6.2f us -> %5.2f TB/s\n", label, 32 * ns2, ns2, rps, us, bytes / us / 1e6); \
  }
    RUNC(8, 1, 1, "unroll 8, B global pitch 64")
    RUNC(8, 2, 1, "unroll 8, B global pitch 16")
    RUNC(8, 0, 1, "unroll 8, no B")
    RUNC(4, 2, 1, "unroll 4, B global pitch 16")
#define RUNL(KERN, UN, OCC, label)                                                                                      \
  {                                                                                                                     \
    const int ns = wgs / 32;                                                                                            \
    int rps = (n_pad + ns - 1) / ns; rps = (rps + 31) / 32 * 32;                                                        \
    const int ns2 = (n_pad + rps - 1) / rps;                                                                            \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&KERN<UN, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    const double us = time_graph(st, [&]() { hipLaunchKernelGGL((KERN<UN, OCC>), dim3(32 * ns2), dim3(512), (size_t)rps * 64, st, X, lda, 32, F, P, n_pad, rps); }); \
    printf("%-40s %4d WGs (%3d splits of %4d rows): %6.2f us -> %5.2f TB/s\n", label, 32 * ns2, ns2, rps, us, bytes / us / 1e6); \
  }
    RUNL(strip_ldsb, 8, 4, "B via LDS, prefetch next trip, un 8")
    RUNL(strip_ldsb, 4, 4, "B via LDS, prefetch next trip, un 4")
    RUNL(strip_ldsb_np, 8, 4, "B via LDS, no prefetch, un 8")
    RUNL(strip_ldsb_np, 4, 4, "B via LDS, no prefetch, un 4")
    RUNL(strip_ldsb_np, 2, 4, "B via LDS, no prefetch, un 2")
  }
  const size_t n4 = (size_t)n_pad * 2048 / 4;
  for (int grid : {256, 512, 1024, 2048}) {
    double us = time_graph(st, [&]() { hipLaunchKernelGGL(linear_read<8>, dim3(grid), dim3(512), 0, st, (const f32x4*)X, n4, o); });
    printf("linear grid-stride read, %4d WGs: %6.2f us -> %5.2f TB/s\n", grid, us, bytes / us / 1e6);
    us = time_graph(st, [&]() { hipLaunchKernelGGL(chunk_read<8>, dim3(grid), dim3(512), 0, st, (const f32x4*)X, n4, o); });
    printf("contiguous chunk per WG,   %4d WGs: %6.2f us -> %5.2f TB/s\n", grid, us, bytes / us / 1e6);
  }
  return 0;
}
