// First version of the wide form (kept to find out why it ran the c5 X.G pass in 343 us where the product body takes 400)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int NT, int PIECES>
__global__ __launch_bounds__(512, 2) void lab_wide(const float* __restrict__ A, size_t tile_stride, int ntiles,
                                                   const unsigned short* __restrict__ Bk, float* __restrict__ P, int cols_pad,
                                                   int rows_pad, int rps) {
  constexpr int KP = 16 * NT, NQ = NT * 3 * 64;
  extern __shared__ __attribute__((aligned(16))) u32x4_t ldsB[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const int ntg = (ntiles + 7) / 8, tg = blockIdx.x % ntg, split = blockIdx.x / ntg;
  const int tile = tg * 8 + wave;
  const bool valid = tile < ntiles;
  const int r_begin = split * rps, r_end = min(r_begin + rps, rows_pad);
  const int ntrip = (r_end - r_begin) / 32;
  f32x4 acc[4][NT];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[jj][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* a_ptr = A + (size_t)(valid ? tile : 0) * tile_stride + (size_t)(r_begin + q) * 64 + 4 * p;
  const u32x4_t* bg = reinterpret_cast<const u32x4_t*>(Bk) + (size_t)(r_begin >> 5) * NQ;
  constexpr int NB = (NQ + 511) / 512;
  auto load_a = [&](int t, f32x4 (&av)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a_ptr + (size_t)t * 2048 + (size_t)u * 256));
  };
  auto load_bg = [&](int t, u32x4_t (&g)[NB]) {
#pragma unroll
    for (int e = 0; e < NB; ++e) { const int i = threadIdx.x + 512 * e; g[e] = i < NQ ? bg[(size_t)t * NQ + i] : (u32x4_t){0, 0, 0, 0}; }
  };
  auto store_b = [&](int buf, const u32x4_t (&g)[NB]) {
#pragma unroll
    for (int e = 0; e < NB; ++e) { const int i = threadIdx.x + 512 * e; if (i < NQ) ldsB[buf * NQ + i] = g[e]; }
  };
  auto compute = [&](const f32x4 (&av)[8], int buf) {
    u32x4_t bq[NT][PIECES];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) bq[nt][pc] = ldsB[buf * NQ + (nt * 3 + pc) * 64 + lane];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      u32x4_t ah, am, al;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float x0 = av[2 * d][jj], x1 = av[2 * d + 1][jj];
        const unsigned int b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
        ah[d] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
        const unsigned int c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
        am[d] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
        if (PIECES == 3) {
          const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
          al[d] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
        }
      }
      const bf16x8_t fh = __builtin_bit_cast(bf16x8_t, ah), fm = __builtin_bit_cast(bf16x8_t, am);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8_t gh = __builtin_bit_cast(bf16x8_t, bq[nt][0]), gm = __builtin_bit_cast(bf16x8_t, bq[nt][1]);
        if (PIECES == 3) {
          const bf16x8_t fl = __builtin_bit_cast(bf16x8_t, al), gl = __builtin_bit_cast(bf16x8_t, bq[nt][PIECES - 1]);
          acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, gm, acc[jj][nt], 0, 0, 0);
          acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, gl, acc[jj][nt], 0, 0, 0);
          acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, gh, acc[jj][nt], 0, 0, 0);
        }
        acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, gm, acc[jj][nt], 0, 0, 0);
        acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm, gh, acc[jj][nt], 0, 0, 0);
        acc[jj][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, gh, acc[jj][nt], 0, 0, 0);
      }
    }
  };
  if (ntrip > 0) {
    f32x4 a0[8], a1[8];
    u32x4_t g[NB];
    load_a(0, a0);
    load_bg(0, g);
    store_b(0, g);
    __syncthreads();
    for (int t = 0; t < ntrip; t += 2) {
      if (t + 1 < ntrip) { load_a(t + 1, a1); load_bg(t + 1, g); }
      compute(a0, 0);
      if (t + 1 < ntrip) store_b(1, g);
      __syncthreads();
      if (t + 1 < ntrip) {
        if (t + 2 < ntrip) { load_a(t + 2, a0); load_bg(t + 2, g); }
        compute(a1, 1);
        if (t + 2 < ntrip) store_b(0, g);
        __syncthreads();
      }
    }
  }
  if (!valid) return;
  float* out = P + ((size_t)split * cols_pad + (size_t)tile * 64) * KP;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int t = 0; t < 4; ++t) out[(16 * q + 4 * t + jj) * KP + 16 * nt + p] = acc[jj][nt][t];
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
template <typename F>
double time_us(F&& launch, hipStream_t st) {
  for (int r = 0; r < 3; ++r) launch();
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 10; ++r) launch();
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
}
template <int NT, int PIECES>
int run_wide(const char* tag, const float* A, size_t ts, int ntiles, int rows_pad, const unsigned short* Bk, float* P, int rps, hipStream_t st) {
  const int cols_pad = ntiles * 64, ns = (rows_pad + rps - 1) / rps, ntg = (ntiles + 7) / 8;
  const size_t smem = (size_t)2 * NT * 3 * 64 * 16;
  auto fn = lab_wide<NT, PIECES>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const double us = time_us([&] { hipLaunchKernelGGL(fn, dim3(ntg * ns), dim3(512), smem, st, A, ts, ntiles, Bk, P, cols_pad, rows_pad, rps); }, st);
  printf("%s v1wide NT %d pieces %d rps %5d (%5d WGs): %8.1f us -> %5.2f TB/s\n", tag, NT, PIECES, rps, ntg * ns, us,
         ((double)rows_pad * cols_pad * 4 + 4.0 * (rows_pad + cols_pad) * 16 * NT) / us / 1e6);
  return 0;
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *A, *P; unsigned short* Bk;
  const size_t afl = (size_t)50048 * 8064 + (size_t)800 * 64;
  CK(hipMalloc(&A, afl * 4));
  CK(hipMalloc(&Bk, (size_t)50048 * 64 * 6));
  CK(hipMalloc(&P, (size_t)16 * 50048 * 64 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, A, afl);
  hipLaunchKernelGGL(fill_bk_kernel, dim3(1024), dim3(256), 0, st, Bk, (size_t)50048 * 64 * 3);
  CK(hipStreamSynchronize(st));
  run_wide<4, 3>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 1600, st);
  run_wide<4, 3>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 800, st);
  run_wide<4, 3>("c5 xtf", A, (size_t)50049 * 64, 125, 50048, Bk, P, 3136, st);
  run_wide<2, 3>("c4 xg ", A, (size_t)4033 * 64, 313, 4032, Bk, P, 672, st);
  run_wide<4, 3>("c5 xg ", A, (size_t)8001 * 64, 782, 8000, Bk, P, 1600, st);
  return 0;
}
