// resnmtf_hip.hip -- host side of libresnmtf_hip.so: handle, device memory, launch schedule,
// hipGraph capture and the C-ABI declared in include/resnmtf_hip.h.
//
// Schedule of one sweep (R/update_steps.r:272-319) for owned views v = 0..V-1, in order, ONE stream:
//   F_v          factor_update<F>   update_f, reads U_v = X_v G_v and the F coefficients
//   pass Xt.F    main workgroups: T_v = X_v^T F_v'   | k x k job kk_f
//   G_v          factor_update<G>   update_g, reads T_v and the G coefficients
//   pass X.G'    main workgroups: U_v = X_v G_v'     | k x k job kk_s
// Every k x k chain (S rule, lambda/mu, error, coefficient matrices) depends only on the factor that
// was just updated; it runs once inside the pass launch that follows that update, while the main
// workgroups stream X: as workgroup 0 fed by the update kernel's fp64 partials (mode A, k <= 16) or
// in the last-arriving MFMA aux workgroup (mode B) -- see resnmtf_kernels.hip.inc.  Cross-view
// coupling (phi/psi: running F/G; xi: running S) is ordered by the stream.  A run starts with one
// X.G launch per view whose kk_s runs in mode 0 (F coefficients from the current S and G).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "resnmtf_hip.h"
#include "resnmtf_kernels.hip.inc"
#include "resnmtf_split_tu.h"
#ifdef RESNMTF_SPLIT_TU      // product build: the k <= 16 pass lives in resnmtf_pass_k16.hip (its own scheduling strategy)
#define RESNMTF_EXTERN(NW, UNR, XG, MA) extern template __global__ void pass_kernel<1, NW, UNR, XG, MA, 0>(PassArgs, KKFArgs, KKSArgs);
RESNMTF_PASS_K16_LIST(RESNMTF_EXTERN)
#undef RESNMTF_EXTERN
#endif

namespace {

std::string g_create_error;

inline int round_up(int x, int a) { return (x + a - 1) / a * a; }
inline int ceil_div(int x, int a) { return (x + a - 1) / a; }

struct SharedMap {
  bool set = false;        // false or count < 0  ->  NA
  int count = -1;
  int* dev = nullptr;      // [len_v] row of w or -1
  bool identity = false;   // every row of v maps to the same row of w (auto-named views): no lookup needed
};

struct ViewState {
  int n = 0, m = 0, k = 0, KP = 16, NT = 1;
  bool owned = true, has_x = false, has_factors = false;
  int empty_rows = 0, empty_cols = 0;          // all-zero rows / columns of the latest device-drawn data (shuffle, sub-sample)
  std::vector<unsigned char> empty_mask;     // [n + m], 1 = the row / column sums to zero
  int n_pad = 0, m_pad = 0;
  size_t ldx = 0, ldxt = 0;      // TILE strides of X32 / Xt32 (floats): tile t (64 columns) is a contiguous [rows_pad][64] block
  size_t x32_floats = 0, xt32_floats = 0;
  // fp16 passes (resnmtf_options.x_half, k <= 16): K-packed fp16 images of X / X^T, their tile strides (halves), scale
  bool half = false, u16 = false;   // u16: uniform 16-bit integers instead of fp16 (x_half = 2, 3)
  bool half_capable = false;        // the 2-byte images are allocated; `half` says whether the passes use them (x_half = 3: guard)
  double x_relerr = 0.0;            // || X~ - X ||_F / || X ||_F of the 2-byte image (set at upload)
  _Float16 *X16 = nullptr, *Xt16 = nullptr;
  size_t ld16x = 0, ld16xt = 0, x16_halves = 0, xt16_halves = 0;
  float xscale = 1.f;
  float *X32 = nullptr, *Xt32 = nullptr;
  double* xnorm2 = nullptr;
  double *F = nullptr, *G = nullptr, *S = nullptr, *lambda = nullptr, *mu = nullptr;
  float *F32 = nullptr, *G32 = nullptr, *T32 = nullptr;
  unsigned short *Fk = nullptr, *Gk = nullptr;   // k > 16: K-packed bf16 pieces of F / G (B operands of pass_body_k32)
  int nsplit_xg = 1, rps_xg = 64, nw_xg = 4, nsaux_xg = 1, rpsaux_xg = 64;
  int nsplit_xtf = 1, rps_xtf = 64, nw_xtf = 4, nsaux_xtf = 1, rpsaux_xtf = 64;
  int tw_xg = 8, tw_xtf = 8;         // k > 16 (wide form): 64-column tiles per workgroup
  float *Pxg = nullptr, *Pxtf = nullptr, *Paux_xg = nullptr, *Paux_xtf = nullptr;
  int *cnt_xg = nullptr, *cnt_xtf = nullptr;
  int* fuse_cnt = nullptr;           // [2] arrivals of the update blocks fused into the Xt.F ([0]) / X.G ([1]) launch (pass_fused_kernel)
  int rpbF = 16, nblkF = 1, rpbG = 16, nblkG = 1;
  void* fblk = nullptr;              // replicate_f: [Usum | Ma_F | Md_F | lambda] contiguous (the F-update's inputs), a
  size_t fblk_bytes = 0;             // slice of the handle's arena; Usum = the X.G split slabs folded into one f32 slab
  float* Usum = nullptr;
  bool f_replica = false;            // non-owned view whose F update runs here too (replicate_f)
  void* gblk = nullptr; size_t gblk_bytes = 0;   // replicate_gs: [Tsum | Ma_G | Md_G | mu] (the G update's inputs), arena slice
  float* Tsum = nullptr;             //   Tsum = the Xt.F split slabs folded into one f32 slab
  bool g_replica = false;            // non-owned view whose G update runs here too (replicate_gs)
  double* sblk = nullptr;            // replicate_gs: the S update's inputs (sblock_layout), arena slice
  bool pp_xg = false, pp_xtf = false; // k <= 16, streamed geometry: ping-pong prefetch form of the pass (UNROLL 4)
  int kk_mode = 0;                   // 0 = A: Gram partials from the update kernels, k x k job = workgroup 0
                                     // 1 = B: Gram/cross/colsum on MFMA aux tiles, k x k job = last-arriving aux workgroup
  double *partF = nullptr, *partG = nullptr;
  double *FtF = nullptr, *FtFS = nullptr, *cF = nullptr;
  double *Ma_F = nullptr, *Md_F = nullptr, *Ma_G = nullptr, *Md_G = nullptr;
  std::vector<SharedMap> row_map, col_map;   // indexed by the other view
  UpdateArgs argF{}, argG{};
  KKFArgs argKF{};
  KKSArgs argKS{};
  PassArgs passXG{}, passXtF{};
};

}  // namespace

struct resnmtf_handle {
  int V = 0;
  std::vector<ViewState> views;
  resnmtf_options opt{};
  hipStream_t stream = nullptr;       // every kernel of the loop (+ the host's exchanges)
  bool own_stream = false;
  std::vector<double> phi, xi, psi;   // V x V column-major
  SweepCtl* ctl = nullptr;
  double* err = nullptr;              // [err_cap][V]
  int err_cap = 0;
  int last_owned = -1;
  bool prepared = false;
  bool all_owned = true;
  // graphs
  // ladder of captured sweep graphs: check_every sweeps plus every smaller power of two, so that any run length
  // is a handful of graph launches (R/main.r:83-108 is the loop being replayed)
  std::vector<std::pair<int, hipGraphExec_t>> ladder;      // (sweeps, executable), descending
  std::vector<std::pair<int, hipGraphExec_t>> exact;       // short runs (< 3 batches) repeated with the same length: one graph each
  double graph_tol = -2.0;
  bool resume_ok = false;             // the device state is exactly what the run prologue would produce: skip it
  bool ctl_clean = false;             // ... and the loop control needs no reset either (fixed sweeps after fixed sweeps):
  int sweep_base = 0;                 //     the device's sweep counter then simply runs on: value at the START of the
  int next_base = 0;                  //     latest run / at its end
  SweepCtl* ctl_host = nullptr;       // pinned, device-mapped mirrors written by the k x k job of a sweep's last view /
  double* err_host = nullptr;         //   every view (fixed-iteration runs end with one stream synchronisation, no copy)
  SweepCtl* ctl_host_dev = nullptr;
  double* err_host_dev = nullptr;
  int n_cu = 256;                     // compute units of the device (multiProcessorCount)
  ChainArgs<8> chain{};               // RESNMTF_PHASE_F_ALL: the F updates of every view in one launch (when eligible)
  int chain_views = 0;                // 0 = not eligible: one launch per view
  ChainArgs<8> gchain{};              // RESNMTF_PHASE_G_ALL at k <= 16 (replicated G chain): the G updates of every view in one launch
  int gchain_views = 0, gchain_blocks = 0;
  WideChainArgs<8> wchain[2]{};       // k = 32 / 64: the F ([0]) and G ([1]) updates of every view in one launch (wide_chain_kernel)
  bool wchain_ok[2] = {false, false};
  int wchain_grid[2] = {0, 0};
  int chain_blocks = 0;
  void* fblk_arena = nullptr;         // replicate_f: the F exchange blocks of all views, in view order
  size_t fblk_arena_bytes = 0;
  void* gblk_arena = nullptr; size_t gblk_arena_bytes = 0;     // replicate_gs: the G / S exchange blocks of all views
  double* sblk_arena = nullptr; size_t sblk_stride = 0;        //   (S blocks: sblk_stride doubles each)
  bool sblk_embedded = false;         // the S block of a view sits at the end of its F block (equal-shaped views): they travel together
  double* sblk_base = nullptr; size_t sblk_step = 0;           // S block of view v = sblk_base + v * sblk_step (doubles)
  int* view_sweep = nullptr;          // replicate_gs: [V] sweeps closed per view (s_chain_kernel) + [1] its arrival counter
  double phase_tol = -1.0;            // phase API: >= 0 = convergence mode (resnmtf_set_stop_tolerance)
  int* fuse_err = nullptr;            // pinned, device-mapped: set by a fused pass launch whose wait for its update blocks ran out
  int* fuse_err_dev = nullptr;
  // slice_chains: rank r walks the F (G) chain of every view on the row (column) slice r; exchange buffers of the four
  // all-to-alls of a sweep (V chunks each) and the two chain launches
  bool sliced = false;
  int sl_rows = 0, sl_cols = 0;       // rows / columns per slice (multiples of 32)
  char *u_send = nullptr, *u_recv = nullptr, *t_send = nullptr, *t_recv = nullptr;
  size_t u_chunk = 0, t_chunk = 0;    // bytes per chunk: [per_slice][KP] f32 (+ the fp64 tail Ma_G | Md_G for T)
  float *f_send = nullptr, *f_recv = nullptr, *g_send = nullptr, *g_recv = nullptr;      // [V][per_slice][KP] f32 each
  // slice_p2p: the exchange as peer stores + stream-ordered flags (no collective): every rank's receive buffers and flag words,
  // mapped through hipIpc (own rank: the local pointers); flags[e] counts the arrivals of exchange e (0 U + S blocks, 1 new F
  // rows, 2 T slices, 3 new G rows): V per sweep
  struct Peer { char *u_recv = nullptr, *t_recv = nullptr; float *f_recv = nullptr, *g_recv = nullptr; double* sblk = nullptr;
                char *fblk_arena = nullptr, *gblk_arena = nullptr;      // block_p2p: the replicated layouts' exchange arenas
                unsigned int* flags = nullptr; bool imported = false; void* opened[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; };
  std::vector<Peer> peers;
  unsigned int* p2p_flags = nullptr;
  bool p2p_ready = false, p2p_prepared = false;
  unsigned int probe_epoch = 0;       // resnmtf_p2p_selftest calls so far (its arrival counter is cumulative)
  bool block_p2p = false;             // slice_p2p without slice_chains: the exchange blocks of the replicated layouts by peer stores
  double* slice_nd = nullptr;         // sliced chains in two launches: num | den of every view on this rank's slice (slice_products_kernel)
  WideChainArgs<8> schain[2]{};       // SLICE_F ([0]) / SLICE_G ([1])
  int schain_grid[2] = {0, 0};
  double ktime_ms[RESNMTF_TIMED_KINDS] = {0, 0, 0, 0, 0, 0};     // time_kernels: per kind (resnmtf_kernel_timings)
  long long klaunch[RESNMTF_TIMED_KINDS] = {0, 0, 0, 0, 0, 0};
  // pass timing (eager mode)
  std::vector<hipEvent_t> ev;         // pairs
  std::vector<int> ev_kind;           // 0 = xg, 1 = xtf per pair
  size_t ev_used = 0;
  resnmtf_pass_timing timing{};
  std::string last_error;

  int fail(int code, const std::string& msg) {
    last_error = msg;
    return code;
  }
  int fail_hip(const char* what, hipError_t e) {
    last_error = std::string(what) + ": " + hipGetErrorString(e);
    return RESNMTF_ERR_HIP;
  }
};

#define HIP_TRY(h, expr)                                      \
  do {                                                        \
    hipError_t e_ = (expr);                                   \
    if (e_ != hipSuccess) return (h)->fail_hip(#expr, e_);    \
  } while (0)

namespace {

template <typename T>
hipError_t dev_alloc_zero(T** p, size_t count) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(T));
}

// time_kernels: a pair of events for one launch of the given kind (RESNMTF_TIMED_*), attached to the dispatch itself
bool take_events(resnmtf_handle* h, int kind, hipEvent_t* e0, hipEvent_t* e1) {
  if (!h->opt.time_kernels || h->ev_used + 2 > h->ev.size()) return false;
  *e0 = h->ev[h->ev_used]; *e1 = h->ev[h->ev_used + 1];
  h->ev_kind[h->ev_used / 2] = kind;
  h->ev_used += 2;
  return true;
}
#define LAUNCH_TIMED(h, kind, kern, grid, block, smem, ...)                                                        \
  do {                                                                                                             \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                                       \
    if (take_events(h, kind, &e0_, &e1_)) hipExtLaunchKernelGGL(kern, grid, block, smem, (h)->stream, e0_, e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, smem, (h)->stream, __VA_ARGS__);                                     \
  } while (0)

void free_view(ViewState& v) {
  if (v.fblk) { v.fblk = nullptr; v.Usum = nullptr; v.Ma_F = nullptr; v.Md_F = nullptr; v.lambda = nullptr; }   // arena slices
  if (v.gblk) { v.gblk = nullptr; v.Tsum = nullptr; v.Ma_G = nullptr; v.Md_G = nullptr; v.mu = nullptr; }
  v.sblk = nullptr;
  void* ptrs[] = {v.fuse_cnt, v.Fk, v.Gk, v.X16, v.Xt16, v.X32, v.Xt32, v.xnorm2, v.F, v.G, v.S, v.lambda, v.mu, v.F32, v.G32, v.T32, v.Pxg, v.Pxtf,
                  v.Paux_xg, v.Paux_xtf, v.cnt_xg, v.cnt_xtf, v.partF, v.partG, v.FtF, v.FtFS, v.cF, v.Ma_F, v.Md_F, v.Ma_G, v.Md_G};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& mp : v.row_map)
    if (mp.dev) (void)hipFree(mp.dev);
  for (auto& mp : v.col_map)
    if (mp.dev) (void)hipFree(mp.dev);
}

int check_view(resnmtf_handle* h, int v) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (v < 0 || v >= h->V) return h->fail(RESNMTF_ERR_INVALID, "view index out of range");
  return RESNMTF_OK;
}

// column-major host -> row-major host
void to_row_major(const double* src, int rows, int cols, std::vector<double>& dst) {
  dst.resize((size_t)rows * cols);
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) dst[(size_t)i * cols + j] = src[(size_t)j * rows + i];
}
void to_col_major(const std::vector<double>& src, int rows, int cols, double* dst) {
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) dst[(size_t)j * rows + i] = src[(size_t)i * cols + j];
}

size_t update_smem_bytes(int KP) {
  const int UT = update_threads(KP), RG = UT / KP;      // (padded pitches of the fp64-MFMA form included)
  const int P16 = (KP % 32 == 0) ? KP + 16 : KP + 32;
  return sizeof(double) * ((size_t)2 * KP * P16 + 2 * (size_t)RG * (KP + 2) + (size_t)RG * KP + 2 * UT + 2 * (size_t)RG * P16);
}
size_t kk_smem_bytes(int KP, int NW) { return sizeof(double) * ((size_t)4 * KP * KP + 3 * 64 * (size_t)NW); }
size_t pass_smem_bytes(int KP, int NW) {
  const size_t tile_form = std::max(sizeof(float) * (size_t)std::max(NW / 2, 1) * 64 * KP, kk_smem_bytes(KP, NW));
  return KP > 16 ? std::max(tile_form, wide_smem_bytes(KP / 16)) : tile_form;
}
constexpr int kMaxLds = 160 * 1024;

template <int NT, int NW, int UNROLL>
hipError_t set_pass_attr() {
  constexpr int S3 = NT >= 2 ? 3 : 0;      // (NT = 1 has no bf16 form: the lists coincide)
  const void* fns[8] = {reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, false, false>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, true, false>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, false, true>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, true, true>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, false, false, S3>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, true, false, S3>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, false, true, S3>),
                         reinterpret_cast<const void*>(&pass_kernel<NT, NW, UNROLL, true, true, S3>)};
  for (const void* fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <int KP>
hipError_t set_smem_attrs() {
  hipError_t e;
#define SET_ATTR(fn, bytes)                                                                                    \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)(bytes))) != hipSuccess)                                                   \
  return e
  SET_ATTR((factor_update_kernel<KP, false, 0, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, 4, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, 8, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, RESNMTF_MAX_COUPLE, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, 0, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, 4, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, 8, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, RESNMTF_MAX_COUPLE, false>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, 0, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, 4, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, 8, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, false, RESNMTF_MAX_COUPLE, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, 0, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, 4, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, 8, true>), update_smem_bytes(KP));
  SET_ATTR((factor_update_kernel<KP, true, RESNMTF_MAX_COUPLE, true>), update_smem_bytes(KP));
#undef SET_ATTR
  return hipSuccess;
}

hipError_t set_all_attrs() {
  hipError_t e;
#define TRY_ATTR(x) if ((e = (x)) != hipSuccess) return e
  TRY_ATTR(set_smem_attrs<16>()); TRY_ATTR(set_smem_attrs<32>());
  TRY_ATTR(set_smem_attrs<48>()); TRY_ATTR(set_smem_attrs<64>());
  TRY_ATTR((set_pass_attr<1, 4, 8>())); TRY_ATTR((set_pass_attr<1, 8, 8>())); TRY_ATTR((set_pass_attr<1, 8, 4>())); TRY_ATTR((set_pass_attr<1, 16, 8>()));
  TRY_ATTR((set_pass_attr<2, 8, 4>())); TRY_ATTR((set_pass_attr<3, 8, 4>())); TRY_ATTR((set_pass_attr<4, 8, 4>()));
  for (const void* fn : {reinterpret_cast<const void*>(&pass_fused_kernel<8, false, true>), reinterpret_cast<const void*>(&pass_fused_kernel<8, true, true>),
                         reinterpret_cast<const void*>(&pass_fused_kernel<4, false, true>), reinterpret_cast<const void*>(&pass_fused_kernel<4, true, true>),
                         reinterpret_cast<const void*>(&pass_fused_kernel<8, false, false>), reinterpret_cast<const void*>(&pass_fused_kernel<8, true, false>),
                         reinterpret_cast<const void*>(&pass_fused_kernel<4, false, false>), reinterpret_cast<const void*>(&pass_fused_kernel<4, true, false>)})
    TRY_ATTR(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
  // (s_chain_kernel also holds a static table of NVB x NVB weights: static + dynamic must stay within the CU's LDS)
#define S_CHAIN_ATTR(KPV, NVBV) TRY_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&s_chain_kernel<KPV, NVBV>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096))
  S_CHAIN_ATTR(16, 4); S_CHAIN_ATTR(16, 8); S_CHAIN_ATTR(16, RESNMTF_MAX_COUPLE + 1);
  S_CHAIN_ATTR(32, 4); S_CHAIN_ATTR(32, 8); S_CHAIN_ATTR(32, RESNMTF_MAX_COUPLE + 1);
  S_CHAIN_ATTR(48, 4); S_CHAIN_ATTR(48, 8); S_CHAIN_ATTR(48, RESNMTF_MAX_COUPLE + 1);
  S_CHAIN_ATTR(64, 4); S_CHAIN_ATTR(64, 8); S_CHAIN_ATTR(64, RESNMTF_MAX_COUPLE + 1);
#undef S_CHAIN_ATTR
#define WCHAIN_ATTR(KPV, G, NVBV) TRY_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_chain_kernel<KPV, G, NVBV>), \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_chain_smem_bytes(KPV)))
  WCHAIN_ATTR(32, false, 4); WCHAIN_ATTR(32, true, 4); WCHAIN_ATTR(32, false, 8); WCHAIN_ATTR(32, true, 8);
  WCHAIN_ATTR(48, false, 4); WCHAIN_ATTR(48, true, 4); WCHAIN_ATTR(48, false, 8); WCHAIN_ATTR(48, true, 8);
  WCHAIN_ATTR(64, false, 4); WCHAIN_ATTR(64, true, 4); WCHAIN_ATTR(64, false, 8); WCHAIN_ATTR(64, true, 8);
#undef WCHAIN_ATTR
#define SCHAIN_ATTR(KPV, G, NVBV) TRY_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_chain_kernel<KPV, G, NVBV, true>), \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_chain_smem_bytes(KPV)))
  SCHAIN_ATTR(16, false, 4); SCHAIN_ATTR(16, true, 4); SCHAIN_ATTR(16, false, 8); SCHAIN_ATTR(16, true, 8);
  SCHAIN_ATTR(32, false, 4); SCHAIN_ATTR(32, true, 4); SCHAIN_ATTR(32, false, 8); SCHAIN_ATTR(32, true, 8);
  SCHAIN_ATTR(48, false, 4); SCHAIN_ATTR(48, true, 4); SCHAIN_ATTR(48, false, 8); SCHAIN_ATTR(48, true, 8);
  SCHAIN_ATTR(64, false, 4); SCHAIN_ATTR(64, true, 4); SCHAIN_ATTR(64, false, 8); SCHAIN_ATTR(64, true, 8);
#undef SCHAIN_ATTR
#define SLCHAIN_ATTR(KPV, G, NVBV) TRY_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&slice_chain_kernel<KPV, G, NVBV>), \
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)slice_chain_smem_bytes(KPV, NVBV)))
  SLCHAIN_ATTR(16, false, 4); SLCHAIN_ATTR(16, true, 4); SLCHAIN_ATTR(16, false, 8); SLCHAIN_ATTR(16, true, 8);
  SLCHAIN_ATTR(32, false, 4); SLCHAIN_ATTR(32, true, 4); SLCHAIN_ATTR(32, false, 8); SLCHAIN_ATTR(32, true, 8);
  SLCHAIN_ATTR(48, false, 4); SLCHAIN_ATTR(48, true, 4); SLCHAIN_ATTR(48, false, 8); SLCHAIN_ATTR(48, true, 8);
  SLCHAIN_ATTR(64, false, 4); SLCHAIN_ATTR(64, true, 4); SLCHAIN_ATTR(64, false, 8); SLCHAIN_ATTR(64, true, 8);
#undef SLCHAIN_ATTR
#define CHAIN_ATTR(NVB, PFV) TRY_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&f_chain_kernel<NVB, PFV>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)f_chain_smem_bytes<NVB>()))
  CHAIN_ATTR(2, 1); CHAIN_ATTR(4, 1); CHAIN_ATTR(8, 1); CHAIN_ATTR(2, 4); CHAIN_ATTR(4, 4); CHAIN_ATTR(8, 4);
#undef CHAIN_ATTR
#define GCHAIN_ATTR(NVB) TRY_ATTR(hipFuncSetAttribute(reinterpret_cast<const void*>(&f_chain_kernel<NVB, 1, true>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)f_chain_smem_bytes<NVB>()))
  GCHAIN_ATTR(2); GCHAIN_ATTR(4); GCHAIN_ATTR(8);
#undef GCHAIN_ATTR
#undef TRY_ATTR
  return hipSuccess;
}

// waves per workgroup: 8; k <= 16 also has 4- and 16-wave instantiations (tuning sweeps)
int max_pass_waves(int NT) { return NT <= 1 ? 16 : 8; }
// workgroups of a pass launch one CU holds at once: the kernels' __launch_bounds__ (kernels.hip.inc,
// pass_min_blocks) keeps the k <= 32 instantiations within 128 VGPRs
int pass_blocks_per_cu(int NT, int nw) { return pass_min_blocks(NT, nw); }

// xg = false: Xt.F pass + kk_f;  xg = true: X.G pass + kk_s (mode 0 = run prologue, 1 = full S update)
// can the F (kind 0) / G (kind 1) update of view v ride in the pass launch that consumes it (pass_fused_kernel)?  Mode A at
// k <= 16 with the f32 images and 8-wave workgroups, the unrestricted update form (the coupled forms exceed the pass's
// register budget), and few enough row blocks that every updater is resident among the launch's first workgroups
bool can_fuse_update(const resnmtf_handle* h, const ViewState& v, int kind) {
  if (h->opt.fuse_updates == 0 || v.kk_mode != 0 || v.NT != 1 || v.half || !v.fuse_cnt) return false;
  const UpdateArgs& u = kind == 0 ? v.argF : v.argG;
  const PassArgs& p = kind == 0 ? v.passXtF : v.passXG;
  const int nblk = kind == 0 ? v.nblkF : v.nblkG, nw = kind == 0 ? v.nw_xtf : v.nw_xg;
  return nw == 8 && !u.restricted && nblk <= p.ntiles * p.nsplit && nblk <= h->n_cu;
}
void launch_pass(resnmtf_handle* h, const ViewState& v, bool xg, int mode, double tol, bool check_done, bool fuse_update = false) {
  PassArgs a = xg ? v.passXG : v.passXtF;
  a.check_done = check_done ? 1 : 0;
  KKFArgs kf = v.argKF;
  KKSArgs ks = v.argKS;
  ks.mode = mode; ks.tol = tol;
  // mode A: the k x k job is workgroup 0 and reads the update kernel's fp64 partials
  a.kk_block0 = (v.kk_mode == 0) ? 1 : 0;
  a.fuse_zero = (a.kk_block0 && v.fuse_cnt && !v.half) ? v.fuse_cnt + (xg ? 0 : 2) : nullptr;   // the sibling launch's arrival counter
  if (!a.kk_block0) { kf.part = nullptr; ks.part = nullptr; }
  const int nw = xg ? v.nw_xg : v.nw_xtf;
  // MFMA form of the main tiles (resnmtf_options.bf16_split): k <= 16 always the f32 MFMA; k > 16: three bf16 pieces per
  // operand on the K = 32 MFMA in wide workgroups (f32-grade, default; 1 is accepted as an alias), 2 = plain f32 MFMA
  const int split = v.NT < 2 ? 0 : (h->opt.bf16_split == 2 ? 0 : 3);
  const int main_blocks = split == 3 ? a.ntg * a.nsplit : a.ntiles * a.nsplit;
  const int lead_blocks = a.kk_block0 ? 1 : a.naux * a.nsplit_aux;       // the k x k job / the aux workgroups head the grid
  const dim3 grid(lead_blocks + main_blocks), block(64 * nw);
  a.xcd_n = 0;
  if (split == 3 && h->opt.xcd_order) {
    // XCD-aware order of the wide form's main workgroups (PassArgs::xcd_n).  A short last split stays at the end of the grid.
    const bool short_last = a.nsplit > 1 && a.rows_pad - (a.nsplit - 1) * a.rows_per_split < a.rows_per_split;
    const int n_map = short_last ? (a.nsplit - 1) * a.ntg : main_blocks;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 0; x < 8; ++x) a.xcd_first[x] = -1;
    for (int b = 0; b < n_map; ++b) {
      const int x = (lead_blocks + b) & 7;
      if (a.xcd_first[x] < 0) a.xcd_first[x] = lead_blocks + b;
      ++cnt[x];
    }
    int off = 0;
    for (int x = 0; x < 8; ++x) { a.xcd_off[x] = off; off += cnt[x]; if (a.xcd_first[x] < 0) a.xcd_first[x] = 0; }
    a.xcd_n = n_map;
  }
  const size_t smem = std::min<size_t>(pass_smem_bytes(v.KP, nw) + (size_t)h->opt.pass_lds_pad_kb * 1024, kMaxLds);
  // timed mode: the start/stop events are attached to the dispatch itself (hipExtLaunchKernelGGL), so
  // the elapsed time is the kernel's own begin->end, the same quantity rocprofv3 --kernel-trace reports
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool timed = take_events(h, xg ? RESNMTF_TIMED_XG : RESNMTF_TIMED_XTF, &ev0, &ev1);
  if (fuse_update) {   // the update that feeds this pass rides in its first workgroups (pass_fused_kernel)
    UpdateArgs u = xg ? v.argG : v.argF;
    u.check_done = 0; u.gram_only = 0;
    FuseArgs fz{};
    fz.cnt_own = v.fuse_cnt + (xg ? 2 : 0);      // [0] arrivals, [1] the flag the waiters poll
    { static const int nap = std::getenv("RESNMTF_FUSE_NAP") ? std::atoi(std::getenv("RESNMTF_FUSE_NAP")) : 2; fz.nap = nap; }
    fz.n_upd = xg ? v.nblkG : v.nblkF; fz.err = h->fuse_err_dev;
    const size_t smem_f = std::min<size_t>(std::max(smem, update_smem_bytes(16)), kMaxLds);
    const bool pp = xg ? v.pp_xg : v.pp_xtf;
    const bool pre = h->opt.fuse_updates == 1;   // (2: without the prefetch of the first X trip)
#define LAUNCH_FUSED_P(UV, XG, PV)                                                                                              \
    if (timed) hipExtLaunchKernelGGL((pass_fused_kernel<UV, XG, PV>), grid, block, smem_f, h->stream, ev0, ev1, 0, a, kf, ks, u, fz);    \
    else hipLaunchKernelGGL((pass_fused_kernel<UV, XG, PV>), grid, block, smem_f, h->stream, a, kf, ks, u, fz)
#define LAUNCH_FUSED(UV, XG) if (pre) { LAUNCH_FUSED_P(UV, XG, true); } else { LAUNCH_FUSED_P(UV, XG, false); }
    if (xg) { if (pp) { LAUNCH_FUSED(4, true); } else { LAUNCH_FUSED(8, true); } }
    else { if (pp) { LAUNCH_FUSED(4, false); } else { LAUNCH_FUSED(8, false); } }
#undef LAUNCH_FUSED
#undef LAUNCH_FUSED_P
    return;
  }
  if (v.half) {       // fp16 image of X: the run-time scale (set at upload) is taken out in the slab store
    a.out_scale = v.u16 ? 1.f / v.xscale : 1.f / (v.xscale * RESNMTF_B16_SCALE);
    // wave-steps per trip: 4 for fp16; 2 for the 16-bit integers (their widening to f32 wants the registers: c2 26.5 k
    // sweeps/s at 2, 23.5 k at 4)
    const int un_def = v.u16 ? 2 : 4;
    const int un = h->opt.half_unroll == 2 ? 2 : (h->opt.half_unroll == 6 ? 6 : (h->opt.half_unroll == 3 ? 3 : (h->opt.half_unroll == 4 ? 4 : un_def)));
#define LAUNCH_HALF(UV, XG, U)                                                                                             \
    if (timed) hipExtLaunchKernelGGL((pass_half_kernel<UV, XG, U>), grid, block, smem, h->stream, ev0, ev1, 0, a, kf, ks);  \
    else hipLaunchKernelGGL((pass_half_kernel<UV, XG, U>), grid, block, smem, h->stream, a, kf, ks)
#define LAUNCH_HALF_U(UV)                                                     \
    if (v.u16) { if (xg) { LAUNCH_HALF(UV, true, true); } else { LAUNCH_HALF(UV, false, true); } } \
    else { if (xg) { LAUNCH_HALF(UV, true, false); } else { LAUNCH_HALF(UV, false, false); } }
    switch (un) {
      case 2: LAUNCH_HALF_U(2); break;
      case 3: LAUNCH_HALF_U(3); break;
      case 6: LAUNCH_HALF_U(6); break;
      default: LAUNCH_HALF_U(4); break;
    }
#undef LAUNCH_HALF_U
#undef LAUNCH_HALF
    return;
  }
#define LAUNCH_PASS_B(NTV, NWV, UV, XG, MA, SP)                                                                              \
  if (timed) hipExtLaunchKernelGGL((pass_kernel<NTV, NWV, UV, XG, MA, SP>), grid, block, smem, h->stream, ev0, ev1, 0, a, kf, ks); \
  else hipLaunchKernelGGL((pass_kernel<NTV, NWV, UV, XG, MA, SP>), grid, block, smem, h->stream, a, kf, ks)
#define LAUNCH_PASS_M(NTV, NWV, UV, XG, MA)                                              \
  if (split == 3) { LAUNCH_PASS_B(NTV, NWV, UV, XG, MA, ((NTV) >= 2 ? 3 : 0)); }         \
  else { LAUNCH_PASS_B(NTV, NWV, UV, XG, MA, 0); }
#define LAUNCH_PASS(NTV, NWV, UV)                                   \
  if (xg && a.kk_block0) { LAUNCH_PASS_M(NTV, NWV, UV, true, true); }      \
  else if (xg) { LAUNCH_PASS_M(NTV, NWV, UV, true, false); }               \
  else if (a.kk_block0) { LAUNCH_PASS_M(NTV, NWV, UV, false, true); }      \
  else { LAUNCH_PASS_M(NTV, NWV, UV, false, false); }
  switch (v.NT * 100 + nw) {
    case 104: LAUNCH_PASS(1, 4, 8); break;
    case 108: if (xg ? v.pp_xg : v.pp_xtf) { LAUNCH_PASS(1, 8, 4); } else { LAUNCH_PASS(1, 8, RESNMTF_K16_UNROLL); } break;
    case 116: LAUNCH_PASS(1, 16, 8); break;
    case 208: LAUNCH_PASS(2, 8, 4); break;
    case 308: LAUNCH_PASS(3, 8, 4); break;
    default: LAUNCH_PASS(4, 8, 4); break;
  }
#undef LAUNCH_PASS
#undef LAUNCH_PASS_M
#undef LAUNCH_PASS_B
}

// a streaming pass without a k x k job (SVD initialisation): mode A kernel, workgroup 0 idles
void launch_pass_plain(resnmtf_handle* h, PassArgs a, int NT, bool xg) {
  a.kk_block0 = 1; a.no_kk = 1; a.check_done = 0;
  const KKFArgs kf{};
  const KKSArgs ks{};
  const dim3 grid(1 + a.ntiles * a.nsplit), block(64 * 8);
  const size_t smem = std::min<size_t>(pass_smem_bytes(16 * NT, 8), kMaxLds);
#define LAUNCH_PLAIN(NTV, UV)                                                                                      \
  if (xg) hipLaunchKernelGGL((pass_kernel<NTV, 8, UV, true, true>), grid, block, smem, h->stream, a, kf, ks);      \
  else hipLaunchKernelGGL((pass_kernel<NTV, 8, UV, false, true>), grid, block, smem, h->stream, a, kf, ks)
  switch (NT) {
    case 1: LAUNCH_PLAIN(1, 8); break;
    case 2: LAUNCH_PLAIN(2, 4); break;
    case 3: LAUNCH_PLAIN(3, 4); break;
    default: LAUNCH_PLAIN(4, 4); break;
  }
#undef LAUNCH_PLAIN
}

template <int NVB>
ChainArgs<NVB> narrow_chain(const ChainArgs<8>& c) {
  ChainArgs<NVB> a{};
  a.len = c.len; a.k = c.k; a.n_views = c.n_views; a.rows_per_block = c.rows_per_block;
  a.n_emit = c.n_emit; a.restricted = c.restricted; a.ctl = c.ctl; a.check_done = c.check_done; a.pstride = c.pstride;
  a.kpack32 = c.kpack32;
  for (int e = 0; e < 4; ++e) { a.W32e[e] = c.W32e[e]; a.parte[e] = c.parte[e]; a.T32e[e] = c.T32e[e]; }
  for (int v = 0; v < NVB; ++v) {
    a.emit_slot[v] = c.emit_slot[v];
    a.W[v] = c.W[v]; a.U[v] = c.U[v]; a.nsplit[v] = c.nsplit[v]; a.Ma[v] = c.Ma[v]; a.Md[v] = c.Md[v]; a.lm[v] = c.lm[v];
    a.sigma[v] = c.sigma[v]; a.n_other[v] = c.n_other[v]; a.cmask[v] = c.cmask[v];
    for (int w = 0; w < NVB; ++w) a.weight[v][w] = c.weight[v][w];
  }
  return a;
}

// kind: 0 = F update, 1 = G update, 2 = mode A run prologue: partials of the current G, nothing updated
void launch_update(resnmtf_handle* h, const ViewState& v, int kind, bool check_done) {
  UpdateArgs a = kind == 0 ? v.argF : v.argG;
  a.check_done = check_done ? 1 : 0;
  a.gram_only = kind == 2 ? 1 : 0;
  if (kind == 2) { a.restricted = 0; a.n_couple = 0; }
  const int nblk = kind == 0 ? v.nblkF : v.nblkG;
  const size_t smem = update_smem_bytes(v.KP);
  const bool emit = v.kk_mode == 0;
  // coupling-count bucket of the kernel instantiation: 0 = unrestricted form, else room for 4, 8 or 16 coupled views
#define LAUNCH_UPD_K(KPV, G_, C_)                                                                                      \
  if (emit) hipLaunchKernelGGL((factor_update_kernel<KPV, G_, C_, true>), dim3(nblk), dim3(update_threads(KPV)), smem, h->stream, a); \
  else hipLaunchKernelGGL((factor_update_kernel<KPV, G_, C_, false>), dim3(nblk), dim3(update_threads(KPV)), smem, h->stream, a)
#define LAUNCH_UPD_C(KPV, G_)                                               \
  if (!a.restricted) { LAUNCH_UPD_K(KPV, G_, 0); }                          \
  else if (a.n_couple <= 4) { LAUNCH_UPD_K(KPV, G_, 4); }                   \
  else if (a.n_couple <= 8) { LAUNCH_UPD_K(KPV, G_, 8); }                   \
  else { LAUNCH_UPD_K(KPV, G_, RESNMTF_MAX_COUPLE); }
#define LAUNCH_UPD(KPV)                                  \
  if (kind == 0) { LAUNCH_UPD_C(KPV, false); }           \
  else { LAUNCH_UPD_C(KPV, true); }
  switch (v.NT) {
    case 1: LAUNCH_UPD(16); break;
    case 2: LAUNCH_UPD(32); break;
    case 3: LAUNCH_UPD(48); break;
    default: LAUNCH_UPD(64); break;
  }
#undef LAUNCH_UPD
#undef LAUNCH_UPD_C
#undef LAUNCH_UPD_K
}

template <int NVB>
WideChainArgs<NVB> narrow_wchain(const WideChainArgs<8>& c) {
  WideChainArgs<NVB> a{};
  a.len = c.len; a.k = c.k; a.n_views = c.n_views; a.ngroups = c.ngroups; a.own = c.own; a.n_self = c.n_self; a.O32 = c.O32; a.o32_stride = c.o32_stride;
  a.W32 = c.W32; a.Wk = c.Wk; a.T32 = c.T32; a.ld32 = c.ld32; a.ctl = c.ctl; a.check_done = c.check_done; a.restricted = c.restricted;
  for (int v = 0; v < NVB; ++v) {
    a.W[v] = c.W[v]; a.U[v] = c.U[v]; a.Ma[v] = c.Ma[v]; a.Md[v] = c.Md[v]; a.lm[v] = c.lm[v]; a.O32v[v] = c.O32v[v];
    a.sigma[v] = c.sigma[v]; a.n_other[v] = c.n_other[v]; a.cmask[v] = c.cmask[v];
    for (int w = 0; w < NVB; ++w) a.weight[v][w] = c.weight[v][w];
  }
  return a;
}
// update_f (g == 0) or update_g (g == 1) of every view in one launch (wide_chain_kernel): the replicated chains (all rows,
// k > 16) or -- sliced -- this rank's row / column slice of them (any k)
void launch_wide_chain(resnmtf_handle* h, int g, bool checked, bool sliced = false) {
  WideChainArgs<8>& c = sliced ? h->schain[g] : h->wchain[g];
  c.check_done = checked ? 1 : 0;
  const int KP = h->views[0].KP;
  const size_t smem = wide_chain_smem_bytes(KP);
  const int ngrid = sliced ? h->schain_grid[g] : h->wchain_grid[g];
  if (ngrid < 1) return;                                   // (an empty slice)
  const int kind = g == 0 ? RESNMTF_TIMED_F_CHAIN : RESNMTF_TIMED_G_CHAIN;
  // sliced: the products-first kernel on 16-row groups (slice_chain_kernel); RESNMTF_SLICE_WIDE=1 keeps the view-by-view
  // walk of wide_chain_kernel on 32-row groups (A/B testing -- same bits either way)
  // Which of the two: a slice_chain workgroup finishes one 16-row group in ~35 us at k = 64 (latency: 4 product stages + 8
  // walk steps), a wide_chain workgroup a 32-row group in ~50 us -- so the 16-row form wins while its groups fit ONE round
  // of resident workgroups (c5 x 8: the G slices, 64 groups; c4 x 4: both chains) and loses when they need two (c5 x 8, F:
  // 392 groups on 256 CUs, 70 against 53 us).  RESNMTF_SLICE_WIDE=1 / =0 forces one form (A/B testing).
  static const char* slice_env = std::getenv("RESNMTF_SLICE_WIDE");
  const int groups16_all = ceil_div(c.len, 16);
  const int n_slots = h->n_cu * (int)std::max<size_t>(1, std::min<size_t>(kMaxLds / slice_chain_smem_bytes(KP, c.n_views <= 4 ? 4 : 8), 2048 / (16 * KP)));
  const bool slice_wide = slice_env ? slice_env[0] == '1' : groups16_all > n_slots;
  // ... and the 16-row form itself in two launches (slice_products_kernel: every (group, pair of views) a workgroup of its own;
  // slice_walk_kernel: the element-wise chain), unless RESNMTF_SLICE_FUSED=1 keeps it in one (slice_chain_kernel) -- same bits
  // Measured (tools/round3/slice_split_ab.sh): c5 x 8, G slice (8 views, k = 64, 63 groups) 36.0 -> 22.4 us; c4 x 4 (4 views,
  // k = 32: two short product stages) 11.2 -> 15.2 / 9.4 -> 10.5 us -- the second launch costs more than the stages it spreads.
  // So: more than two pairs of views and k > 32; RESNMTF_SLICE_FUSED=1 / =0 forces one form.
  static const char* fused_env = std::getenv("RESNMTF_SLICE_FUSED");
  const bool split_form = fused_env ? fused_env[0] == '0' : (c.n_views > 4 && KP > 32);
  if (sliced && !slice_wide && h->slice_nd && split_form) {
    const int groups16 = ceil_div(c.len, 16);
    const unsigned nd_stride = (unsigned)groups16 * 16u * (unsigned)KP;
    const dim3 gridp(groups16, (c.n_views + 1) / 2), gridw(ceil_div(c.len, 4)), block16(16 * KP), blockw(4 * KP);
#define SLSPLIT(KPV, NVBV, ARGS) do { \
    LAUNCH_TIMED(h, kind, (slice_products_kernel<KPV, NVBV>), gridp, block16, 0, ARGS, h->slice_nd, nd_stride); \
    LAUNCH_TIMED(h, kind, (slice_walk_kernel<KPV, NVBV, 4>), gridw, blockw, 0, ARGS, (const double*)h->slice_nd, nd_stride); } while (0)
#define SLSPLIT_K(NVBV, ARGS) do { \
    switch (KP) { case 16: SLSPLIT(16, NVBV, ARGS); break; case 32: SLSPLIT(32, NVBV, ARGS); break; \
                  case 48: SLSPLIT(48, NVBV, ARGS); break; default: SLSPLIT(64, NVBV, ARGS); break; } } while (0)
    if (c.n_views <= 4) { WideChainArgs<4> a4 = narrow_wchain<4>(c); SLSPLIT_K(4, a4); }
    else { SLSPLIT_K(8, c); }
#undef SLSPLIT_K
#undef SLSPLIT
    return;
  }
  if (sliced && !slice_wide) {
    const int groups16 = ceil_div(c.len, 16);
    const dim3 grid16(groups16), block16(16 * KP);          // (one row group per workgroup)
#define SLCHAIN(KPV, NVBV, ARGS) do { const size_t sm = slice_chain_smem_bytes(KPV, NVBV); \
    if (g == 0) LAUNCH_TIMED(h, kind, (slice_chain_kernel<KPV, false, NVBV>), grid16, block16, sm, ARGS); \
    else LAUNCH_TIMED(h, kind, (slice_chain_kernel<KPV, true, NVBV>), grid16, block16, sm, ARGS); } while (0)
#define SLCHAIN_K(NVBV, ARGS) do { \
    switch (KP) { case 16: SLCHAIN(16, NVBV, ARGS); break; case 32: SLCHAIN(32, NVBV, ARGS); break; \
                  case 48: SLCHAIN(48, NVBV, ARGS); break; default: SLCHAIN(64, NVBV, ARGS); break; } } while (0)
    if (c.n_views <= 4) { WideChainArgs<4> a4 = narrow_wchain<4>(c); SLCHAIN_K(4, a4); }
    else { SLCHAIN_K(8, c); }
#undef SLCHAIN_K
#undef SLCHAIN
    return;
  }
  const dim3 grid(ngrid), block(16 * KP);
#define WCHAIN(KPV, NVBV, ARGS) do { \
    if (sliced) { if (g == 0) LAUNCH_TIMED(h, kind, (wide_chain_kernel<KPV, false, NVBV, true>), grid, block, smem, ARGS); \
                  else LAUNCH_TIMED(h, kind, (wide_chain_kernel<KPV, true, NVBV, true>), grid, block, smem, ARGS); } \
    else if (KPV >= 32) { if (g == 0) LAUNCH_TIMED(h, kind, (wide_chain_kernel<(KPV >= 32 ? KPV : 32), false, NVBV>), grid, block, smem, ARGS); \
                          else LAUNCH_TIMED(h, kind, (wide_chain_kernel<(KPV >= 32 ? KPV : 32), true, NVBV>), grid, block, smem, ARGS); } } while (0)
#define WCHAIN_K(NVBV, ARGS) do { \
    switch (KP) { case 16: WCHAIN(16, NVBV, ARGS); break; case 32: WCHAIN(32, NVBV, ARGS); break; \
                  case 48: WCHAIN(48, NVBV, ARGS); break; default: WCHAIN(64, NVBV, ARGS); break; } } while (0)
  if (c.n_views <= 4) {
    WideChainArgs<4> a = narrow_wchain<4>(c);
    WCHAIN_K(4, a);
  } else {
    WCHAIN_K(8, c);
  }
#undef WCHAIN_K
#undef WCHAIN
}

// RESNMTF_PHASE_F_ALL: update_f of every view in view order (one launch when the chain is eligible)
void enqueue_phase_f_all(resnmtf_handle* h, bool checked = false) {
  if (h->wchain_ok[0]) { launch_wide_chain(h, 0, checked); return; }
  if (h->chain_views > 0) {
    h->chain.check_done = checked ? 1 : 0;
    const size_t smem = h->chain_views <= 2 ? f_chain_smem_bytes<2>() : h->chain_views <= 4 ? f_chain_smem_bytes<4>() : f_chain_smem_bytes<8>();
    bool one_slab = true;
    for (int v = 0; v < h->chain_views; ++v) one_slab = one_slab && h->chain.nsplit[v] == 1;
#define LAUNCH_CHAIN(NVB, ARGS)                                                                                         \
    if (one_slab) LAUNCH_TIMED(h, RESNMTF_TIMED_F_CHAIN, (f_chain_kernel<NVB, 1>), dim3(h->chain_blocks), dim3(512), smem, ARGS);  \
    else LAUNCH_TIMED(h, RESNMTF_TIMED_F_CHAIN, (f_chain_kernel<NVB, 4>), dim3(h->chain_blocks), dim3(512), smem, ARGS)
    if (h->chain_views <= 2) {
      ChainArgs<2> a = narrow_chain<2>(h->chain);
      LAUNCH_CHAIN(2, a);
    } else if (h->chain_views <= 4) {
      ChainArgs<4> a = narrow_chain<4>(h->chain);
      LAUNCH_CHAIN(4, a);
    } else {
      LAUNCH_CHAIN(8, h->chain);
    }
#undef LAUNCH_CHAIN
    return;
  }
  for (const auto& v : h->views)
    if (v.owned || v.f_replica) launch_update(h, v, 0, checked);
}

// RESNMTF_PHASE_G_ALL at k <= 16: update_g of every view in one launch (f_chain_kernel, G form) when eligible
bool enqueue_g_chain(resnmtf_handle* h, bool checked) {
  if (h->gchain_views <= 0) return false;
  h->gchain.check_done = checked ? 1 : 0;
  const int nvw = h->gchain_views;
  const size_t smem = nvw <= 2 ? f_chain_smem_bytes<2>() : nvw <= 4 ? f_chain_smem_bytes<4>() : f_chain_smem_bytes<8>();
  const dim3 grid(h->gchain_blocks), block(512);
  if (nvw <= 2) { ChainArgs<2> a = narrow_chain<2>(h->gchain); LAUNCH_TIMED(h, RESNMTF_TIMED_G_CHAIN, (f_chain_kernel<2, 1, true>), grid, block, smem, a); }
  else if (nvw <= 4) { ChainArgs<4> a = narrow_chain<4>(h->gchain); LAUNCH_TIMED(h, RESNMTF_TIMED_G_CHAIN, (f_chain_kernel<4, 1, true>), grid, block, smem, a); }
  else LAUNCH_TIMED(h, RESNMTF_TIMED_G_CHAIN, (f_chain_kernel<8, 1, true>), grid, block, smem, h->gchain);
  return true;
}

// ---- the phases of one view (see the header comment)
void enqueue_phase_f(resnmtf_handle* h, const ViewState& v, bool checked) { launch_update(h, v, 0, checked); }
// replicate_f: the X.G split slabs of an owned view -> the one f32 slab of its exchange block (what every
// rank's F update of this view reads, the owner's included: a third of the bytes on the wire at c2)
void launch_fold(resnmtf_handle* h, const ViewState& v) {
  if (!v.Usum) return;
  const int quads = v.n_pad * v.KP / 4;
  LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slab_fold_kernel, dim3(ceil_div(quads, 256)), dim3(256), 0, v.Pxg, v.nsplit_xg, quads, v.Usum);
}
void launch_fold_t(resnmtf_handle* h, const ViewState& v) {
  if (!v.Tsum) return;
  const int quads = v.m_pad * v.KP / 4;
  LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slab_fold_kernel, dim3(ceil_div(quads, 256)), dim3(256), 0, v.Pxtf, v.nsplit_xtf, quads, v.Tsum);
}
// slice_chains: the own view's pass result, folded and cut into the V chunks of the next all-to-all (slice_pack_kernel)
void launch_slice_pack(resnmtf_handle* h, const ViewState& v, bool xg, bool checked) {
  SlicePackArgs a{};
  a.P = xg ? v.Pxg : v.Pxtf; a.nsplit = xg ? v.nsplit_xg : v.nsplit_xtf; a.rows_pad = xg ? v.n_pad : v.m_pad;
  a.KP = v.KP; a.NT = v.NT;
  a.rows_per_slice = xg ? h->sl_rows : h->sl_cols; a.n_slices = h->opt.slice_count;
  a.out = xg ? h->u_send : h->t_send; a.chunk_bytes = xg ? h->u_chunk : h->t_chunk;
  if (!xg) { a.tail[0] = v.Ma_G; a.tail[1] = v.Md_G; a.tail_count = v.k * v.k; a.T32 = v.T32; a.ld32 = 64; }
  if (h->opt.slice_p2p) {      // chunk c straight into rank c's receive slot for this rank; U: the own S block into every rank's arena
    const int r = h->opt.slice_index, vi = (int)(&v - h->views.data());
    a.out = nullptr;
    for (int c = 0; c < a.n_slices; ++c) {
      const resnmtf_handle::Peer& pc = h->peers[(size_t)c];
      a.outv[c] = (xg ? pc.u_recv : pc.t_recv) + (size_t)r * a.chunk_bytes;
      if (xg) a.tailv[c] = pc.sblk + (size_t)vi * h->sblk_stride;
    }
    if (xg) { a.tail[0] = v.sblk; a.tail[1] = v.sblk + h->sblk_stride / 2; a.tail_count = (int)(h->sblk_stride / 2); }
  }
  a.ctl = h->ctl; a.check_done = checked ? 1 : 0;
  const size_t quads = (size_t)a.n_slices * a.rows_per_slice * (a.KP / 4);
  LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slice_pack_kernel, dim3((unsigned)std::min<size_t>((quads + 255) / 256, 65535)), dim3(256), 0, a);
}
// slice_p2p: one arrival on every rank's counter of exchange e (stream-ordered behind the kernels that stored the data) /
// the stream waits until `arrivals` of them are in
void p2p_signal(resnmtf_handle* h, int e) {
  SliceSignalArgs a{};
  a.n = h->V;
  for (int c = 0; c < h->V; ++c) a.flag[c] = h->peers[(size_t)c].flags + e;
  hipLaunchKernelGGL(slice_signal_kernel, dim3(1), dim3(64), 0, h->stream, a);
}
// the stream goes on when counter e has V * (waits of this call site so far + add) arrivals.  slice_p2p = 1: the command
// processor waits (hipStreamWaitValue32; the host's sweep index numbers the wait); 2: p2p_wait_kernel (site counters on the device)
hipError_t p2p_wait(resnmtf_handle* h, int e, int site, int sweep, int add) {
  if (h->opt.slice_p2p != 2)
    return hipStreamWaitValue32(h->stream, h->p2p_flags + e, (unsigned)h->V * (unsigned)(sweep + add), hipStreamWaitValueGte, 0xFFFFFFFFu);
  P2pWaitArgs a{};
  a.flag = h->p2p_flags + e; a.site = h->p2p_flags + 16 + site; a.per_wait = (unsigned)h->V; a.add = (unsigned)add; a.err = h->fuse_err_dev;
  hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, h->stream, a);
  return hipGetLastError();
}
// block_p2p: byte ranges [off0, off0 + b0) and [off1, off1 + b1) of this rank's part of an arena -> the same place on every peer
// kind 0: F exchange arena, 1: G exchange arena, 2: S block arena
void launch_block_push(resnmtf_handle* h, int kind, size_t off0, size_t b0, size_t off1, size_t b1) {
  BlockPushArgs a{};
  a.src = static_cast<const char*>(kind == 0 ? h->fblk_arena : kind == 1 ? h->gblk_arena : (void*)h->sblk_arena);
  for (int c = 0; c < h->V; ++c) {
    if (c == h->opt.slice_index) continue;
    const resnmtf_handle::Peer& pc = h->peers[(size_t)c];
    a.dst[a.n_dst++] = kind == 0 ? pc.fblk_arena : kind == 1 ? pc.gblk_arena : reinterpret_cast<char*>(pc.sblk);
  }
  if (a.n_dst == 0) return;
  a.off[0] = off0; a.bytes[0] = b0; a.off[1] = off1; a.bytes[1] = b1;
  const size_t quads = (b0 + b1) / 16;
  LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, block_push_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((quads + 255) / 256, 2048))), dim3(256), 0, a);
}
// the F exchange block of an owned view to the peers: everything (replicate_f alone: the owner's coefficients and lambda are
// the only copy) or, with the replicated S chain, the U rows and the embedded S block only -- every rank computes the
// coefficients, lambda and mu of every view itself and a late store must not land on them
void push_f_block(resnmtf_handle* h, const ViewState& v) {
  const size_t base = (size_t)(static_cast<const char*>(v.fblk) - static_cast<const char*>(h->fblk_arena));
  if (!h->opt.replicate_gs) { launch_block_push(h, 0, base, v.fblk_bytes, 0, 0); return; }
  const size_t usum = ((size_t)v.n_pad * v.KP * sizeof(float) + 255) / 256 * 256;
  if (h->sblk_embedded) {
    const size_t s_off = (size_t)(reinterpret_cast<const char*>(v.sblk) - static_cast<const char*>(h->fblk_arena));
    launch_block_push(h, 0, base, usum, s_off, h->sblk_stride * sizeof(double));
  } else {
    launch_block_push(h, 0, base, usum, 0, 0);
    const size_t sb = (h->sblk_stride * sizeof(double));
    launch_block_push(h, 2, (size_t)(&v - h->views.data()) * sb, sb / 16 * 16, 0, 0);
  }
}
void push_g_block(resnmtf_handle* h, const ViewState& v) {      // [Tsum | Ma_G | Md_G], not mu
  const size_t base = (size_t)(static_cast<const char*>(v.gblk) - static_cast<const char*>(h->gblk_arena));
  const size_t tsum = ((size_t)v.m_pad * v.KP * sizeof(float) + 255) / 256 * 256;
  launch_block_push(h, 1, base, (tsum + 2 * (size_t)v.k * v.k * sizeof(double)) / 16 * 16, 0, 0);
}
// slice_chains: the own view's new F (g == 0) / G (g == 1) rows, as received, -> the operand copies of the next pass
void launch_slice_unpack(resnmtf_handle* h, const ViewState& v, int g, bool checked) {
  SliceUnpackArgs a{};
  a.in = g == 0 ? h->f_recv : h->g_recv; a.len = g == 0 ? v.n : v.m; a.k = v.k;
  a.W32 = g == 0 ? v.F32 : v.G32; a.ld32 = 64; a.Wk = g == 0 ? v.Fk : v.Gk;
  a.ctl = h->ctl; a.check_done = checked ? 1 : 0;
  const dim3 grid(ceil_div(a.len, 32)), block(256);
  switch (v.KP) {
    case 16: LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slice_unpack_kernel<16>, grid, block, 0, a); break;
    case 32: LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slice_unpack_kernel<32>, grid, block, 0, a); break;
    case 48: LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slice_unpack_kernel<48>, grid, block, 0, a); break;
    default: LAUNCH_TIMED(h, RESNMTF_TIMED_PACK, slice_unpack_kernel<64>, grid, block, 0, a); break;
  }
}
// replicate_gs: the second half of the k x k job of EVERY view (update_s chain, update_lm, error, F coefficients)
int launch_s_chain(resnmtf_handle* h, bool checked) {
  SChainArgs a{};
  const int V = h->V;
  const ViewState& v0 = h->views[0];
  a.k = v0.k; a.n_views = V;
  a.view_sweep = h->view_sweep; a.ticket = h->view_sweep + V;
  a.tol = h->phase_tol; a.check_done = checked ? 1 : 0;
  a.sblocks = h->sblk_base; a.sblock_stride = h->sblk_step;
  double sum_xi = 0.0;
  for (double x : h->xi) sum_xi += x;
  a.restricted = sum_xi != 0.0 ? 1 : 0;
  for (int w = 0; w < V; ++w) {
    const ViewState& vs = h->views[w];
    a.S[w] = vs.S; a.lambda[w] = vs.lambda; a.mu[w] = vs.mu; a.Ma_F[w] = vs.Ma_F; a.Md_F[w] = vs.Md_F;
    double sg = 0.0;
    for (int c = 0; c < V; ++c) {
      const double wgt = h->xi[(size_t)c + (size_t)w * V];                          // xi[c, w]
      a.xi[c][w] = (c == w) ? 0.0 : wgt;
      sg += wgt;
    }
    a.sigma[w] = sg;
  }
  a.err = h->err; a.err_stride = V; a.err_cap = h->err_cap; a.err_host = h->err_host_dev;
  a.ctl = h->ctl; a.ctl_host = h->ctl_host_dev;
  const size_t smem = kk_smem_bytes(v0.KP, 16);
#define S_CHAIN(KPV) do { \
    if (V <= 4) LAUNCH_TIMED(h, RESNMTF_TIMED_S_CHAIN, (s_chain_kernel<KPV, 4>), dim3(V), dim3(s_chain_threads(KPV)), smem, a); \
    else if (V <= 8) LAUNCH_TIMED(h, RESNMTF_TIMED_S_CHAIN, (s_chain_kernel<KPV, 8>), dim3(V), dim3(s_chain_threads(KPV)), smem, a); \
    else LAUNCH_TIMED(h, RESNMTF_TIMED_S_CHAIN, (s_chain_kernel<KPV, RESNMTF_MAX_COUPLE + 1>), dim3(V), dim3(s_chain_threads(KPV)), smem, a); } while (0)
  switch (v0.NT) {
    case 1: S_CHAIN(16); break;
    case 2: S_CHAIN(32); break;
    case 3: S_CHAIN(48); break;
    default: S_CHAIN(64); break;
  }
#undef S_CHAIN
  return RESNMTF_OK;
}
// fuse_f: the view's F update has NOT been enqueued -- it rides in the Xt.F launch (enqueue_sweep decides)
void enqueue_phase_g(resnmtf_handle* h, const ViewState& v, double tol, bool checked, bool fuse_f = false) {
  launch_pass(h, v, false, 1, tol, checked, fuse_f);
  const bool fuse_g = can_fuse_update(h, v, 1);
  if (!fuse_g) launch_update(h, v, 1, checked);
  launch_pass(h, v, true, 1, tol, checked, fuse_g);
  launch_fold(h, v);
}
// run prologue of one view: X.G launch whose kk_s runs in mode 0 (F coefficients from the current S, G);
// mode A first emits the fp64 partials of the current G so that a resumed run is bitwise identical
void enqueue_prologue(resnmtf_handle* h, const ViewState& v) {
  if (v.kk_mode == 0) launch_update(h, v, 2, false);
  launch_pass(h, v, true, 0, -1.0, false);
  launch_fold(h, v);
  if (h->sliced) launch_slice_pack(h, v, true, false);
  if (h->sliced && h->opt.slice_p2p) p2p_signal(h, 0);
}

void enqueue_sweep(resnmtf_handle* h, double tol) {
  const bool checked = tol >= 0.0;
  // F_w' reads neither G nor S of the same sweep: when the fused chain applies (several k <= 16 views sharing
  // their rows in the same order) every F update of the sweep runs first, in one launch
  const bool hoist = h->chain_views > 0 && h->all_owned;
  if (hoist) enqueue_phase_f_all(h, checked);
  for (const auto& v : h->views) {
    if (!v.owned) continue;
    const bool fuse_f = !hoist && can_fuse_update(h, v, 0);
    if (!hoist && !fuse_f) enqueue_phase_f(h, v, checked);
    enqueue_phase_g(h, v, tol, checked, fuse_f);
  }
}

void destroy_graphs(resnmtf_handle* h) {
  for (auto& rung : h->ladder)
    if (rung.second) (void)hipGraphExecDestroy(rung.second);
  h->ladder.clear();
  for (auto& g : h->exact)
    if (g.second) (void)hipGraphExecDestroy(g.second);
  h->exact.clear();
  h->graph_tol = -2.0;
}

int capture_graph(resnmtf_handle* h, int sweeps, double tol, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < sweeps; ++s) enqueue_sweep(h, tol);
  HIP_TRY(h, hipStreamEndCapture(h->stream, &g));
  hipError_t e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return h->fail_hip("hipGraphInstantiate", e);
  (void)hipGraphUpload(*out, h->stream);      // first replay does not pay the upload
  return RESNMTF_OK;
}
// the whole ladder at once (a few hundred kernel nodes): a later run of any length never captures inside a timed region
int capture_ladder(resnmtf_handle* h, int batch, double tol) {
  for (auto& rung : h->ladder)
    if (rung.second) (void)hipGraphExecDestroy(rung.second);
  h->ladder.clear();
  std::vector<int> rungs{batch};
  int p2 = 1;
  while (p2 * 2 < batch) p2 *= 2;
  for (; p2 >= 1; p2 /= 2)
    if (p2 < batch) rungs.push_back(p2);
  for (int sweeps : rungs) {
    hipGraphExec_t ex = nullptr;
    if (int rc = capture_graph(h, sweeps, tol, &ex)) { destroy_graphs(h); return rc; }
    h->ladder.emplace_back(sweeps, ex);
  }
  return RESNMTF_OK;
}

// RESNMTF_PHASE_LOCAL_SWEEP: every F update this handle holds inputs for, then PHASE_G of every owned view
void enqueue_local_sweep(resnmtf_handle* h) {
  enqueue_phase_f_all(h);
  for (const auto& v : h->views)
    if (v.owned) enqueue_phase_g(h, v, -1.0, false);
}
// (Replaying these five launches from a hipGraph between two RCCL collectives was measured SLOWER than the plain
// launches: 61.8 against 54.8 us per sweep with one rank -- a graph launch per sweep costs more than it saves.)
int launch_local_sweep(resnmtf_handle* h) {
  enqueue_local_sweep(h);
  return RESNMTF_OK;
}

int flush_timing(resnmtf_handle* h) {
  if (h->ev_used == 0) return RESNMTF_OK;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    const int kind = h->ev_kind[i / 2];
    if (kind == RESNMTF_TIMED_XG) { h->timing.xg_ms_total += ms; h->timing.xg_launches++; }
    else if (kind == RESNMTF_TIMED_XTF) { h->timing.xtf_ms_total += ms; h->timing.xtf_launches++; }
    if (kind >= 0 && kind < RESNMTF_TIMED_KINDS) { h->ktime_ms[kind] += ms; h->klaunch[kind]++; }
  }
  h->ev_used = 0;
  return RESNMTF_OK;
}

// sizes a streaming pass.  Measured on MI355X (tools/sweep_pass.py, tools/stamps.py, tools/micro/):
// 8 waves per workgroup.  `slots` = workgroups the device holds at once (CUs x resident workgroups
// per CU, minus the k x k / aux workgroups of the launch).
//   * one-round geometry: if the whole pass fits the slots with splits of at most 2048 rows, use
//     floor(slots / ntiles) equal splits -- every workgroup is resident from t = 0, none waits for a
//     slot and no CU is left with half the work of its neighbour (the quantisation that cost the
//     628-workgroup / 255-slot X.G launch of c2 a third of its time);
//   * otherwise 16-step workgroups (512 rows; longer for k > 32) that the dispatcher streams through
//     the slots.
// Splits are capped at 16, the depth of the consumer's prefetch.
void size_pass(int NT, int ntiles, int rows_pad, int slots, int max_nw, int force_nw, int force_ns, bool fine,
               int* nsplit, int* rps, int* nw) {
  int w = std::min(8, max_nw);
  if (max_nw > 8 && (force_nw == 4 || force_nw == 8 || force_nw == 16)) w = force_nw;
  const int quantum = 4 * w * 8;                       // rows of one unrolled trip of a workgroup
  // streamed geometry: 16-step workgroups (512 rows); for k > 32 only about 8 workgroups per slot, i.e.
  // longer splits -- there the per-workgroup epilogue (tree sum of 64 accumulator registers per lane +
  // slab store) costs as much as several trips (c5 X.G pass: 744 -> 602 us; k <= 32 prefers short splits)
  // k > 16 (pass_body_k32): a workgroup trip is 256 rows, so splits are whole trips where the extent allows
  // k <= 16, f32 image: a split is any whole number of 4 w-row steps (the ragged last trip is masked), so splits come out
  // nearly equal -- c2 X^T.F: 15 x 672 rows instead of 14 x 704 + 192.  Measured neutral (16.4 us either way: with 480
  // workgroups in flight the launch runs at what the memory system delivers, 5.3 TB/s between ramp and drain); kept for
  // the even timeline.  The 2-byte images step in 16 w rows and keep 64
  const int gran = NT >= 2 ? quantum : (fine ? 4 * w : 64);
  int r = 2 * quantum;
  if (NT >= 3) {
    const int ns_stream = std::max(1, std::min(16, ceil_div(8 * (slots + 1), std::max(ntiles, 1))));
    r = std::max(r, round_up(ceil_div(rows_pad, ns_stream), gran));
  }
  if (ceil_div(rows_pad, r) > 16) r = round_up(ceil_div(rows_pad, 16), quantum);
  const int ns_one = std::min(16, slots / std::max(ntiles, 1));
  if (ns_one >= 1) {
    const int r_one = round_up(ceil_div(rows_pad, ns_one), gran);
    if (r_one <= 2048 && r_one >= quantum) r = r_one;
  }
  if (force_ns > 0) r = round_up(ceil_div(rows_pad, force_ns), 64);
  r = std::min(r, round_up(rows_pad, 64));
  *rps = r;
  *nsplit = ceil_div(rows_pad, r);
  *nw = w;
}
// k > 16, wide form (pass_body_wide): a workgroup = TW tiles x 8 / TW row groups, one workgroup per CU.  Picks TW in
// {8, 4} and the number of row splits (<= 16, the depth of the consumer's prefetch) so that the grid fills whole rounds
// of the CUs -- 294 workgroups on 256 CUs cost a c5 X.G pass 468 us against 343 us with 490 (tools/micro/pass_k32_lab.hip)
// -- preferring wide workgroups (the B block is read once per workgroup) and few splits (slab traffic).
void size_aux(int rows_pad, int nw, int max_splits, int* nsplit, int* rps);
// k > 16, wide form (pass_body_wide): a workgroup = TW tiles x 8 / TW row groups, one workgroup per CU.  Picks TW in
// {8, 4}, the number of row splits (<= 16, the depth of the consumer's prefetch) and -- hand-off mode B -- the row splits
// of the aux tiles by a small model of the launch: main workgroups of equal length dealt greedily to the CUs as they
// become free, the aux workgroups (they head the grid) holding one CU each for t_aux and the k x k job one more for
// t_kk behind the last of them.  What the model has to get right (tools/stamps.py, tools/micro/pass_k32_lab.hip on c5):
//   * 294 main workgroups on 256 CUs are two rounds, the second almost empty: 468 us against 343 us with 490;
//   * 32 aux workgroups of 45-100 us in front of 240 main ones: 16 main workgroups start that much later, and the one
//     that gets the k x k job's CU started at 184 us and ended the launch at 442 us where the others ended at 320;
//   * CUs that only ran an aux workgroup idle for the rest of a one-round launch.
// Fewer splits mean less slab traffic (2 KP / rows of the X bytes per split) and fewer per-workgroup prologues.
struct WidePlan { int tw = 8, nsplit = 1, rps = 64, nsaux = 1, rpsaux = 64; double makespan = 1e300; };
// Workgroups are dealt round-robin to the 8 XCDs whatever their load (MI355X_MICROARCH.md, Workgroup dispatch): XCD x
// serves the workgroups i = x (mod 8) of the grid, in order, with ITS CUs -- a workgroup starts on the CU of its XCD that
// becomes free first.  Grid order: the aux workgroups (t_aux each; the k x k job keeps the CU of the last one t_kk longer),
// then the main workgroups, split-major; all splits have d_long except the last (d_last).
double wide_makespan(int ntg, int nsplit, double d_long, double d_last, int n_cu, int aux_wgs, double t_aux, double t_kk) {
  const int cus = std::max(1, n_cu / 8);
  std::vector<std::vector<double>> free_at(8, std::vector<double>((size_t)cus, 0.0));
  auto place = [&](int index, double d) {
    std::vector<double>& f = free_at[(size_t)(index & 7)];
    size_t cu = 0;
    for (size_t c = 1; c < f.size(); ++c)
      if (f[c] < f[cu]) cu = c;
    f[cu] += d;
    return f[cu];
  };
  double end = 0.0;
  for (int i = 0; i < aux_wgs; ++i) end = std::max(end, place(i, i == aux_wgs - 1 ? t_aux + t_kk : t_aux));
  for (int s = 0; s < nsplit; ++s)
    for (int g = 0; g < ntg; ++g) end = std::max(end, place(aux_wgs + s * ntg + g, s == nsplit - 1 ? d_last : d_long));
  // (the estimate of the main tiles is pessimistic for views that partly fit the Infinity Cache: keep the aux + k x k
  // path well inside it)
  return std::max(end, aux_wgs > 0 ? 1.5 * (t_aux + t_kk) : 0.0);
}
WidePlan plan_wide(int ntiles, int rows_pad, int KP, int kinds /* aux products, 0 = hand-off mode A */, int n_cu, int force_ns, int nw) {
  const double t_pass = 4.0 * ntiles * 64.0 * rows_pad / 5.0e6;          // us for the X bytes at 5 TB/s
  // the k x k job of the last-arriving aux workgroup (tools/stamps.py, after its slab loads were batched): Xt.F launch (two aux
  // kinds) 6 us at k = 32, 18 at k = 64; X.G launch (three kinds, the S rule and the error inside) 14 / 49 us; hand-off mode A
  // (job in workgroup 0 for the whole launch): the earlier, slower figure
  const double r3 = (KP / 64.0) * (KP / 64.0) * (KP / 64.0);
  // (round 3, k x k products with conflict-free operand reads: 14.8 / 32 us at k = 64, 6 / 13 us at k = 32 -- profiles/r03_stamps_c5_kk.txt)
  const double t_kk = kinds == 3 ? 9.0 + 23.0 * r3 : kinds == 2 ? 4.5 + 10.3 * r3 : 12.0 + 50.0 * r3;
  WidePlan best;
  double best_aux_time = 1e300;
  int last_na = -1;
  for (int want : {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}) {
    int na = 1, ra = 64;
    size_aux(rows_pad, nw, want, &na, &ra);
    if (kinds > 0 && na == last_na) continue;
    last_na = na;
    const int aux_wgs = kinds > 0 ? kinds * na : 1;                       // mode A: workgroup 0 is the k x k job
    const double t_aux = kinds > 0 ? 2.0 * 256.0 * ra / 25.0e3 + 5.0 : 0.0;   // both operands at one CU's ~25 GB/s share
    if (aux_wgs > n_cu / 2) break;
    for (int tw : {8, 4}) {
      const int trip = 32 * (8 / tw), ntg = ceil_div(ntiles, tw);
      for (int ns = 1; ns <= 16; ++ns) {
        if (force_ns > 0 && ns != force_ns) continue;
        // equal splits, or a SHORT last split: its workgroups are the last of the grid and start on the CUs the aux
        // workgroups (and the k x k job) free -- the launch then ends with every CU busy instead of leaving those idle
        for (double last_share : {1.0, 0.85, 0.7, 0.55, 0.4}) {
          // (a short last split is there to fill the CUs the aux workgroups free: about as many workgroups as those)
          if (last_share < 1.0 && (ns < 2 || force_ns > 0 || ntg > aux_wgs + aux_wgs / 2)) continue;
          const int r = round_up((int)std::ceil(rows_pad / (ns - 1 + last_share)), trip);
          const int ns_real = ceil_div(rows_pad, r);
          if (ns_real != ns && force_ns <= 0) continue;                   // (a shorter list of splits is scored under its own count)
          const int r_last = rows_pad - (ns_real - 1) * r;
          const double per_row = t_pass * n_cu * (1.0 + ns_real * 2.0 * KP / rows_pad) * (tw == 8 ? 1.0 : 1.03) / ((double)ntg * rows_pad);   // CU us per row of a workgroup
          const double ms = wide_makespan(ntg, ns_real, per_row * r + 4.0, per_row * r_last + 4.0, n_cu, aux_wgs, t_aux,
                                          kinds > 0 ? t_kk : t_kk + 10.0);
          // ties (within 1 %) go to the plan whose aux workgroups and k x k job are done first: which aux workgroup arrives
          // last -- and hosts the job -- is not known, and a main workgroup that has to wait for that CU in a launch without
          // slack ends it late (c5 X.G with 3 aux workgroups of 180 us: job done at 245 us, last main workgroup 245 -> 395 us
          // where the others ended at 337)
          const bool tie = ms < best.makespan * 1.01 && ms >= best.makespan * 0.99;
          if (tie && t_aux >= best_aux_time) continue;
          if (!tie && ms >= best.makespan) continue;
          best_aux_time = t_aux;
          best.makespan = std::min(ms, best.makespan); best.tw = tw; best.nsplit = ns_real; best.rps = r; best.nsaux = na; best.rpsaux = ra;
        }
      }
    }
    if (kinds == 0) break;
  }
  return best;
}
// (wide form, k > 16: at most 16 -- the k x k job sums the slabs of every split while the pass saturates the memory
// system, and with one workgroup per CU and one or two rounds of long main workgroups its CU is missing for as long as
// it runs: c5 passes 409 / 486 us with 32 / 49 aux splits against 326 / 313 us for the main tiles alone)
void size_aux(int rows_pad, int nw, int max_splits, int* nsplit, int* rps) {
  const int quantum = 4 * nw * 8;
  int r = quantum;
  if (ceil_div(rows_pad, r) > max_splits) r = round_up(ceil_div(rows_pad, max_splits), quantum);
  r = std::min(r, round_up(rows_pad, 64));
  *rps = r;
  *nsplit = ceil_div(rows_pad, r);
}

int sync_both(resnmtf_handle* h) {
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return RESNMTF_OK;
}

// pinned, device-mapped host mirror of the per-sweep errors ([cap][V]) and of the loop control
hipError_t alloc_host_mirrors(resnmtf_handle* h, int cap) {
  if (h->err_host) (void)hipHostFree(h->err_host);
  h->err_host = nullptr; h->err_host_dev = nullptr;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h->err_host), (size_t)cap * h->V * sizeof(double),
                               hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return e;
  std::memset(h->err_host, 0, (size_t)cap * h->V * sizeof(double));
  if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->err_host_dev), h->err_host, 0)) != hipSuccess) return e;
  if (!h->fuse_err) {
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->fuse_err), sizeof(int), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return e;
    *h->fuse_err = 0;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->fuse_err_dev), h->fuse_err, 0)) != hipSuccess) return e;
  }
  if (!h->ctl_host) {
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->ctl_host), sizeof(SweepCtl), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return e;
    std::memset(h->ctl_host, 0, sizeof(SweepCtl));
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->ctl_host_dev), h->ctl_host, 0)) != hipSuccess) return e;
  }
  return hipSuccess;
}

int ensure_err_capacity(resnmtf_handle* h, int sweeps) {
  if (sweeps <= h->err_cap) return RESNMTF_OK;
  if (int rc = sync_both(h)) return rc;
  if (h->err) (void)hipFree(h->err);
  h->err = nullptr;
  const int cap = std::max(sweeps, 1024);
  hipError_t e = dev_alloc_zero(&h->err, (size_t)cap * h->V);
  if (e == hipSuccess) e = hipDeviceSynchronize();      // (NULL-stream memset vs the handle's non-blocking stream)
  if (e == hipSuccess) e = alloc_host_mirrors(h, cap);
  if (e != hipSuccess) { h->err_cap = 0; return h->fail_hip("hipMalloc err", e); }
  h->err_cap = cap;
  h->prepared = false;   // kernel argument blocks hold the pointer
  return RESNMTF_OK;
}

}  // namespace

extern "C" {

int resnmtf_abi_version(void) { return RESNMTF_ABI_VERSION; }

int resnmtf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void resnmtf_default_options(resnmtf_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (int)sizeof(*o);
  o->device_id = 0;
  o->stream = nullptr;
  o->use_graph = 1;
  o->check_every = 32;     // (sweeps after the stop test fired are empty launches: ~2 us each)
}

const char* resnmtf_last_error(const resnmtf_handle* h) {
  return h ? h->last_error.c_str() : g_create_error.c_str();
}

int resnmtf_create(int n_views, const int* n_rows, const int* n_cols, const int* k, const int* owned,
                   const resnmtf_options* opts, resnmtf_handle** out) {
  if (!out) { g_create_error = "out is NULL"; return RESNMTF_ERR_INVALID; }
  *out = nullptr;
  if (n_views < 1 || !n_rows || !n_cols || !k) { g_create_error = "bad view description"; return RESNMTF_ERR_INVALID; }
  if (n_views > RESNMTF_MAX_COUPLE + 1) { g_create_error = "too many views (max 17)"; return RESNMTF_ERR_INVALID; }
  for (int v = 0; v < n_views; ++v) {
    if (n_rows[v] < 1 || n_cols[v] < 1) { g_create_error = "view dimensions must be positive"; return RESNMTF_ERR_INVALID; }
    if (k[v] < 1 || k[v] > RESNMTF_MAX_K) { g_create_error = "k must be in [1, 64]"; return RESNMTF_ERR_INVALID; }
    if (k[v] > n_cols[v] || k[v] > n_rows[v]) { g_create_error = "k exceeds a view dimension (R/utils.r:444,449)"; return RESNMTF_ERR_INVALID; }
  }
  resnmtf_options o;
  resnmtf_default_options(&o);
  if (opts) {
    if (opts->struct_size != (int)sizeof(resnmtf_options)) { g_create_error = "options struct_size mismatch"; return RESNMTF_ERR_INVALID; }
    o = *opts;
  }
  if (o.pass_splits_xg < 0 || o.pass_splits_xg > 16 || o.pass_splits_xtf < 0 || o.pass_splits_xtf > 16) {
    g_create_error = "pass_splits_xg / pass_splits_xtf must be in 0 ... 16 (0 = the launch model's choice)";      // (a forced count beyond
    return RESNMTF_ERR_INVALID;                                                                          //  the planner's range found no plan)
  }
  if (o.target_workgroups != 0 && o.target_workgroups < 64) {      // (the launch planner needs room for the aux workgroups of the k > 16 passes)
    g_create_error = "target_workgroups must be 0 (the device's own figure) or at least 64";
    return RESNMTF_ERR_INVALID;
  }
  if (o.bf16_split == 1) { g_create_error = "bf16_split = 1 (the two-piece form) is retired: use 0 (three pieces, f32-grade) or 2 (f32 MFMA)"; return RESNMTF_ERR_INVALID; }
  if (o.slice_chains) {
    const char* why = nullptr;
    if (!o.replicate_f || !o.replicate_gs) why = "slice_chains needs replicate_f and replicate_gs";
    else if (o.slice_count != n_views || n_views < 1 || n_views > 8) why = "slice_chains needs slice_count = number of views <= 8";
    else if (o.slice_index < 0 || o.slice_index >= o.slice_count) why = "slice_index out of range";
    else if (o.x_half != 0 || o.kk_mode == 1) why = "slice_chains uses the f32 images and hand-off mode B";
    else if (!owned) why = "slice_chains needs exactly one owned view (view index = slice_index)";
    else
      for (int v = 0; v < n_views && !why; ++v) {
        if (n_rows[v] != n_rows[0] || n_cols[v] != n_cols[0] || k[v] != k[0]) why = "slice_chains needs equal shapes and k in all views";
        else if ((owned[v] != 0) != (v == o.slice_index)) why = "slice_chains needs exactly one owned view (view index = slice_index)";
      }
    if (why) { g_create_error = why; return RESNMTF_ERR_INVALID; }
  }
  if (o.slice_p2p && !o.slice_chains) {      // peer stores for the exchange blocks of the replicated layouts
    const char* why = nullptr;
    if (!o.replicate_f) why = "slice_p2p needs slice_chains or the replicated chains (replicate_f)";
    else if (o.slice_count != n_views || n_views < 1 || n_views > 8) why = "slice_p2p needs slice_count = number of views <= 8";
    else if (o.slice_index < 0 || o.slice_index >= o.slice_count) why = "slice_index out of range";
    else if (!owned) why = "slice_p2p needs exactly one owned view (view index = slice_index)";
    else
      for (int v = 0; v < n_views && !why; ++v)
        if ((owned[v] != 0) != (v == o.slice_index)) why = "slice_p2p needs exactly one owned view (view index = slice_index)";
    if (why) { g_create_error = why; return RESNMTF_ERR_INVALID; }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_error = "no HIP device available (this library has no CPU fallback)";
    return RESNMTF_ERR_NO_DEVICE;
  }
  if (o.device_id < 0 || o.device_id >= ndev) { g_create_error = "device_id out of range"; return RESNMTF_ERR_INVALID; }
  hipError_t e = hipSetDevice(o.device_id);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return RESNMTF_ERR_HIP; }
  hipDeviceProp_t prop;
  int n_cu = 256;
  if (hipGetDeviceProperties(&prop, o.device_id) == hipSuccess) {
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
      return RESNMTF_ERR_NO_DEVICE;
    }
    if (prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
  }
  auto* h = new resnmtf_handle();
  h->n_cu = n_cu;
  h->V = n_views;
  h->opt = o;
  if (h->opt.check_every < 1) h->opt.check_every = 8;
  h->views.resize(n_views);
  h->phi.assign((size_t)n_views * n_views, 0.0);
  h->xi = h->phi;
  h->psi = h->phi;
  for (int v = 0; v < n_views; ++v) {
    ViewState& vs = h->views[v];
    vs.n = n_rows[v]; vs.m = n_cols[v]; vs.k = k[v];
    vs.NT = ceil_div(k[v], 16); vs.KP = 16 * vs.NT;
    vs.n_pad = round_up(vs.n, 64); vs.m_pad = round_up(vs.m, 64);
    // tile-major images: tile t of X32 = columns 64 t .. 64 t + 63 of X as [n_pad][64], contiguous; one extra
    // 256-B row per tile keeps the tile starts off a common power-of-two stride (memory channels)
    const size_t pad_rows = o.no_pitch_pad ? 0 : 1;
    vs.ldx = ((size_t)vs.n_pad + pad_rows) * 64; vs.ldxt = ((size_t)vs.m_pad + pad_rows) * 64;
    vs.x32_floats = (size_t)(vs.m_pad / 64) * vs.ldx; vs.xt32_floats = (size_t)(vs.n_pad / 64) * vs.ldxt;
    vs.owned = owned ? owned[v] != 0 : true;
    if (!vs.owned) h->all_owned = false;
    else h->last_owned = v;
    vs.row_map.resize(n_views);
    vs.col_map.resize(n_views);
  }
  auto bail = [&](hipError_t err, const char* what) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    resnmtf_destroy(h);
    return err == hipErrorOutOfMemory ? RESNMTF_ERR_ALLOC : RESNMTF_ERR_HIP;
  };
  if (o.stream) { h->stream = reinterpret_cast<hipStream_t>(o.stream); h->own_stream = false; }
  else {
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return bail(e, "hipStreamCreate");
    h->own_stream = true;
  }
  if ((e = dev_alloc_zero(&h->ctl, 1)) != hipSuccess) return bail(e, "hipMalloc ctl");
  h->err_cap = 1024;
  if ((e = dev_alloc_zero(&h->err, (size_t)h->err_cap * n_views)) != hipSuccess) return bail(e, "hipMalloc err");
  if ((e = alloc_host_mirrors(h, h->err_cap)) != hipSuccess) return bail(e, "hipHostMalloc error mirror");
  // replicate_f: one arena holds the F exchange block of every view, in view order (equal-shaped views
  // give equal strides, so that one in-place all-gather moves every rank's block -- sharded.py)
  auto fblk_usum_bytes = [](const ViewState& vs) { return ((size_t)vs.n_pad * vs.KP * sizeof(float) + 255) / 256 * 256; };
  auto fblk_size = [&](const ViewState& vs) {
    return (fblk_usum_bytes(vs) + (2 * (size_t)vs.k * vs.k + (size_t)vs.k) * sizeof(double) + 255) / 256 * 256;
  };
  // replicate_gs, equal-shaped F blocks: the S block of a view (k x k inputs of the S rule, written by the same pass launch
  // that produces U) is appended to its F block, so that ONE all-gather after the X.G pass moves both -- two collectives per
  // sweep between dependent steps instead of three
  size_t sblk_tail_bytes = 0;
  if (o.replicate_f && o.replicate_gs && !o.slice_chains) {      // (sliced chains: the S blocks travel on their own, beside the U slices)
    bool equal = true;
    for (const auto& vs : h->views) equal = equal && fblk_size(vs) == fblk_size(h->views[0]) && vs.k == h->views[0].k;
    if (equal) {
      const size_t kk0 = (size_t)h->views[0].k * h->views[0].k;
      h->sblk_stride = (5 * kk0 + 2 * (size_t)h->views[0].k + 1 + 31) / 32 * 32;
      sblk_tail_bytes = (h->sblk_stride * sizeof(double) + 255) / 256 * 256;
      h->sblk_embedded = true;
    }
  }
  if (o.replicate_f) {
    for (const auto& vs : h->views) h->fblk_arena_bytes += fblk_size(vs) + sblk_tail_bytes;
    if ((e = hipMalloc(&h->fblk_arena, h->fblk_arena_bytes)) != hipSuccess) return bail(e, "hipMalloc F exchange blocks");
    if ((e = hipMemset(h->fblk_arena, 0, h->fblk_arena_bytes)) != hipSuccess) return bail(e, "hipMemset F exchange blocks");
  }
  auto gblk_tsum_bytes = [](const ViewState& vs) { return ((size_t)vs.m_pad * vs.KP * sizeof(float) + 255) / 256 * 256; };
  auto gblk_size = [&](const ViewState& vs) {
    return (gblk_tsum_bytes(vs) + (2 * (size_t)vs.k * vs.k + (size_t)vs.k) * sizeof(double) + 255) / 256 * 256;
  };
  if (o.replicate_gs) {
    if (!o.replicate_f) { g_create_error = "replicate_gs needs replicate_f"; resnmtf_destroy(h); return RESNMTF_ERR_INVALID; }
    for (const auto& vs : h->views)
      if (vs.k != h->views[0].k) { g_create_error = "replicate_gs needs the same k in every view"; resnmtf_destroy(h); return RESNMTF_ERR_INVALID; }
    for (const auto& vs : h->views) h->gblk_arena_bytes += gblk_size(vs);
    if ((e = hipMalloc(&h->gblk_arena, h->gblk_arena_bytes)) != hipSuccess) return bail(e, "hipMalloc G exchange blocks");
    if ((e = hipMemset(h->gblk_arena, 0, h->gblk_arena_bytes)) != hipSuccess) return bail(e, "hipMemset G exchange blocks");
    if (!h->sblk_embedded) {
      const size_t kk0 = (size_t)h->views[0].k * h->views[0].k;
      h->sblk_stride = (5 * kk0 + 2 * (size_t)h->views[0].k + 1 + 31) / 32 * 32;
      if ((e = dev_alloc_zero(&h->sblk_arena, h->sblk_stride * n_views)) != hipSuccess) return bail(e, "hipMalloc S exchange blocks");
      h->sblk_base = h->sblk_arena; h->sblk_step = h->sblk_stride;
    }
    if ((e = dev_alloc_zero(&h->view_sweep, (size_t)n_views + 1)) != hipSuccess) return bail(e, "hipMalloc view_sweep");
  }
  if (o.slice_chains) {
    const ViewState& v0 = h->views[0];
    const int V = n_views;
    h->sliced = true;
    h->sl_rows = round_up(ceil_div(v0.n, V), 32); h->sl_cols = round_up(ceil_div(v0.m, V), 32);
    h->u_chunk = (size_t)h->sl_rows * v0.KP * sizeof(float);
    h->t_chunk = ((size_t)h->sl_cols * v0.KP * sizeof(float) + 2 * (size_t)v0.k * v0.k * sizeof(double) + 255) / 256 * 256;
    char** bufs[4] = {&h->u_send, &h->u_recv, &h->t_send, &h->t_recv};
    for (int b = 0; b < 4; ++b) {
      const size_t bytes = (b < 2 ? h->u_chunk : h->t_chunk) * V;
      if ((e = hipMalloc(reinterpret_cast<void**>(bufs[b]), bytes)) != hipSuccess) return bail(e, "hipMalloc slice exchange buffers");
      if ((e = hipMemset(*bufs[b], 0, bytes)) != hipSuccess) return bail(e, "hipMemset slice exchange buffers");
    }
    if ((e = dev_alloc_zero(&h->f_send, (size_t)V * h->sl_rows * v0.KP)) != hipSuccess) return bail(e, "hipMalloc slice exchange buffers");
    if ((e = dev_alloc_zero(&h->f_recv, (size_t)V * h->sl_rows * v0.KP)) != hipSuccess) return bail(e, "hipMalloc slice exchange buffers");
    if ((e = dev_alloc_zero(&h->slice_nd, (size_t)V * 2 * (size_t)round_up(std::max(h->sl_rows, h->sl_cols), 16) * v0.KP)) != hipSuccess)
      return bail(e, "hipMalloc slice product scratch");
    if ((e = dev_alloc_zero(&h->g_send, (size_t)V * h->sl_cols * v0.KP)) != hipSuccess) return bail(e, "hipMalloc slice exchange buffers");
    if ((e = dev_alloc_zero(&h->g_recv, (size_t)V * h->sl_cols * v0.KP)) != hipSuccess) return bail(e, "hipMalloc slice exchange buffers");
  }
  if (o.slice_p2p) {
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, o.device_id);
    if (!can) { g_create_error = "slice_p2p needs hipStreamWaitValue32 (hipDeviceAttributeCanUseStreamWaitValue)"; resnmtf_destroy(h); return RESNMTF_ERR_NO_DEVICE; }
    // the arrival counters are written by other devices' atomics and polled by this device's command processor: fine-grained
    // (coherent) memory where the runtime offers it
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&h->p2p_flags), 64 * sizeof(unsigned int), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      h->p2p_flags = nullptr;
      if ((e = hipMalloc(reinterpret_cast<void**>(&h->p2p_flags), 64 * sizeof(unsigned int))) != hipSuccess) return bail(e, "hipMalloc p2p flags");
    }
    if ((e = hipMemset(h->p2p_flags, 0, 64 * sizeof(unsigned int))) != hipSuccess) return bail(e, "hipMemset p2p flags");
    h->peers.resize((size_t)n_views);
    h->block_p2p = !o.slice_chains;
  }
  size_t fblk_off = 0, gblk_off = 0;
  for (int v = 0; v < n_views; ++v) {
    ViewState& vs = h->views[v];
    const size_t kk = (size_t)vs.k * vs.k;
    if ((e = dev_alloc_zero(&vs.F, (size_t)vs.n * vs.k)) != hipSuccess) return bail(e, "hipMalloc F");
    if ((e = dev_alloc_zero(&vs.G, (size_t)vs.m * vs.k)) != hipSuccess) return bail(e, "hipMalloc G");
    if ((e = dev_alloc_zero(&vs.S, kk)) != hipSuccess) return bail(e, "hipMalloc S");
    // ---- geometry (identical on every rank for a given view: n, m, k and the options decide it)
    // k x k mode (kernels.hip.inc, pass_kernel).  A (k <= 16): the update kernels emit fp64 partial
    // Grams (1-4 KB per workgroup), the job is workgroup 0 of the next pass launch and runs beside the
    // whole pass.  B (k > 16, where those partials would be 8-32 KB per workgroup): MFMA aux tiles in
    // the pass launch, tiny slab volume at any k, the job starts once the aux workgroups are done and
    // hides behind the (long) pass.  Measured on c2 ... 40000 x 2000 / 10000 x 8000 (tools/tune_c2.py):
    // A wins at k = 16 for every size tried; B wins at k = 32 / 64 (tools/bench_configs.py).
    const size_t xbytes = (size_t)vs.n * vs.m * sizeof(float);
    vs.kk_mode = (vs.KP == 16) ? 0 : 1;
    if (o.kk_mode == 1) vs.kk_mode = 0;
    if (o.kk_mode == 2 || o.slice_chains) vs.kk_mode = 1;      // (sliced chains: the Gram products come from the received f32 copy)
    // workgroup slots of a pass launch: target_workgroups overrides CUs x resident workgroups per CU
    const int nw_guess = (vs.NT <= 1 && (o.pass_waves == 4 || o.pass_waves == 8 || o.pass_waves == 16)) ? o.pass_waves : 8;
    const int aux_cap = (vs.NT >= 2 && o.bf16_split != 2) ? 16 : 64;
    size_aux(vs.m_pad, nw_guess, aux_cap, &vs.nsaux_xg, &vs.rpsaux_xg);
    size_aux(vs.n_pad, nw_guess, aux_cap, &vs.nsaux_xtf, &vs.rpsaux_xtf);
    const int slots_all = o.target_workgroups > 0 ? o.target_workgroups : h->n_cu * pass_blocks_per_cu(vs.NT, nw_guess);
    const int slots_xg = slots_all - (vs.kk_mode == 0 ? 1 : 3 * vs.nsaux_xg);
    const int slots_xtf = slots_all - (vs.kk_mode == 0 ? 1 : 2 * vs.nsaux_xtf);
    const bool fine = o.x_half == 0;
    size_pass(vs.NT, vs.n_pad / 64, vs.m_pad, slots_xg, max_pass_waves(vs.NT), o.pass_waves, o.pass_splits_xg, fine,
              &vs.nsplit_xg, &vs.rps_xg, &vs.nw_xg);
    size_pass(vs.NT, vs.m_pad / 64, vs.n_pad, slots_xtf, max_pass_waves(vs.NT), o.pass_waves, o.pass_splits_xtf, fine,
              &vs.nsplit_xtf, &vs.rps_xtf, &vs.nw_xtf);
    if (vs.NT >= 2 && o.bf16_split != 2) {      // k > 16: the wide bf16-piece form
      const WidePlan pg = plan_wide(vs.n_pad / 64, vs.m_pad, vs.KP, vs.kk_mode == 0 ? 0 : 3, slots_all, o.pass_splits_xg, nw_guess);
      const WidePlan pf = plan_wide(vs.m_pad / 64, vs.n_pad, vs.KP, vs.kk_mode == 0 ? 0 : 2, slots_all, o.pass_splits_xtf, nw_guess);
      vs.tw_xg = pg.tw; vs.nsplit_xg = pg.nsplit; vs.rps_xg = pg.rps;
      vs.tw_xtf = pf.tw; vs.nsplit_xtf = pf.nsplit; vs.rps_xtf = pf.rps;
      if (vs.kk_mode != 0) { vs.nsaux_xg = pg.nsaux; vs.rpsaux_xg = pg.rpsaux; vs.nsaux_xtf = pf.nsaux; vs.rpsaux_xtf = pf.rpsaux; }
    }
    // k <= 16: when the workgroups of a pass outnumber the slots (streamed geometry) each wave keeps the next
    // trip's loads in flight while it multiplies (two buffers of 4 steps instead of one of 8): X.G pass of a
    // 40000 x 2000 view 74 -> 67 us.  With everything resident from t = 0 (c2) the plain form is faster.
    vs.pp_xg = vs.NT == 1 && vs.nw_xg == 8 && (vs.n_pad / 64) * vs.nsplit_xg > slots_xg;
    vs.pp_xtf = vs.NT == 1 && vs.nw_xtf == 8 && (vs.m_pad / 64) * vs.nsplit_xtf > slots_xtf;
    const int RG = update_threads(vs.KP) / vs.KP;
    // update workgroups: mode A ~160 (few partials for the k x k job, two prefetched row groups each
    // at c2 -- tools/tune_c2.py); mode B (k > 16) ONE round of resident workgroups -- one 1024-thread workgroup per CU
    // at k > 32 (127 KB of LDS), two 512-thread ones at k = 32: every workgroup first copies the two k x k coefficient
    // matrices into LDS (64 KB at k = 64, 13 of the 61 us of a c5-sized F update when three rounds of workgroups each
    // did it: tools/ab_update_blocks.sh, F update 61 -> 43 us at c5, 13.5 -> 11.8 at c4; G 19 -> 14.5 at c5)
    const int nblk_round = h->n_cu * (update_threads(vs.KP) >= 1024 ? 1 : 2);
    const int nblk_target = o.update_blocks > 0 ? o.update_blocks : (vs.kk_mode == 0 ? (xbytes <= ((size_t)256 << 20) ? 160 : 512) : nblk_round);
    vs.rpbF = round_up(std::max(RG, ceil_div(vs.n, nblk_target)), RG); vs.nblkF = ceil_div(vs.n, vs.rpbF);
    vs.rpbG = round_up(std::max(RG, ceil_div(vs.m, nblk_target)), RG); vs.nblkG = ceil_div(vs.m, vs.rpbG);
    const size_t kkp = (size_t)vs.KP * vs.KP;
    // ---- the F update's inputs.  replicate_f: one contiguous exchange block per view, on every rank
    // (the sharded driver broadcasts it from the owner and runs the F update of coupled views everywhere)
    const size_t pxg_floats = (size_t)vs.nsplit_xg * vs.n_pad * vs.KP;
    if (o.replicate_f) {
      vs.fblk_bytes = fblk_size(vs) + sblk_tail_bytes;
      char* base = static_cast<char*>(h->fblk_arena) + fblk_off;
      if (h->sblk_embedded) {
        vs.sblk = reinterpret_cast<double*>(base + fblk_size(vs));
        if (v == 0) { h->sblk_base = vs.sblk; h->sblk_step = vs.fblk_bytes / sizeof(double); }
      }
      fblk_off += vs.fblk_bytes;
      vs.fblk = base;
      vs.Usum = reinterpret_cast<float*>(base);
      vs.Ma_F = reinterpret_cast<double*>(base + fblk_usum_bytes(vs));
      vs.Md_F = vs.Ma_F + kk;
      vs.lambda = vs.Md_F + kk;
      if (!vs.owned) {
        vs.f_replica = true;
        if ((e = dev_alloc_zero(&vs.F32, (size_t)vs.n_pad * 64)) != hipSuccess) return bail(e, "hipMalloc F32");
        if (vs.kk_mode == 0 && (e = dev_alloc_zero(&vs.partF, (size_t)vs.nblkF * (kkp + vs.KP))) != hipSuccess) return bail(e, "hipMalloc partF");
      }
    }
    if (o.replicate_gs) {
      vs.gblk_bytes = gblk_size(vs);
      char* base = static_cast<char*>(h->gblk_arena) + gblk_off;
      gblk_off += vs.gblk_bytes;
      vs.gblk = base;
      vs.Tsum = reinterpret_cast<float*>(base);
      vs.Ma_G = reinterpret_cast<double*>(base + gblk_tsum_bytes(vs));
      vs.Md_G = vs.Ma_G + kk;
      vs.mu = vs.Md_G + kk;
      if (!h->sblk_embedded) vs.sblk = h->sblk_arena + (size_t)v * h->sblk_stride;
      if (!vs.owned) {                // (a G replica writes the same copies as the owner: nobody reads them here)
        vs.g_replica = true;
        if ((e = dev_alloc_zero(&vs.G32, (size_t)vs.m_pad * 64)) != hipSuccess) return bail(e, "hipMalloc G32");
        if ((e = dev_alloc_zero(&vs.T32, (size_t)vs.m_pad * 64)) != hipSuccess) return bail(e, "hipMalloc T32");
        if (vs.kk_mode == 0 && (e = dev_alloc_zero(&vs.partG, (size_t)vs.nblkG * (2 * kkp + vs.KP))) != hipSuccess) return bail(e, "hipMalloc partG");
      }
    }
    if (!vs.owned) continue;
    if ((e = dev_alloc_zero(&vs.Pxg, pxg_floats)) != hipSuccess) return bail(e, "hipMalloc Pxg");
    if (!o.replicate_f) {
      if ((e = dev_alloc_zero(&vs.lambda, (size_t)vs.k)) != hipSuccess) return bail(e, "hipMalloc lambda");
      for (double** pp : {&vs.Ma_F, &vs.Md_F})
        if ((e = dev_alloc_zero(pp, kk)) != hipSuccess) return bail(e, "hipMalloc kxk");
    }
    if (!o.replicate_gs && (e = dev_alloc_zero(&vs.mu, (size_t)vs.k)) != hipSuccess) return bail(e, "hipMalloc mu");
    if ((e = dev_alloc_zero(&vs.xnorm2, 1)) != hipSuccess) return bail(e, "hipMalloc xnorm2");
    if ((e = dev_alloc_zero(&vs.X32, vs.x32_floats)) != hipSuccess) return bail(e, "hipMalloc X32");
    if ((e = dev_alloc_zero(&vs.Xt32, vs.xt32_floats)) != hipSuccess) return bail(e, "hipMalloc Xt32");
    vs.half = (o.x_half >= 1 && o.x_half <= 3) && vs.NT == 1 && vs.kk_mode == 0 && vs.nw_xg == 8 && vs.nw_xtf == 8;
    vs.u16 = vs.half && o.x_half >= 2;
    vs.half_capable = vs.half;
    if (vs.half) {      // one spare row group per tile keeps the tile starts off a common power-of-two stride
      vs.ld16x = ((size_t)vs.n_pad + 4) * 64; vs.ld16xt = ((size_t)vs.m_pad + 4) * 64;
      vs.x16_halves = (size_t)(vs.m_pad / 64) * vs.ld16x; vs.xt16_halves = (size_t)(vs.n_pad / 64) * vs.ld16xt;
      if ((e = dev_alloc_zero(&vs.X16, vs.x16_halves)) != hipSuccess) return bail(e, "hipMalloc X16");
      if ((e = dev_alloc_zero(&vs.Xt16, vs.xt16_halves)) != hipSuccess) return bail(e, "hipMalloc Xt16");
    }
    if ((e = dev_alloc_zero(&vs.F32, (size_t)vs.n_pad * 64)) != hipSuccess) return bail(e, "hipMalloc F32");
    if ((e = dev_alloc_zero(&vs.G32, (size_t)vs.m_pad * 64)) != hipSuccess) return bail(e, "hipMalloc G32");
    if ((e = dev_alloc_zero(&vs.T32, (size_t)vs.m_pad * 64)) != hipSuccess) return bail(e, "hipMalloc T32");
    if (vs.NT >= 2) {
      if ((e = dev_alloc_zero(&vs.Fk, (size_t)vs.n_pad * vs.KP * 3)) != hipSuccess) return bail(e, "hipMalloc Fk");
      if ((e = dev_alloc_zero(&vs.Gk, (size_t)vs.m_pad * vs.KP * 3)) != hipSuccess) return bail(e, "hipMalloc Gk");
    }
    if ((e = dev_alloc_zero(&vs.fuse_cnt, 4)) != hipSuccess) return bail(e, "hipMalloc cnt");
    if ((e = dev_alloc_zero(&vs.cnt_xg, 4)) != hipSuccess) return bail(e, "hipMalloc cnt");
    if ((e = dev_alloc_zero(&vs.cnt_xtf, 4)) != hipSuccess) return bail(e, "hipMalloc cnt");
    for (double** pp : {&vs.FtF, &vs.FtFS})
      if ((e = dev_alloc_zero(pp, kk)) != hipSuccess) return bail(e, "hipMalloc kxk");
    if (!o.replicate_gs)
      for (double** pp : {&vs.Ma_G, &vs.Md_G})
        if ((e = dev_alloc_zero(pp, kk)) != hipSuccess) return bail(e, "hipMalloc kxk");
    if ((e = dev_alloc_zero(&vs.cF, (size_t)vs.k)) != hipSuccess) return bail(e, "hipMalloc cF");
    if ((e = dev_alloc_zero(&vs.Pxtf, (size_t)vs.nsplit_xtf * vs.m_pad * vs.KP)) != hipSuccess) return bail(e, "hipMalloc Pxtf");
    if ((e = dev_alloc_zero(&vs.Paux_xg, (size_t)3 * vs.nsaux_xg * 64 * vs.KP)) != hipSuccess) return bail(e, "hipMalloc Paux");
    if ((e = dev_alloc_zero(&vs.Paux_xtf, (size_t)2 * vs.nsaux_xtf * 64 * vs.KP)) != hipSuccess) return bail(e, "hipMalloc Paux");
    if (vs.kk_mode == 0) {
      if ((e = dev_alloc_zero(&vs.partF, (size_t)vs.nblkF * (kkp + vs.KP))) != hipSuccess) return bail(e, "hipMalloc partF");
      if ((e = dev_alloc_zero(&vs.partG, (size_t)vs.nblkG * (2 * kkp + vs.KP))) != hipSuccess) return bail(e, "hipMalloc partG");
    }
  }
  if ((e = set_all_attrs()) != hipSuccess) return bail(e, "hipFuncSetAttribute");
  // the zero fills above ran on the NULL stream, which the handle's (non-blocking) stream does not wait
  // for: finish them before anything is enqueued there (a late memset would wipe uploaded data)
  if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(e, "hipDeviceSynchronize");
  if (o.time_kernels) {
    h->ev.resize(8192);
    h->ev_kind.resize(4096);
    for (auto& evt : h->ev)
      if ((e = hipEventCreate(&evt)) != hipSuccess) return bail(e, "hipEventCreate");
  }
  *out = h;
  return RESNMTF_OK;
}

int resnmtf_destroy(resnmtf_handle* h) {
  if (!h) return RESNMTF_OK;
  (void)hipSetDevice(h->opt.device_id);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  destroy_graphs(h);
  for (auto& v : h->views) free_view(v);
  if (h->fblk_arena) (void)hipFree(h->fblk_arena);
  if (h->gblk_arena) (void)hipFree(h->gblk_arena);
  if (h->sblk_arena) (void)hipFree(h->sblk_arena);
  for (auto& pc : h->peers)
    for (void* q : pc.opened)
      if (q) (void)hipIpcCloseMemHandle(q);
  for (void* p : {(void*)h->slice_nd, (void*)h->p2p_flags, (void*)h->view_sweep, (void*)h->u_send, (void*)h->u_recv, (void*)h->t_send, (void*)h->t_recv, (void*)h->f_send,
                  (void*)h->f_recv, (void*)h->g_send, (void*)h->g_recv})
    if (p) (void)hipFree(p);
  if (h->ctl) (void)hipFree(h->ctl);
  if (h->err) (void)hipFree(h->err);
  if (h->err_host) (void)hipHostFree(h->err_host);
  if (h->ctl_host) (void)hipHostFree(h->ctl_host);
  if (h->fuse_err) (void)hipHostFree(h->fuse_err);
  for (auto& evt : h->ev)
    if (evt) (void)hipEventDestroy(evt);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RESNMTF_OK;
}

namespace {
// raw = false: x is already non-negative and column-normalised.  raw = true: make_non_neg_inner +
// matrix_normalisation (R/utils.r:20-27, 86-88) run on the device, fused into the conversion.
// Source of the staging image: the host matrix x, or (x == NULL) a pseudo-random permutation of the
// entries of another view's device copy (shuffle_src, see resnmtf_shuffle_view).
// fp16 images of an uploaded view: per-view power-of-two scale that puts the largest entry near 2^14
// relative quantisation error of X below which the guarded mode (x_half = 3) lets the passes use the 16-bit image:
// F / G move by 0.2 ... 2 x that error (tools/quant_study.py), the bar is 1e-4
constexpr double kHalfGuard = 3.0e-5;
int build_half_images(resnmtf_handle* h, ViewState& vs) {
  double* scratch = nullptr;          // [0] = max entry bits (as unsigned), [1] = sum (x~ - x)^2, [2] = copy of ||X||^2
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&scratch), 3 * sizeof(double)));
  hipError_t e = hipMemsetAsync(scratch, 0, 3 * sizeof(double), h->stream);
  unsigned int bits = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(max_entry_kernel, dim3(1024), dim3(256), 0, h->stream, vs.X32, vs.x32_floats,
                       reinterpret_cast<unsigned int*>(scratch));
    e = hipMemcpyAsync(&bits, scratch, sizeof(bits), hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { (void)hipFree(scratch); return h->fail_hip("2-byte images (max)", e); }
  float mx;
  std::memcpy(&mx, &bits, sizeof(mx));
  int ex = 0;
  vs.xscale = 1.f;
  if (mx > 0.f && std::isfinite(mx)) {
    (void)std::frexp(mx, &ex);
    // fp16: power of two, max * scale in [2^13, 2^14) (exact scaling); integers: the full range, max -> 65535 (the
    // widening is exact whatever the step, the step is taken out once per output in f32)
    vs.xscale = vs.u16 ? 65535.f / mx : std::ldexp(1.f, 14 - ex);
  }
  hipLaunchKernelGGL(pack_half_kernel, dim3((unsigned)(((size_t)(vs.n_pad / 4) * 64 * (vs.m_pad / 64) + 255) / 256)), dim3(256), 0,
                     h->stream, vs.X32, vs.ldx, vs.n_pad, vs.m_pad / 64, vs.xscale, vs.X16, vs.ld16x, vs.u16 ? 1 : 0, scratch + 1);
  hipLaunchKernelGGL(pack_half_kernel, dim3((unsigned)(((size_t)(vs.m_pad / 4) * 64 * (vs.n_pad / 64) + 255) / 256)), dim3(256), 0,
                     h->stream, vs.Xt32, vs.ldxt, vs.m_pad, vs.n_pad / 64, vs.xscale, vs.Xt16, vs.ld16xt, vs.u16 ? 1 : 0,
                     (double*)nullptr);
  double host[2] = {0.0, 0.0};
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&host[0], scratch + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&host[1], vs.xnorm2, sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(scratch);
  if (e != hipSuccess) return h->fail_hip("2-byte images (pack)", e);
  vs.x_relerr = host[1] > 0.0 ? std::sqrt(host[0] / host[1]) : 0.0;
  const bool use = h->opt.x_half == 3 ? vs.x_relerr <= kHalfGuard : true;
  if (use != vs.half) {               // the factor operand copies follow the image's layout: rewrite them
    vs.half = use;
    if (vs.has_factors) {
      HIP_TRY(h, hipMemsetAsync(vs.F32, 0, (size_t)vs.n_pad * 64 * sizeof(float), h->stream));
      HIP_TRY(h, hipMemsetAsync(vs.G32, 0, (size_t)vs.m_pad * 64 * sizeof(float), h->stream));
      hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(vs.n * vs.k, 256)), dim3(256), 0, h->stream, vs.F, vs.n,
                         vs.k, vs.F32, vs.kk_mode == 0 ? vs.KP : 64, vs.NT, vs.half ? 1 : 0, vs.Fk);
      hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(vs.m * vs.k, 256)), dim3(256), 0, h->stream, vs.G, vs.m,
                         vs.k, vs.G32, vs.kk_mode == 0 ? vs.KP : 64, vs.NT, vs.half ? 1 : 0, vs.Gk);
      HIP_TRY(h, hipGetLastError());
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  h->prepared = false;
  h->resume_ok = false;
  return RESNMTF_OK;
}
struct ShuffleSrc { const float* X32; size_t ldx; unsigned long long seed; const int* rows; const int* cols; };   // rows != NULL: sub-sample
int upload_view(resnmtf_handle* h, int v, const double* x, bool raw, int* was_negative, const ShuffleSrc* shuffle_src = nullptr) {
  if (int rc = check_view(h, v)) return rc;
  if (!x && !shuffle_src) return h->fail(RESNMTF_ERR_INVALID, "x is NULL");
  ViewState& vs = h->views[v];
  if (!vs.owned) return h->fail(RESNMTF_ERR_STATE, "set_view on a view this handle does not own");
  h->resume_ok = false;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  const size_t count = (size_t)vs.n * vs.m;
  double* staging = nullptr;
  double* partial = nullptr;
  double* colstat = nullptr;      // [2][m]: shift, colsum; then one int flag
  const dim3 grid(ceil_div(vs.n, 32), ceil_div(vs.m, 32));
  const int nparts = grid.x * grid.y;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&staging), count * sizeof(double)));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&partial), (size_t)nparts * sizeof(double));
  if (e == hipSuccess && raw) e = hipMalloc(reinterpret_cast<void**>(&colstat), ((size_t)2 * vs.m + 1) * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(staging); (void)hipFree(partial); return h->fail_hip("hipMalloc upload buffers", e); }
  double* shift = raw ? colstat : nullptr;
  double* colsum = raw ? colstat + vs.m : nullptr;
  int* neg = raw ? reinterpret_cast<int*>(colstat + 2 * (size_t)vs.m) : nullptr;
  int neg_host = 0;
  if (x) e = hipMemcpyAsync(staging, x, count * sizeof(double), hipMemcpyHostToDevice, h->stream);
  else {
    if (shuffle_src->rows)
      hipLaunchKernelGGL(subsample_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, shuffle_src->X32,
                         shuffle_src->ldx, shuffle_src->rows, vs.n, shuffle_src->cols, vs.m, staging);
    else
      hipLaunchKernelGGL(shuffle_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, shuffle_src->X32,
                         shuffle_src->ldx, vs.n, vs.m, shuffle_src->seed, staging);
    e = hipGetLastError();
  }
  unsigned char* line_mask = nullptr;      // device-drawn data: which rows / columns came out all zero
  std::vector<int> line_counts(2, 0);
  vs.empty_rows = vs.empty_cols = 0; vs.empty_mask.clear();
  if (e == hipSuccess && !x) {
    e = hipMalloc(reinterpret_cast<void**>(&line_mask), (size_t)vs.n + vs.m + 2 * sizeof(int) + 8);
    if (e == hipSuccess) {
      int* counts = reinterpret_cast<int*>(line_mask + (((size_t)vs.n + vs.m + 7) / 8) * 8);
      e = hipMemsetAsync(counts, 0, 2 * sizeof(int), h->stream);
      hipLaunchKernelGGL(empty_lines_kernel, dim3(ceil_div(vs.n + vs.m, 256)), dim3(256), 0, h->stream, staging, vs.n, vs.m, line_mask, counts);
      vs.empty_mask.resize((size_t)vs.n + vs.m);
      if (e == hipSuccess) e = hipMemcpyAsync(vs.empty_mask.data(), line_mask, (size_t)vs.n + vs.m, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(line_counts.data(), counts, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    }
  }
  if (e == hipSuccess) e = hipMemsetAsync(vs.X32, 0, vs.x32_floats * sizeof(float), h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(vs.Xt32, 0, vs.xt32_floats * sizeof(float), h->stream);
  if (e == hipSuccess && raw) e = hipMemsetAsync(neg, 0, sizeof(double), h->stream);
  if (e == hipSuccess) {
    if (raw) hipLaunchKernelGGL(column_stats_kernel, dim3(vs.m), dim3(256), 0, h->stream, staging, vs.n, vs.m, shift, colsum, neg);
    hipLaunchKernelGGL(convert_x_kernel, grid, dim3(256), 0, h->stream, staging, vs.n, vs.m, vs.X32, vs.ldx,
                       vs.Xt32, vs.ldxt, partial, shift, colsum);
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, h->stream, partial, nparts, vs.xnorm2);
    e = hipGetLastError();
  }
  if (e == hipSuccess && raw) e = hipMemcpyAsync(&neg_host, neg, sizeof(int), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(staging);
  (void)hipFree(partial);
  (void)hipFree(colstat);
  (void)hipFree(line_mask);
  if (e != hipSuccess) return h->fail_hip("set_view", e);
  vs.empty_rows = line_counts[0]; vs.empty_cols = line_counts[1];
  if (was_negative) *was_negative = neg_host;
  vs.has_x = true;
  if (vs.half_capable) return build_half_images(h, vs);
  return RESNMTF_OK;
}
}  // namespace

int resnmtf_set_view(resnmtf_handle* h, int v, const double* x) { return upload_view(h, v, x, false, nullptr); }
int resnmtf_set_view_raw(resnmtf_handle* h, int v, const double* x_raw, int* was_negative) {
  return upload_view(h, v, x_raw, true, was_negative);
}

// ---- view data without a host round trip (the k sweep re-uses one upload, the shuffles of
// spurious-bicluster removal are drawn on the device)
static int check_view_pair(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src) {
  if (int rc = check_view(dst, v)) return rc;
  if (!src || v_src < 0 || v_src >= src->V) return dst->fail(RESNMTF_ERR_INVALID, "bad source handle / view");
  const ViewState& a = dst->views[v];
  const ViewState& b = src->views[v_src];
  if (!a.owned || !b.owned || !b.has_x) return dst->fail(RESNMTF_ERR_STATE, "both views must be owned and the source uploaded");
  if (a.n != b.n || a.m != b.m) return dst->fail(RESNMTF_ERR_INVALID, "views differ in shape");
  if (dst->opt.device_id != src->opt.device_id) return dst->fail(RESNMTF_ERR_INVALID, "handles live on different devices");
  return RESNMTF_OK;
}

int resnmtf_copy_view(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src) {
  if (int rc = check_view_pair(dst, v, src, v_src)) return rc;
  ViewState& a = dst->views[v];
  const ViewState& b = src->views[v_src];
  dst->resume_ok = false;
  HIP_TRY(dst, hipSetDevice(dst->opt.device_id));
  HIP_TRY(dst, hipStreamSynchronize(src->stream));
  if (int rc = sync_both(dst)) return rc;
  // same shape and options decide the same pitches; guard anyway
  if (a.ldx != b.ldx || a.ldxt != b.ldxt) return dst->fail(RESNMTF_ERR_INVALID, "views differ in device layout (no_pitch_pad)");
  HIP_TRY(dst, hipMemcpyAsync(a.X32, b.X32, a.x32_floats * sizeof(float), hipMemcpyDeviceToDevice, dst->stream));
  HIP_TRY(dst, hipMemcpyAsync(a.Xt32, b.Xt32, a.xt32_floats * sizeof(float), hipMemcpyDeviceToDevice, dst->stream));
  HIP_TRY(dst, hipMemcpyAsync(a.xnorm2, b.xnorm2, sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
  HIP_TRY(dst, hipStreamSynchronize(dst->stream));
  a.has_x = true;
  if (a.half_capable) return build_half_images(dst, a);
  return RESNMTF_OK;
}

int resnmtf_shuffle_view(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src, unsigned long long seed, int normalise) {
  if (int rc = check_view_pair(dst, v, src, v_src)) return rc;
  const ViewState& b = src->views[v_src];
  HIP_TRY(dst, hipSetDevice(dst->opt.device_id));
  HIP_TRY(dst, hipStreamSynchronize(src->stream));
  const ShuffleSrc sh{b.X32, b.ldx, seed, nullptr, nullptr};
  return upload_view(dst, v, nullptr, normalise != 0, nullptr, &sh);
}

int resnmtf_subsample_view(resnmtf_handle* dst, int v, resnmtf_handle* src, int v_src, const int* rows, const int* cols) {
  if (int rc = check_view(dst, v)) return rc;
  if (!src || v_src < 0 || v_src >= src->V) return dst->fail(RESNMTF_ERR_INVALID, "bad source handle / view");
  if (!rows || !cols) return dst->fail(RESNMTF_ERR_INVALID, "rows / cols are NULL");
  const ViewState& a = dst->views[v];
  const ViewState& b = src->views[v_src];
  if (!a.owned || !b.owned || !b.has_x) return dst->fail(RESNMTF_ERR_STATE, "both views must be owned and the source uploaded");
  if (dst->opt.device_id != src->opt.device_id) return dst->fail(RESNMTF_ERR_INVALID, "handles live on different devices");
  for (int r = 0; r < a.n; ++r) if (rows[r] < 0 || rows[r] >= b.n) return dst->fail(RESNMTF_ERR_INVALID, "row index out of range");
  for (int c = 0; c < a.m; ++c) if (cols[c] < 0 || cols[c] >= b.m) return dst->fail(RESNMTF_ERR_INVALID, "column index out of range");
  HIP_TRY(dst, hipSetDevice(dst->opt.device_id));
  HIP_TRY(dst, hipStreamSynchronize(src->stream));
  int* idx = nullptr;
  HIP_TRY(dst, hipMalloc(reinterpret_cast<void**>(&idx), ((size_t)a.n + a.m) * sizeof(int)));
  hipError_t e = hipMemcpy(idx, rows, (size_t)a.n * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(idx + a.n, cols, (size_t)a.m * sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(idx); return dst->fail_hip("subsample_view", e); }
  const ShuffleSrc sh{b.X32, b.ldx, 0ull, idx, idx + a.n};
  const int rc = upload_view(dst, v, nullptr, false, nullptr, &sh);      // sub-samples are NOT re-normalised (Appendix B11)
  (void)hipFree(idx);
  return rc;
}

int resnmtf_view_empty_lines(resnmtf_handle* h, int v, int* n_empty_rows, int* n_empty_cols, unsigned char* row_mask,
                             unsigned char* col_mask) {
  if (int rc = check_view(h, v)) return rc;
  const ViewState& vs = h->views[v];
  if (n_empty_rows) *n_empty_rows = vs.empty_rows;
  if (n_empty_cols) *n_empty_cols = vs.empty_cols;
  const bool have = vs.empty_mask.size() == (size_t)vs.n + vs.m;
  if (row_mask) for (int r = 0; r < vs.n; ++r) row_mask[r] = have ? vs.empty_mask[(size_t)r] : 0;
  if (col_mask) for (int c = 0; c < vs.m; ++c) col_mask[c] = have ? vs.empty_mask[(size_t)vs.n + c] : 0;
  return RESNMTF_OK;
}

int resnmtf_get_view(resnmtf_handle* h, int v, double* x) {
  if (int rc = check_view(h, v)) return rc;
  if (!x) return h->fail(RESNMTF_ERR_INVALID, "x is NULL");
  const ViewState& vs = h->views[v];
  if (!vs.owned || !vs.has_x) return h->fail(RESNMTF_ERR_STATE, "no data on this handle for the view");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  std::vector<float> t(vs.xt32_floats);                  // Xt32(c, r) = X[r][c], tile-major over r
  HIP_TRY(h, hipMemcpy(t.data(), vs.Xt32, t.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int c = 0; c < vs.m; ++c)
    for (int r = 0; r < vs.n; ++r) x[(size_t)c * vs.n + r] = (double)t[xidx(c, r, vs.ldxt)];
  return RESNMTF_OK;
}

int resnmtf_set_factors(resnmtf_handle* h, int v, const double* F, const double* S, const double* G,
                        const double* lambda, const double* mu) {
  if (int rc = check_view(h, v)) return rc;
  if (!F || !S || !G) return h->fail(RESNMTF_ERR_INVALID, "F, S and G are required");
  ViewState& vs = h->views[v];
  h->resume_ok = false;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  std::vector<double> f, s, g;
  to_row_major(F, vs.n, vs.k, f);
  to_row_major(S, vs.k, vs.k, s);
  to_row_major(G, vs.m, vs.k, g);
  HIP_TRY(h, hipMemcpyAsync(vs.F, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(vs.S, s.data(), s.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(vs.G, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  std::vector<double> lam(vs.k, 0.0), muv(vs.k, 0.0);
  if (vs.owned) {
    // explicit-init branch: lambda = colSums(F), mu = colSums(G) (R/update_steps.r:55-56)
    for (int j = 0; j < vs.k; ++j) {
      if (lambda) lam[j] = lambda[j];
      else { double t = 0.0; for (int i = 0; i < vs.n; ++i) t += F[(size_t)j * vs.n + i]; lam[j] = t; }
      if (mu) muv[j] = mu[j];
      else { double t = 0.0; for (int i = 0; i < vs.m; ++i) t += G[(size_t)j * vs.m + i]; muv[j] = t; }
    }
    HIP_TRY(h, hipMemcpyAsync(vs.lambda, lam.data(), lam.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(vs.mu, muv.data(), muv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(vs.F32, 0, (size_t)vs.n_pad * 64 * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(vs.G32, 0, (size_t)vs.m_pad * 64 * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(vs.T32, 0, (size_t)vs.m_pad * 64 * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(vs.cnt_xg, 0, 4 * sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(vs.cnt_xtf, 0, 4 * sizeof(int), h->stream));
    hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(vs.n * vs.k, 256)), dim3(256), 0, h->stream, vs.F, vs.n,
                       vs.k, vs.F32, vs.kk_mode == 0 ? vs.KP : 64, vs.NT, vs.half ? 1 : 0, vs.Fk);
    hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(vs.m * vs.k, 256)), dim3(256), 0, h->stream, vs.G, vs.m,
                       vs.k, vs.G32, vs.kk_mode == 0 ? vs.KP : 64, vs.NT, vs.half ? 1 : 0, vs.Gk);
    HIP_TRY(h, hipGetLastError());
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // host vectors go out of scope
  vs.has_factors = true;
  return RESNMTF_OK;
}

// ---------------------------------------------------------------------------------------------
// resnmtf_init_svd -- init_mats_inner (R/update_steps.r:78-125) for one view, SURVEY 8(f1).
// Randomized subspace iteration (L = 16 ceil((k + 8) / 16) <= 64 columns): the two big products per
// iteration are the streaming-pass kernels; CholeskyQR2 orthonormalisation and the final L x L
// symmetric eigenproblem are fp64, their L x L factorisations on the host.
// ---------------------------------------------------------------------------------------------
namespace {
// C (L x L, row-major, symmetric positive definite) = R^T R, R upper triangular; returns false if not PD
bool cholesky_upper(const std::vector<double>& C, int L, std::vector<double>& R) {
  R.assign((size_t)L * L, 0.0);
  for (int j = 0; j < L; ++j) {
    double d = C[(size_t)j * L + j];
    for (int t = 0; t < j; ++t) d -= R[(size_t)t * L + j] * R[(size_t)t * L + j];
    if (!(d > 0.0)) return false;
    const double rjj = std::sqrt(d);
    R[(size_t)j * L + j] = rjj;
    for (int c = j + 1; c < L; ++c) {
      double v = C[(size_t)j * L + c];
      for (int t = 0; t < j; ++t) v -= R[(size_t)t * L + j] * R[(size_t)t * L + c];
      R[(size_t)j * L + c] = v / rjj;
    }
  }
  return true;
}
void invert_upper(const std::vector<double>& R, int L, std::vector<double>& Ri) {
  Ri.assign((size_t)L * L, 0.0);
  for (int j = 0; j < L; ++j) {
    Ri[(size_t)j * L + j] = 1.0 / R[(size_t)j * L + j];
    for (int i = j - 1; i >= 0; --i) {
      double v = 0.0;
      for (int t = i + 1; t <= j; ++t) v += R[(size_t)i * L + t] * Ri[(size_t)t * L + j];
      Ri[(size_t)i * L + j] = -v / R[(size_t)i * L + i];
    }
  }
}
// cyclic Jacobi for a symmetric L x L matrix: A -> eigenvalues (diagonal), V columns = eigenvectors
void jacobi_eigen(std::vector<double>& A, int L, std::vector<double>& V) {
  V.assign((size_t)L * L, 0.0);
  for (int i = 0; i < L; ++i) V[(size_t)i * L + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < L; ++i)
      for (int j = 0; j < L; ++j) (i == j ? diag : off) += A[(size_t)i * L + j] * A[(size_t)i * L + j];
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < L - 1; ++p)
      for (int q = p + 1; q < L; ++q) {
        const double apq = A[(size_t)p * L + q];
        if (apq == 0.0) continue;
        const double theta = (A[(size_t)q * L + q] - A[(size_t)p * L + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int r = 0; r < L; ++r) {           // columns p, q
          const double arp = A[(size_t)r * L + p], arq = A[(size_t)r * L + q];
          A[(size_t)r * L + p] = c * arp - sn * arq;
          A[(size_t)r * L + q] = sn * arp + c * arq;
        }
        for (int r = 0; r < L; ++r) {           // rows p, q
          const double apr = A[(size_t)p * L + r], aqr = A[(size_t)q * L + r];
          A[(size_t)p * L + r] = c * apr - sn * aqr;
          A[(size_t)q * L + r] = sn * apr + c * aqr;
        }
        for (int r = 0; r < L; ++r) {
          const double vrp = V[(size_t)r * L + p], vrq = V[(size_t)r * L + q];
          V[(size_t)r * L + p] = c * vrp - sn * vrq;
          V[(size_t)r * L + q] = sn * vrp + c * vrq;
        }
      }
  }
}

struct InitScratch {
  double *Yn = nullptr, *Yn2 = nullptr, *Zm = nullptr, *Zm2 = nullptr, *gpart = nullptr, *gram = nullptr, *M = nullptr;
  float *Pn = nullptr, *Pm = nullptr;
  bool own_pn = false, own_pm = false;
  ~InitScratch() {
    for (double* p : {Yn, Yn2, Zm, Zm2, gpart, gram, M}) if (p) (void)hipFree(p);
    if (own_pn && Pn) (void)hipFree(Pn);
    if (own_pm && Pm) (void)hipFree(Pm);
  }
};
constexpr int kGramBlocks = 256;

// gram = Y^T Y (L x L) on the host
int ts_gram_host(resnmtf_handle* h, InitScratch& sc, const double* Y, int len, int L, std::vector<double>& C) {
  const int rpb = round_up(ceil_div(len, kGramBlocks), 16), nblk = ceil_div(len, rpb);
  hipLaunchKernelGGL(ts_gram_kernel, dim3(nblk), dim3(256), 0, h->stream, Y, len, L, rpb, sc.gpart);
  hipLaunchKernelGGL(reduce_records_kernel, dim3(ceil_div(L * L, 256)), dim3(256), 0, h->stream, sc.gpart, nblk, L * L, sc.gram);
  C.resize((size_t)L * L);
  HIP_TRY(h, hipMemcpyAsync(C.data(), sc.gram, C.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return RESNMTF_OK;
}
int ts_apply(resnmtf_handle* h, InitScratch& sc, const double* Y, int len, int L, const std::vector<double>& M,
             double* Q, float* W32, int nt) {
  HIP_TRY(h, hipMemcpyAsync(sc.M, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(ts_apply_kernel, dim3(ceil_div(len * L, 256)), dim3(256), 0, h->stream, Y, len, L, sc.M, Q, W32, 64, nt);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));       // M (host vector) may go out of scope
  return RESNMTF_OK;
}
// CholeskyQR2: Y (in `a`) -> orthonormal columns (back in `a`, `b` is scratch) + f32 operand copy
int orthonormalise(resnmtf_handle* h, InitScratch& sc, double* a, double* b, int len, int L, float* W32, int nt) {
  std::vector<double> C, R, Ri;
  for (int round = 0; round < 2; ++round) {
    double* src = round == 0 ? a : b;
    double* dst = round == 0 ? b : a;
    if (int rc = ts_gram_host(h, sc, src, len, L, C)) return rc;
    if (round == 0) {            // tiny ridge: random sketches of rank-deficient data stay factorable
      double tr = 0.0;
      for (int i = 0; i < L; ++i) tr += C[(size_t)i * L + i];
      for (int i = 0; i < L; ++i) C[(size_t)i * L + i] += 1e-14 * tr;
    }
    if (!cholesky_upper(C, L, R)) return h->fail(RESNMTF_ERR_INVALID, "init_svd: sketch is rank deficient");
    invert_upper(R, L, Ri);
    if (int rc = ts_apply(h, sc, src, len, L, Ri, dst, round == 1 ? W32 : nullptr, nt)) return rc;
  }
  return RESNMTF_OK;
}
}  // namespace

namespace {
// R/update_steps.r:93-115 on the k leading triplets: U (n x ldu), V (m x ldv) row-major, d descending.
// Column-major outputs go through resnmtf_set_factors.
int finish_init(resnmtf_handle* h, int v, const std::vector<double>& U, int ldu, const std::vector<double>& V, int ldv,
                const std::vector<double>& d, double sigma, std::mt19937_64& gen, double* singular_values) {
  const ViewState& vs = h->views[v];
  const int n = vs.n, m = vs.m, k = vs.k;
  std::vector<double> F0((size_t)n * k), G0((size_t)m * k), S0((size_t)k * k, 0.0), cf(k, 0.0), cg(k, 0.0), lam(k, 0.0), muv(k, 0.0);
  for (int j = 0; j < k; ++j) {
    for (int i = 0; i < n; ++i) { const double a = std::fabs(U[(size_t)i * ldu + j]); F0[(size_t)j * n + i] = a; cf[j] += a; }   // :93,:100
    for (int i = 0; i < m; ++i) { const double a = std::fabs(V[(size_t)i * ldv + j]); G0[(size_t)j * m + i] = a; cg[j] += a; }   // :94,:101
  }
  std::normal_distribution<double> noise(0.0, std::sqrt(sigma));                    // mvrnorm(k, 0, sigma I), :96-99
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i) {
      double sv = (i == j ? std::fabs(d[j]) : 0.0);                                 // :95
      if (sigma > 0.0) sv += std::fabs(noise(gen));
      S0[(size_t)j * k + i] = sv * cf[j] * cg[j];                                   // :102-105 (column sweep)
    }
  for (int j = 0; j < k; ++j) {
    for (int i = 0; i < n; ++i) { F0[(size_t)j * n + i] /= cf[j]; lam[j] += F0[(size_t)j * n + i]; }   // :106-109,:114
    for (int i = 0; i < m; ++i) { G0[(size_t)j * m + i] /= cg[j]; muv[j] += G0[(size_t)j * m + i]; }   // :110-113,:115
  }
  if (singular_values)
    for (int j = 0; j < k; ++j) singular_values[j] = d[j];
  return resnmtf_set_factors(h, v, F0.data(), S0.data(), G0.data(), lam.data(), muv.data());
}

// Thin views (min(n, m) smaller than the sketch): the exact SVD through the Gram matrix of the short
// side -- Y = X (m <= n) or X^T, C = Y^T Y (r x r, fp64, device), Jacobi on the host, the long-side
// vectors Y W Sigma^-1 on the device.  No random sketch, no iteration.
int init_svd_thin(resnmtf_handle* h, int v, double sigma, std::mt19937_64& gen, double* singular_values) {
  ViewState& vs = h->views[v];
  const int n = vs.n, m = vs.m;
  const bool tall = m <= n;                       // Y = X [n][m]  or  X^T [m][n]
  const int len = tall ? n : m, r = tall ? m : n;
  InitScratch sc;
  hipError_t e;
  auto alloc = [&](double** p, size_t cnt) { return hipMalloc(reinterpret_cast<void**>(p), cnt * sizeof(double)); };
  if ((e = alloc(&sc.Yn, (size_t)len * r)) != hipSuccess || (e = alloc(&sc.Yn2, (size_t)len * r)) != hipSuccess ||
      (e = alloc(&sc.gpart, (size_t)(kGramBlocks + 1) * r * r)) != hipSuccess || (e = alloc(&sc.gram, (size_t)r * r)) != hipSuccess ||
      (e = alloc(&sc.M, (size_t)r * r)) != hipSuccess)
    return h->fail_hip("init_svd hipMalloc", e);
  hipLaunchKernelGGL(widen_rows_kernel, dim3(ceil_div(len * r, 256)), dim3(256), 0, h->stream, tall ? vs.X32 : vs.Xt32,
                     tall ? vs.ldx : vs.ldxt, len, r, sc.Yn);
  HIP_TRY(h, hipGetLastError());
  std::vector<double> C, W;
  if (int rc = ts_gram_host(h, sc, sc.Yn, len, r, C)) return rc;
  jacobi_eigen(C, r, W);
  std::vector<int> order(r);
  for (int i = 0; i < r; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return C[(size_t)a * r + a] > C[(size_t)b * r + b]; });
  std::vector<double> d(r), Ws((size_t)r * r), Wn((size_t)r * r);
  for (int j = 0; j < r; ++j) {
    const int src = order[j];
    d[j] = std::sqrt(std::max(C[(size_t)src * r + src], 0.0));
    for (int i = 0; i < r; ++i) {
      Ws[(size_t)i * r + j] = W[(size_t)i * r + src];
      Wn[(size_t)i * r + j] = d[j] > 0.0 ? W[(size_t)i * r + src] / d[j] : 0.0;
    }
  }
  if (int rc = ts_apply(h, sc, sc.Yn, len, r, Wn, sc.Yn2, nullptr, 1)) return rc;     // long-side vectors
  std::vector<double> Lg((size_t)len * r);
  HIP_TRY(h, hipMemcpyAsync(Lg.data(), sc.Yn2, Lg.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return tall ? finish_init(h, v, Lg, r, Ws, r, d, sigma, gen, singular_values)
              : finish_init(h, v, Ws, r, Lg, r, d, sigma, gen, singular_values);
}
}  // namespace

int resnmtf_init_svd(resnmtf_handle* h, int v, unsigned long long seed, double sigma, int n_power,
                     double* singular_values) {
  if (int rc = check_view(h, v)) return rc;
  ViewState& vs = h->views[v];
  if (!vs.owned || !vs.has_x) return h->fail(RESNMTF_ERR_STATE, "init_svd needs an owned view with data (set_view first)");
  if (n_power < 1) n_power = 3;
  if (!(sigma >= 0.0)) return h->fail(RESNMTF_ERR_INVALID, "sigma must be >= 0");
  h->resume_ok = false;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  const int n = vs.n, m = vs.m, k = vs.k;
  const int L = std::min(64, 16 * ceil_div(k + 8, 16)), NTi = L / 16;
  std::mt19937_64 gen(seed);
  if (std::min(n, m) < L) return init_svd_thin(h, v, sigma, gen, singular_values);
  InitScratch sc;
  hipError_t e;
  auto alloc = [&](double** p, size_t cnt) { return hipMalloc(reinterpret_cast<void**>(p), cnt * sizeof(double)); };
  if ((e = alloc(&sc.Yn, (size_t)n * L)) != hipSuccess || (e = alloc(&sc.Yn2, (size_t)n * L)) != hipSuccess ||
      (e = alloc(&sc.Zm, (size_t)m * L)) != hipSuccess || (e = alloc(&sc.Zm2, (size_t)m * L)) != hipSuccess ||
      (e = alloc(&sc.gpart, (size_t)(kGramBlocks + 1) * L * L)) != hipSuccess || (e = alloc(&sc.gram, (size_t)L * L)) != hipSuccess ||
      (e = alloc(&sc.M, (size_t)L * L)) != hipSuccess)
    return h->fail_hip("init_svd hipMalloc", e);
  // slabs: the view's own when the sketch is as wide as its KP, else temporaries
  if (L == vs.KP) { sc.Pn = vs.Pxg; sc.Pm = vs.Pxtf; }
  else {
    if ((e = hipMalloc(reinterpret_cast<void**>(&sc.Pn), (size_t)vs.nsplit_xg * vs.n_pad * L * sizeof(float))) != hipSuccess)
      return h->fail_hip("init_svd hipMalloc slabs", e);
    sc.own_pn = true;
    if ((e = hipMalloc(reinterpret_cast<void**>(&sc.Pm), (size_t)vs.nsplit_xtf * vs.m_pad * L * sizeof(float))) != hipSuccess)
      return h->fail_hip("init_svd hipMalloc slabs", e);
    sc.own_pm = true;
  }
  PassArgs xg{}, xt{};
  xg.A = vs.Xt32; xg.lda = 64; xg.tile_stride = vs.ldxt; xg.ntiles = vs.n_pad / 64; xg.B = vs.G32; xg.ldb = 64; xg.P = sc.Pn;
  xg.cols_pad = vs.n_pad; xg.rows_pad = vs.m_pad; xg.rows_per_split = vs.rps_xg; xg.nsplit = vs.nsplit_xg; xg.ctl = h->ctl;
  xt.A = vs.X32; xt.lda = 64; xt.tile_stride = vs.ldx; xt.ntiles = vs.m_pad / 64; xt.B = vs.F32; xt.ldb = 64; xt.P = sc.Pm;
  xt.cols_pad = vs.m_pad; xt.rows_pad = vs.n_pad; xt.rows_per_split = vs.rps_xtf; xt.nsplit = vs.nsplit_xtf; xt.ctl = h->ctl;

  // Omega: m x L standard normal (host generator: the reference's RNG is not reproducible anyway)
  std::normal_distribution<double> normal(0.0, 1.0);
  std::vector<double> omega((size_t)m * L);
  for (double& x : omega) x = normal(gen);
  HIP_TRY(h, hipMemcpyAsync(sc.Zm, omega.data(), omega.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(vs.F32, 0, (size_t)vs.n_pad * 64 * sizeof(float), h->stream));
  HIP_TRY(h, hipMemsetAsync(vs.G32, 0, (size_t)vs.m_pad * 64 * sizeof(float), h->stream));
  hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(m * L, 256)), dim3(256), 0, h->stream, sc.Zm, m, L, vs.G32, 64, NTi, 0, (unsigned short*)nullptr);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int it = 0; it < n_power; ++it) {
    launch_pass_plain(h, xg, NTi, true);                                                   // Y = X Z
    hipLaunchKernelGGL(slab_sum_kernel, dim3(ceil_div(n * L, 256)), dim3(256), 0, h->stream, sc.Pn, vs.nsplit_xg, vs.n_pad, L, n, sc.Yn);
    HIP_TRY(h, hipGetLastError());
    if (int rc = orthonormalise(h, sc, sc.Yn, sc.Yn2, n, L, vs.F32, NTi)) return rc;          // Q (in Yn) + F32
    launch_pass_plain(h, xt, NTi, false);                                                  // Z = X^T Q
    hipLaunchKernelGGL(slab_sum_kernel, dim3(ceil_div(m * L, 256)), dim3(256), 0, h->stream, sc.Pm, vs.nsplit_xtf, vs.m_pad, L, m, sc.Zm);
    HIP_TRY(h, hipGetLastError());
    if (it + 1 < n_power)
      if (int rc = orthonormalise(h, sc, sc.Zm, sc.Zm2, m, L, vs.G32, NTi)) return rc;
  }
  // Z = X^T Q = V Sigma Ut^T  ->  Z^T Z = Ut Sigma^2 Ut^T;  U = Q Ut,  V = Z Ut Sigma^-1
  std::vector<double> C, Ut;
  if (int rc = ts_gram_host(h, sc, sc.Zm, m, L, C)) return rc;
  jacobi_eigen(C, L, Ut);
  std::vector<int> order(L);
  for (int i = 0; i < L; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return C[(size_t)a * L + a] > C[(size_t)b * L + b]; });
  std::vector<double> d(L), Mu((size_t)L * L), Mv((size_t)L * L);
  for (int j = 0; j < L; ++j) {
    const int src = order[j];
    d[j] = std::sqrt(std::max(C[(size_t)src * L + src], 0.0));
    for (int i = 0; i < L; ++i) {
      Mu[(size_t)i * L + j] = Ut[(size_t)i * L + src];
      Mv[(size_t)i * L + j] = d[j] > 0.0 ? Ut[(size_t)i * L + src] / d[j] : 0.0;
    }
  }
  if (int rc = ts_apply(h, sc, sc.Yn, n, L, Mu, sc.Yn2, nullptr, NTi)) return rc;             // U (n x L)
  if (int rc = ts_apply(h, sc, sc.Zm, m, L, Mv, sc.Zm2, nullptr, NTi)) return rc;             // V (m x L)
  std::vector<double> U((size_t)n * L), V((size_t)m * L);
  HIP_TRY(h, hipMemcpyAsync(U.data(), sc.Yn2, U.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(V.data(), sc.Zm2, V.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return finish_init(h, v, U, L, V, L, d, sigma, gen, singular_values);
}

int resnmtf_set_restrictions(resnmtf_handle* h, const double* phi, const double* xi, const double* psi) {
  if (!h) return RESNMTF_ERR_INVALID;
  const size_t cnt = (size_t)h->V * h->V;
  const double* src[3] = {phi, xi, psi};
  std::vector<double>* dst[3] = {&h->phi, &h->xi, &h->psi};
  for (int t = 0; t < 3; ++t) {
    if (src[t]) {
      for (size_t e = 0; e < cnt; ++e)
        if (!(src[t][e] >= 0.0)) return h->fail(RESNMTF_ERR_INVALID, "restriction matrices must be non-negative (R/utils.r:343-355)");
      dst[t]->assign(src[t], src[t] + cnt);
    } else {
      dst[t]->assign(cnt, 0.0);
    }
  }
  h->prepared = false;
  h->resume_ok = false;
  return RESNMTF_OK;
}

static int set_shared(resnmtf_handle* h, int v, int w, int count, const int* idx_v, const int* idx_w, bool rows) {
  if (int rc = check_view(h, v)) return rc;
  if (int rc = check_view(h, w)) return rc;
  if (v == w) return h->fail(RESNMTF_ERR_INVALID, "shared map needs two different views");
  ViewState& vs = h->views[v];
  const ViewState& ws = h->views[w];
  SharedMap& mp = rows ? vs.row_map[w] : vs.col_map[w];
  const int len_v = rows ? vs.n : vs.m, len_w = rows ? ws.n : ws.m;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  h->prepared = false;
  h->resume_ok = false;
  if (count < 0) {           // NA
    mp.set = true; mp.count = -1;
    if (mp.dev) { (void)hipFree(mp.dev); mp.dev = nullptr; }
    return RESNMTF_OK;
  }
  if (count > 0 && (!idx_v || !idx_w)) return h->fail(RESNMTF_ERR_INVALID, "index arrays are NULL");
  std::vector<int> map((size_t)len_v, -1);
  for (int t = 0; t < count; ++t) {
    if (idx_v[t] < 0 || idx_v[t] >= len_v || idx_w[t] < 0 || idx_w[t] >= len_w)
      return h->fail(RESNMTF_ERR_INVALID, "shared index out of range");
    map[idx_v[t]] = idx_w[t];
  }
  if (!mp.dev) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&mp.dev), (size_t)len_v * sizeof(int)));
  HIP_TRY(h, hipMemcpy(mp.dev, map.data(), (size_t)len_v * sizeof(int), hipMemcpyHostToDevice));
  mp.set = true; mp.count = count;
  mp.identity = true;
  for (int r = 0; r < len_v && mp.identity; ++r) mp.identity = map[r] == r;
  return RESNMTF_OK;
}

int resnmtf_set_shared_rows(resnmtf_handle* h, int v, int w, int count, const int* idx_v, const int* idx_w) {
  return set_shared(h, v, w, count, idx_v, idx_w, true);
}
int resnmtf_set_shared_cols(resnmtf_handle* h, int v, int w, int count, const int* idx_v, const int* idx_w) {
  return set_shared(h, v, w, count, idx_v, idx_w, false);
}

// RESNMTF_PHASE_F_ALL in one launch (f_chain_kernel): every view holds the inputs of its F update here
// (owned, or an F replica), k <= 16 in hand-off mode A, equal row counts and row blocking, every coupling
// through identity row maps, at most four owned views, at most 8 views.  Otherwise one launch per view.
static void build_chain(resnmtf_handle* h) {
  h->chain_views = 0;
  const int V = h->V;
  if (V < 2 || V > 8 || h->opt.no_f_chain) return;
  const ViewState& v0 = h->views[0];
  int n_owned = 0, kpack = -1;
  for (int v = 0; v < V; ++v) {
    const ViewState& vs = h->views[v];
    if (!vs.owned && !vs.f_replica) return;
    if (vs.owned) {                                         // one operand-copy layout for all owned views
      if (kpack < 0) kpack = vs.half ? 1 : 0;
      else if (kpack != (vs.half ? 1 : 0)) return;
    }
    if (vs.KP != 16 || vs.kk_mode != 0 || vs.n != v0.n || vs.k != v0.k || vs.rpbF != v0.rpbF || vs.nblkF != v0.nblkF) return;
    if (vs.argF.nsplit > 4 || vs.argF.cols_pad != v0.argF.cols_pad) return;     // raw split slabs the kernel keeps per view
    if (vs.owned) ++n_owned;
    for (int c = 0; c < vs.argF.n_couple; ++c)
      if (vs.argF.couple[c].map) return;                    // permuted shared rows: rows of different workgroups
  }
  if (n_owned > 4) return;
  ChainArgs<8>& a = h->chain;
  a = ChainArgs<8>{};
  a.len = v0.n; a.k = v0.k; a.n_views = V; a.rows_per_block = v0.rpbF;
  a.ctl = h->ctl;
  a.pstride = (size_t)v0.argF.cols_pad * 16;
  a.kpack32 = kpack > 0 ? 1 : 0;
  for (int v = 0; v < 8; ++v) a.emit_slot[v] = -1;
  for (int v = 0; v < V; ++v) {
    const ViewState& vs = h->views[v];
    const UpdateArgs& f = vs.argF;
    if (vs.owned) {
      a.emit_slot[v] = a.n_emit;
      a.W32e[a.n_emit] = vs.F32; a.parte[a.n_emit] = vs.partF;
      ++a.n_emit;
    }
    a.W[v] = vs.F; a.U[v] = f.P; a.nsplit[v] = f.nsplit; a.Ma[v] = vs.Ma_F; a.Md[v] = vs.Md_F; a.lm[v] = vs.lambda;
    a.sigma[v] = f.sigma;
    a.n_other[v] = (double)vs.n;
    if (f.restricted) a.restricted |= 1u << v;
    for (int c = 0; c < f.n_couple; ++c)
      for (int w = 0; w < V; ++w)
        if (f.couple[c].W == h->views[w].F) { a.cmask[v] |= 1u << w; a.weight[v][w] = f.couple[c].weight; }
  }
  h->chain_views = V;
  h->chain_blocks = v0.nblkF;
}

// RESNMTF_PHASE_G_ALL in one launch at k <= 16 (f_chain_kernel, G form; replicated G chain, R/update_steps.r:195-204): every view
// holds the inputs of its G update here (owned, or a G replica) as ONE folded slab, hand-off mode A, equal column counts and
// blocking, every coupling through identity column maps, at most two owned views, at most 8 views.  Else one launch per view.
static void build_g_chain(resnmtf_handle* h) {
  h->gchain_views = 0;
  const int V = h->V;
  if (V < 2 || V > 8 || h->opt.no_f_chain || !h->opt.replicate_gs || h->sliced) return;
  const ViewState& v0 = h->views[0];
  int n_owned = 0;
  for (int v = 0; v < V; ++v) {
    const ViewState& vs = h->views[v];
    if (!vs.owned && !vs.g_replica) return;
    if (vs.KP != 16 || vs.kk_mode != 0 || vs.half || vs.m != v0.m || vs.k != v0.k || vs.rpbG != v0.rpbG || vs.nblkG != v0.nblkG) return;
    if (vs.argG.nsplit != 1 || vs.argG.cols_pad != v0.argG.cols_pad) return;
    if (vs.owned && ++n_owned > 2) return;
    for (int c = 0; c < vs.argG.n_couple; ++c)
      if (vs.argG.couple[c].map) return;
  }
  ChainArgs<8>& a = h->gchain;
  a = ChainArgs<8>{};
  a.len = v0.m; a.k = v0.k; a.n_views = V; a.rows_per_block = v0.rpbG;
  a.ctl = h->ctl;
  a.pstride = (size_t)v0.argG.cols_pad * 16;
  for (int v = 0; v < 8; ++v) a.emit_slot[v] = -1;
  for (int v = 0; v < V; ++v) {
    const ViewState& vs = h->views[v];
    const UpdateArgs& g = vs.argG;
    if (vs.owned) {
      a.emit_slot[v] = a.n_emit;
      a.W32e[a.n_emit] = vs.G32; a.parte[a.n_emit] = vs.partG; a.T32e[a.n_emit] = vs.T32;
      ++a.n_emit;
    }
    a.W[v] = vs.G; a.U[v] = g.P; a.nsplit[v] = 1; a.Ma[v] = vs.Ma_G; a.Md[v] = vs.Md_G; a.lm[v] = vs.mu;
    a.sigma[v] = g.sigma;
    a.n_other[v] = (double)vs.m;
    if (g.restricted) a.restricted |= 1u << v;
    for (int c = 0; c < g.n_couple; ++c)
      for (int w = 0; w < V; ++w)
        if (g.couple[c].W == h->views[w].G) { a.cmask[v] |= 1u << w; a.weight[v][w] = g.couple[c].weight; }
  }
  h->gchain_views = V;
  h->gchain_blocks = v0.nblkG;
}

// RESNMTF_PHASE_F_ALL / G_ALL in one launch at k = 32 / 64 (wide_chain_kernel): every view holds the inputs of the update
// here (owned or replica) as ONE folded slab (the exchange blocks of replicate_f / replicate_gs), equal lengths and k,
// every coupling through identity maps, at most one owned view, at most 8 views.  Otherwise one launch per view.
static void build_wide_chain(resnmtf_handle* h) {
  h->wchain_ok[0] = h->wchain_ok[1] = false;
  const int V = h->V;
  if (V < 2 || V > 8 || h->opt.no_f_chain) return;
  const ViewState& v0 = h->views[0];
  if (v0.KP < 32 || h->sliced) return;
  for (int g = 0; g < 2; ++g) {
    WideChainArgs<8>& a = h->wchain[g];
    a = WideChainArgs<8>{};
    a.own = -1;
    bool ok = true;
    int n_owned = 0;
    for (int v = 0; v < V && ok; ++v) {
      const ViewState& vs = h->views[v];
      const UpdateArgs& u = g == 0 ? vs.argF : vs.argG;
      if (!vs.owned && !(g == 0 ? vs.f_replica : vs.g_replica)) { ok = false; break; }
      if (vs.KP != v0.KP || vs.k != v0.k || (g == 0 ? vs.n != v0.n : vs.m != v0.m)) { ok = false; break; }
      if (u.nsplit != 1 || u.P == nullptr) { ok = false; break; }            // the folded slab of an exchange block
      for (int c = 0; c < u.n_couple; ++c)
        if (u.couple[c].map) ok = false;                                     // permuted shared rows: rows of other workgroups
      if (vs.owned) {
        if (++n_owned > 1) { ok = false; break; }
        a.own = v; a.W32 = g == 0 ? vs.F32 : vs.G32; a.Wk = g == 0 ? vs.Fk : vs.Gk; a.T32 = g == 0 ? nullptr : vs.T32; a.ld32 = u.ld32;
        if (!a.W32 || !a.Wk || (g == 1 && !a.T32)) { ok = false; break; }
      }
      a.W[v] = u.W; a.U[v] = u.P; a.Ma[v] = u.Ma; a.Md[v] = u.Md; a.lm[v] = u.lm;
      a.sigma[v] = u.sigma;
      a.n_other[v] = (double)(g == 0 ? vs.n : vs.m);
      if (u.restricted) a.restricted |= 1u << v;
      for (int c = 0; c < u.n_couple; ++c)
        for (int w = 0; w < V; ++w)
          if (u.couple[c].W == (g == 0 ? h->views[w].F : h->views[w].G)) { a.cmask[v] |= 1u << w; a.weight[v][w] = u.couple[c].weight; }
    }
    if (!ok) continue;
    a.len = g == 0 ? v0.n : v0.m; a.n_self = a.len; a.k = v0.k; a.n_views = V; a.ngroups = ceil_div(a.len, 32);
    a.ctl = h->ctl;
    // persistent workgroups: one per CU at k > 32 (157 KB of LDS at k = 64, 131 KB at 48), two at k = 32
    h->wchain_grid[g] = std::min(a.ngroups, h->n_cu * (v0.KP == 32 ? 2 : 1));
    h->wchain_ok[g] = true;
  }
}

// slice_chains: RESNMTF_PHASE_SLICE_F / _G -- the chain of every view on this rank's row (column) slice, one launch
// (wide_chain_kernel on the slice: inputs = the received rows of every view's product, outputs = the fp64 rows kept here and
// their f32 copies for the owners)
static int build_slice_chain(resnmtf_handle* h) {
  const int V = h->V, r = h->opt.slice_index;
  const ViewState& v0 = h->views[0];
  for (int g = 0; g < 2; ++g) {
    WideChainArgs<8>& a = h->schain[g];
    a = WideChainArgs<8>{};
    a.own = -1;
    const int per = g == 0 ? h->sl_rows : h->sl_cols, full = g == 0 ? v0.n : v0.m;
    const int begin = std::min(r * per, full), len = std::min(per, full - begin);
    for (int v = 0; v < V; ++v) {
      const ViewState& vs = h->views[v];
      const UpdateArgs& u = g == 0 ? vs.argF : vs.argG;
      for (int c = 0; c < u.n_couple; ++c)
        if (u.couple[c].map)
          return h->fail(RESNMTF_ERR_INVALID, "slice_chains: coupled views must share all their rows / columns in the same order (identity maps)");
      a.W[v] = (g == 0 ? vs.F : vs.G) + (size_t)begin * vs.k;
      if (g == 0) {
        a.U[v] = reinterpret_cast<const float*>(h->u_recv + (size_t)v * h->u_chunk);
        a.Ma[v] = vs.Ma_F; a.Md[v] = vs.Md_F; a.lm[v] = vs.lambda;
      } else {
        const char* chunk = h->t_recv + (size_t)v * h->t_chunk;
        a.U[v] = reinterpret_cast<const float*>(chunk);
        a.Ma[v] = reinterpret_cast<const double*>(chunk + (size_t)per * vs.KP * sizeof(float)); a.Md[v] = a.Ma[v] + (size_t)vs.k * vs.k;
        a.lm[v] = vs.mu;
      }
      a.sigma[v] = u.sigma;
      a.n_other[v] = (double)full;
      if (u.restricted) a.restricted |= 1u << v;
      for (int c = 0; c < u.n_couple; ++c)
        for (int w = 0; w < V; ++w)
          if (u.couple[c].W == (g == 0 ? h->views[w].F : h->views[w].G)) { a.cmask[v] |= 1u << w; a.weight[v][w] = u.couple[c].weight; }
    }
    a.O32 = g == 0 ? h->f_send : h->g_send; a.o32_stride = (unsigned)((size_t)per * v0.KP);
    if (h->opt.slice_p2p) {      // straight into the owner's receive slot for this rank's slice
      if (!h->p2p_ready) return h->fail(RESNMTF_ERR_STATE, "slice_p2p: import every rank's buffers first (resnmtf_p2p_import)");
      a.O32 = nullptr;
      for (int v = 0; v < V; ++v) a.O32v[v] = (g == 0 ? h->peers[(size_t)v].f_recv : h->peers[(size_t)v].g_recv) + (size_t)r * per * v0.KP;
    }
    a.len = len; a.n_self = full; a.k = v0.k; a.n_views = V; a.ngroups = ceil_div(std::max(len, 0), 32);
    a.ctl = h->ctl;
    h->schain_grid[g] = std::min(a.ngroups, h->n_cu * (v0.KP <= 32 ? 2 : 1));
  }
  return RESNMTF_OK;
}

// builds the kernel argument blocks (coupling tables included) from the host-side description
static int build_args(resnmtf_handle* h) {
  const int V = h->V;
  double sum_psi = 0.0, sum_xi = 0.0;
  for (double x : h->psi) sum_psi += x;
  for (double x : h->xi) sum_xi += x;
  for (int v = 0; v < V; ++v) {
    ViewState& vs = h->views[v];
    if (!vs.has_factors) return h->fail(RESNMTF_ERR_STATE, "set_factors missing for a view");
    if (!vs.owned && !vs.f_replica && !vs.g_replica) continue;
    // --- F update (R/update_steps.r:141-165)
    UpdateArgs& f = vs.argF;
    f = UpdateArgs{};
    f.len = vs.n; f.k = vs.k; f.W = vs.F; f.W32 = vs.F32; f.Wk = vs.Fk; f.ld32 = vs.kk_mode == 0 ? vs.KP : 64; f.kpack32 = vs.half ? 1 : 0;
    f.P = vs.Pxg; f.nsplit = vs.nsplit_xg; f.cols_pad = vs.n_pad;
    if (vs.Usum) { f.P = vs.Usum; f.nsplit = 1; }      // replicate_f: the folded slab of the exchange block
    if (!vs.owned && vs.NT >= 2) { f.W32 = nullptr; f.Wk = nullptr; }   // only this view's passes (on its owner) read them
    f.Ma = vs.Ma_F; f.Md = vs.Md_F; f.lm = vs.lambda; f.T32 = nullptr; f.part = vs.partF;
    f.rows_per_block = vs.rpbF; f.ctl = h->ctl;
    {
      double sigma = 0.0;
      for (int i = 0; i < V; ++i) sigma += h->phi[(size_t)i + (size_t)v * V];     // sum(phi[, v])  (:150,:152)
      f.restricted = (sigma != 0.0) ? 1 : 0;
      f.sigma = sigma;
      f.n_couple = 0;
      for (int i = 0; i < V && f.restricted; ++i) {
        const double wgt = h->phi[(size_t)i + (size_t)v * V];
        if (wgt == 0.0 || i == v) continue;                                        // utils.r:66
        const SharedMap& mp = vs.row_map[i];
        if (!mp.set || mp.count < 0) continue;                                     // NA: utils.r:70
        if (h->views[i].k != vs.k) return h->fail(RESNMTF_ERR_INVALID, "phi-coupled views need equal k");
        CoupleDesc& c = f.couple[f.n_couple++];
        c.W = h->views[i].F; c.map = mp.identity ? nullptr : mp.dev; c.weight = wgt; c.n_other = (double)h->views[i].n;
      }
    }
    // --- G update (R/update_steps.r:180-207); branch on the WHOLE psi matrix (:190)
    if (vs.owned || vs.g_replica) {
    UpdateArgs& g = vs.argG;
    g = UpdateArgs{};
    g.len = vs.m; g.k = vs.k; g.W = vs.G; g.W32 = vs.G32; g.Wk = vs.Gk; g.ld32 = vs.kk_mode == 0 ? vs.KP : 64; g.kpack32 = vs.half ? 1 : 0;
    g.P = vs.Pxtf; g.nsplit = vs.nsplit_xtf; g.cols_pad = vs.m_pad;
    if (vs.Tsum) { g.P = vs.Tsum; g.nsplit = 1; }      // replicate_gs: the folded slab of the exchange block
    g.Ma = vs.Ma_G; g.Md = vs.Md_G; g.lm = vs.mu; g.T32 = vs.T32; g.part = vs.partG;
    if (!vs.owned && vs.NT >= 2) { g.W32 = nullptr; g.Wk = nullptr; g.T32 = nullptr; }
    g.rows_per_block = vs.rpbG; g.ctl = h->ctl;
    {
      double sigma = 0.0;
      for (int i = 0; i < V; ++i) sigma += h->psi[(size_t)i + (size_t)v * V];     // sum(psi[, v])  (:195,:200)
      g.restricted = (sum_psi != 0.0) ? 1 : 0;
      g.sigma = sigma;
      g.n_couple = 0;
      for (int i = 0; i < V && g.restricted; ++i) {
        const double wgt = h->psi[(size_t)i + (size_t)v * V];
        if (wgt == 0.0 || i == v) continue;
        const SharedMap& mp = vs.col_map[i];
        if (!mp.set || mp.count < 0) continue;
        if (h->views[i].k != vs.k) return h->fail(RESNMTF_ERR_INVALID, "psi-coupled views need equal k");
        CoupleDesc& c = g.couple[g.n_couple++];
        c.W = h->views[i].G; c.map = mp.identity ? nullptr : mp.dev; c.weight = wgt; c.n_other = (double)h->views[i].m;
      }
    }
    }
    if (!vs.owned) continue;       // (replicas only ever run the update kernels)
    if (!vs.has_x) return h->fail(RESNMTF_ERR_STATE, "set_view missing for an owned view");
    // --- streaming passes
    PassArgs& xg = vs.passXG;
    xg = PassArgs{};
    xg.A = vs.Xt32; xg.lda = 64; xg.tile_stride = vs.ldxt; xg.ntiles = vs.n_pad / 64; xg.B = vs.G32; xg.Bk = vs.Gk; xg.ldb = vs.kk_mode == 0 ? vs.KP : 64; xg.P = vs.Pxg;
    xg.cols_pad = vs.n_pad; xg.rows_pad = vs.m_pad; xg.rows_per_split = vs.rps_xg; xg.nsplit = vs.nsplit_xg;
    xg.tw = vs.tw_xg; xg.ntg = ceil_div(xg.ntiles, xg.tw);
    xg.A16 = vs.Xt16; xg.tile_stride16 = vs.ld16xt;
    xg.aux[0] = vs.G32; xg.aux[1] = vs.T32; xg.aux[2] = nullptr; xg.naux = 3;     // G^T G, T^T G, colSums(G)
    xg.Paux = vs.Paux_xg; xg.rows_per_split_aux = vs.rpsaux_xg; xg.nsplit_aux = vs.nsaux_xg; xg.aux_cnt = vs.cnt_xg;
    xg.ctl = h->ctl;
    PassArgs& xt = vs.passXtF;
    xt = PassArgs{};
    xt.A = vs.X32; xt.lda = 64; xt.tile_stride = vs.ldx; xt.ntiles = vs.m_pad / 64; xt.B = vs.F32; xt.Bk = vs.Fk; xt.ldb = vs.kk_mode == 0 ? vs.KP : 64; xt.P = vs.Pxtf;
    xt.cols_pad = vs.m_pad; xt.rows_pad = vs.n_pad; xt.rows_per_split = vs.rps_xtf; xt.nsplit = vs.nsplit_xtf;
    xt.tw = vs.tw_xtf; xt.ntg = ceil_div(xt.ntiles, xt.tw);
    xt.A16 = vs.X16; xt.tile_stride16 = vs.ld16x;
    xt.aux[0] = vs.F32; xt.aux[1] = nullptr; xt.naux = 2;                          // F^T F, colSums(F)
    xt.Paux = vs.Paux_xtf; xt.rows_per_split_aux = vs.rpsaux_xtf; xt.nsplit_aux = vs.nsaux_xtf; xt.aux_cnt = vs.cnt_xtf;
    xt.ctl = h->ctl;
    // --- k x k side kernels
    KKFArgs& kf = vs.argKF;
    kf = KKFArgs{};
    kf.k = vs.k; kf.S = vs.S; kf.part = vs.partF; kf.nblk = vs.nblkF;
    kf.FtF = vs.FtF; kf.FtFS = vs.FtFS; kf.Ma_G = vs.Ma_G; kf.Md_G = vs.Md_G; kf.cF = vs.cF;
    KKSArgs& ks = vs.argKS;
    ks = KKSArgs{};
    ks.k = vs.k; ks.mode = 1; ks.part = vs.partG; ks.nblk = vs.nblkG;
    ks.FtF = vs.FtF; ks.FtFS = vs.FtFS; ks.cF = vs.cF;
    ks.S = vs.S; ks.lambda = vs.lambda; ks.mu = vs.mu; ks.Ma_F = vs.Ma_F; ks.Md_F = vs.Md_F;
    ks.xnorm2 = vs.xnorm2;
    ks.err = h->err; ks.err_stride = V; ks.err_col = v; ks.err_cap = h->err_cap;
    ks.err_host = h->err_host_dev; ks.ctl_host = h->ctl_host_dev;
    ks.sblock = vs.sblk;
    ks.ctl = h->ctl; ks.last_view = (v == h->last_owned) ? 1 : 0; ks.n_views = V; ks.tol = -1.0;
    {
      double sigma = 0.0;
      for (int i = 0; i < V; ++i) sigma += h->xi[(size_t)i + (size_t)v * V];      // sum(xi[, v])  (:231,:233)
      ks.restricted = (sum_xi != 0.0) ? 1 : 0;                                     // whole matrix (:226)
      ks.sigma = sigma;
      ks.n_couple = 0;
      for (int i = 0; i < V && ks.restricted; ++i) {
        const double wgt = h->xi[(size_t)i + (size_t)v * V];
        if (wgt == 0.0 || i == v) continue;                                        // utils.r:42
        if (h->views[i].k != vs.k) return h->fail(RESNMTF_ERR_INVALID, "xi-coupled views need equal k");
        SCouple& c = ks.couple[ks.n_couple++];
        c.S = h->views[i].S;      // running list: updated in place, in view order, on the side stream
        c.weight = wgt;
      }
    }
  }
  build_chain(h);
  build_g_chain(h);
  build_wide_chain(h);
  if (h->sliced)
    if (int rc = build_slice_chain(h)) return rc;
  return RESNMTF_OK;
}

int resnmtf_reserve_sweeps(resnmtf_handle* h, int sweeps) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (sweeps < 1) return h->fail(RESNMTF_ERR_INVALID, "sweeps must be positive");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  return ensure_err_capacity(h, sweeps);
}

// allow_resume: resnmtf_run directly after a completed resnmtf_run with nothing set in between -- the device state
// (X.G slabs, F coefficients, Gram partials) is bit for bit what the run prologue would recompute, so only the loop
// control is reset, in stream order, without a host synchronisation
static int prepare_impl(resnmtf_handle* h, bool allow_resume, bool keep_ctl = false) {
  if (!h->prepared) {
    if (!h->ladder.empty() || !h->exact.empty()) HIP_TRY(h, hipStreamSynchronize(h->stream));   // (a graph may still be executing:
    destroy_graphs(h);                                                                           //  resnmtf_run returns on a counter)
    if (int rc = build_args(h)) return rc;
    h->prepared = true;
  }
  const bool resume = allow_resume && h->resume_ok;
  if (resume && keep_ctl && h->ctl_clean && h->next_base < (1 << 30)) {      // not even a memset
    h->sweep_base = h->next_base;
    return RESNMTF_OK;
  }
  h->sweep_base = h->next_base = 0;
  std::memset(h->ctl_host, 0, sizeof(SweepCtl));
  HIP_TRY(h, hipMemsetAsync(h->ctl, 0, sizeof(SweepCtl), h->stream));      // all-zero bytes = SweepCtl{}
  if (h->view_sweep) HIP_TRY(h, hipMemsetAsync(h->view_sweep, 0, ((size_t)h->V + 1) * sizeof(int), h->stream));
  if (resume) return RESNMTF_OK;
  // run prologue: F coefficients and the first X.G pass of every owned view
  h->resume_ok = false;
  for (const auto& v : h->views)
    if (v.owned && v.fuse_cnt) HIP_TRY(h, hipMemsetAsync(v.fuse_cnt, 0, 4 * sizeof(int), h->stream));
  for (const auto& v : h->views)
    if (v.owned) enqueue_prologue(h, v);
  HIP_TRY(h, hipGetLastError());
  return RESNMTF_OK;
}

int resnmtf_prepare(resnmtf_handle* h) {
  if (!h) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  if (h->opt.slice_p2p) {      // the arrival counters only count up: one prepare per handle (the driver prepares once)
    if (!h->p2p_ready) return h->fail(RESNMTF_ERR_STATE, "slice_p2p: import every rank's buffers first (resnmtf_p2p_import)");
    if (h->p2p_prepared) return h->fail(RESNMTF_ERR_STATE, "slice_p2p: a handle is prepared once (its arrival counters are cumulative)");
    h->p2p_prepared = true;
  }
  return prepare_impl(h, false);
}

int resnmtf_phase(resnmtf_handle* h, int v, int phase, int sweep) {
  if (int rc = check_view(h, v)) return rc;
  if (!h->prepared) return h->fail(RESNMTF_ERR_STATE, "resnmtf_prepare has not been called");
  const ViewState& vs = h->views[v];
  if (!vs.owned && !(vs.f_replica && phase == RESNMTF_PHASE_F) && phase != RESNMTF_PHASE_F_ALL && phase != RESNMTF_PHASE_LOCAL_SWEEP &&
      phase != RESNMTF_PHASE_G_ALL && phase != RESNMTF_PHASE_S_ALL)
    return h->fail(RESNMTF_ERR_STATE, "phase on a view this handle does not own");
  if (phase >= RESNMTF_PHASE_XTF && phase <= RESNMTF_PHASE_S_ALL && !h->opt.replicate_gs)
    return h->fail(RESNMTF_ERR_STATE, "this phase needs a handle created with replicate_gs = 1");
  if (phase >= RESNMTF_PHASE_SLICE_F && phase <= RESNMTF_PHASE_SLICE_XG && !h->sliced)
    return h->fail(RESNMTF_ERR_STATE, "this phase needs a handle created with slice_chains = 1");
  if (h->sliced && phase != RESNMTF_PHASE_S_ALL && !(phase >= RESNMTF_PHASE_SLICE_F && phase <= RESNMTF_PHASE_SLICE_XG))
    return h->fail(RESNMTF_ERR_STATE, "slice_chains: use PHASE_SLICE_F / SLICE_XTF / SLICE_G / SLICE_XG / S_ALL");
  if ((phase == RESNMTF_PHASE_SLICE_XTF || phase == RESNMTF_PHASE_SLICE_XG) && !vs.owned)
    return h->fail(RESNMTF_ERR_STATE, "phase on a view this handle does not own");
  if (h->block_p2p && !h->opt.replicate_gs && phase != RESNMTF_PHASE_LOCAL_SWEEP)
    return h->fail(RESNMTF_ERR_STATE, "slice_p2p with replicate_f alone: the sweep is RESNMTF_PHASE_LOCAL_SWEEP (one exchange per sweep)");
  if (h->opt.replicate_gs && (phase == RESNMTF_PHASE_G || phase == RESNMTF_PHASE_LOCAL_SWEEP))
    return h->fail(RESNMTF_ERR_STATE, "replicate_gs: use PHASE_XTF / G_ALL / XG / S_ALL instead of PHASE_G");
  if (sweep < 0) return h->fail(RESNMTF_ERR_INVALID, "negative sweep index");
  if (sweep >= h->err_cap) return h->fail(RESNMTF_ERR_STATE, "sweep beyond the reserved error buffer (resnmtf_reserve_sweeps)");
  h->resume_ok = false;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (h->opt.time_kernels && h->ev_used + 64 > h->ev.size())
    if (int rc = flush_timing(h)) return rc;
  // convergence mode of the phase API (resnmtf_set_stop_tolerance): the replicated S chain runs the stop test, every kernel
  // of a checked phase leaves at once when the flag is up
  const double tol = h->opt.replicate_gs ? h->phase_tol : -1.0;
  const bool checked = tol >= 0.0;
  switch (phase) {
    case RESNMTF_PHASE_F: enqueue_phase_f(h, vs, false); break;
    case RESNMTF_PHASE_G: enqueue_phase_g(h, vs, -1.0, false); break;
    case RESNMTF_PHASE_S: break;   // S is complete behind PHASE_G on the handle's stream
    case RESNMTF_PHASE_F_ALL: enqueue_phase_f_all(h, checked); break;
    case RESNMTF_PHASE_LOCAL_SWEEP:
      if (h->block_p2p) {
        // peer-store form of the one-exchange layout: the F chain waits for the V blocks stored after sweep t - 1 (the first
        // ones travel by the caller's collective after resnmtf_prepare), says so to every rank once it has read them, and the
        // own block of this sweep is stored only after every rank has said so (the ack has long arrived: two passes lie between)
        if (sweep > 0 || h->opt.slice_p2p == 2) HIP_TRY(h, p2p_wait(h, 0, 0, sweep, 0));
        enqueue_phase_f_all(h);
        p2p_signal(h, 1);
        for (const auto& w : h->views)
          if (w.owned) enqueue_phase_g(h, w, -1.0, false);
        HIP_TRY(h, p2p_wait(h, 1, 1, sweep, 1));
        for (const auto& w : h->views)
          if (w.owned) push_f_block(h, w);
        p2p_signal(h, 0);
        break;
      }
      if (int rc = launch_local_sweep(h)) return rc;
      break;
    // block_p2p with the replicated G / S chains: the own T block is stored to the peers behind the Xt.F pass, the own U rows
    // and S block behind the X.G pass; G_ALL / S_ALL wait for the V arrivals of their sweep.  Single arenas: the sweep's own
    // order keeps a writer behind its readers, and what every rank computes itself (coefficients,
    // lambda, mu) is never stored to a peer
    case RESNMTF_PHASE_XTF:
      launch_pass(h, vs, false, 1, tol, checked); launch_fold_t(h, vs);
      if (h->block_p2p) { push_g_block(h, vs); p2p_signal(h, 1); }
      break;
    case RESNMTF_PHASE_G_ALL:
      if (h->block_p2p) HIP_TRY(h, p2p_wait(h, 1, 2, sweep, 1));
      if (h->wchain_ok[1]) { launch_wide_chain(h, 1, checked); break; }
      if (enqueue_g_chain(h, checked)) break;
      for (const auto& w : h->views)
        if (w.owned || w.g_replica) launch_update(h, w, 1, checked);
      break;
    case RESNMTF_PHASE_XG:
      launch_pass(h, vs, true, 1, tol, checked); launch_fold(h, vs);
      if (h->block_p2p) { push_f_block(h, vs); p2p_signal(h, 0); }
      break;
    // slice_p2p: every phase first waits (in stream order, hipStreamWaitValue32) until the V arrivals of the exchange that
    // feeds it are in, and ends with one arrival on every rank's counter of the exchange it fed with peer stores.
    // Arrivals so far: U + S blocks V (t + 2) after sweep t's X.G (the run prologue is the first), the others V (t + 1).
    case RESNMTF_PHASE_S_ALL:
      if (h->opt.slice_p2p) HIP_TRY(h, p2p_wait(h, 0, 3, sweep, h->block_p2p ? 1 : 2));
      if (int rc = launch_s_chain(h, checked)) return rc;
      break;
    case RESNMTF_PHASE_SLICE_F:
      if (h->opt.slice_p2p) HIP_TRY(h, p2p_wait(h, 0, 4, sweep, 1));
      launch_wide_chain(h, 0, checked, true);
      if (h->opt.slice_p2p) p2p_signal(h, 1);
      break;
    case RESNMTF_PHASE_SLICE_XTF:
      if (h->opt.slice_p2p) HIP_TRY(h, p2p_wait(h, 1, 5, sweep, 1));
      launch_slice_unpack(h, vs, 0, checked);
      launch_pass(h, vs, false, 1, tol, checked);
      launch_slice_pack(h, vs, false, checked);
      if (h->opt.slice_p2p) p2p_signal(h, 2);
      break;
    case RESNMTF_PHASE_SLICE_G:
      if (h->opt.slice_p2p) HIP_TRY(h, p2p_wait(h, 2, 6, sweep, 1));
      launch_wide_chain(h, 1, checked, true);
      if (h->opt.slice_p2p) p2p_signal(h, 3);
      break;
    case RESNMTF_PHASE_SLICE_XG:
      if (h->opt.slice_p2p) HIP_TRY(h, p2p_wait(h, 3, 7, sweep, 1));
      launch_slice_unpack(h, vs, 1, checked);
      launch_pass(h, vs, true, 1, tol, checked);
      launch_slice_pack(h, vs, true, checked);
      if (h->opt.slice_p2p) p2p_signal(h, 0);
      break;
    default: return h->fail(RESNMTF_ERR_INVALID, "unknown phase");
  }
  HIP_TRY(h, hipGetLastError());
  return RESNMTF_OK;
}

int resnmtf_run(resnmtf_handle* h, int n_iters, double tol, int max_iters, double* all_err, int err_capacity,
                int* iters_done) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (iters_done) *iters_done = 0;
  if (!h->all_owned) return h->fail(RESNMTF_ERR_STATE, "resnmtf_run needs a handle that owns every view; use the phase API");
  if (h->opt.replicate_gs) return h->fail(RESNMTF_ERR_STATE, "resnmtf_run does not drive the replicated G / S chains (replicate_gs); use the phase API");
  if (n_iters < 0) return h->fail(RESNMTF_ERR_INVALID, "n_iters must be >= 0");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  int total;
  double tol_arg;
  if (n_iters > 0) {
    total = n_iters; tol_arg = -1.0;
    if (all_err && err_capacity < n_iters) return h->fail(RESNMTF_ERR_INVALID, "all_err shorter than n_iters");
  } else {
    if (!(tol >= 0.0)) return h->fail(RESNMTF_ERR_INVALID, "tol must be >= 0 in convergence mode");
    tol_arg = tol;
    total = max_iters > 0 ? max_iters : err_capacity;
    if (all_err) total = std::min(total, err_capacity);
    if (total < 1) return h->fail(RESNMTF_ERR_INVALID, "convergence mode needs max_iters > 0 or an all_err buffer");
  }
  // RESNMTF_TRACE_RUN=1: host-side stage times of this call on stderr (diagnostic, tools/time_run_overhead.py)
  static const bool trace = std::getenv("RESNMTF_TRACE_RUN") != nullptr;
  using clk = std::chrono::steady_clock;
  clk::time_point tp[6];
  if (trace) tp[0] = clk::now();
  if (int rc = ensure_err_capacity(h, total)) return rc;
  // (the host mirror of the loop control is rewritten below: the previous run has been waited for)
  if (int rc = prepare_impl(h, true, tol_arg < 0.0)) return rc;
  const int base = h->sweep_base;      // sweeps the device counter stood at when this run began
  h->ctl_clean = false;
  if (trace) tp[1] = clk::now();
  h->resume_ok = false;
  const bool eager = !h->opt.use_graph || h->opt.time_kernels;
  const int batch = std::max(1, h->opt.check_every);
  if (!eager && h->graph_tol != tol_arg) {                 // graphs are captured per stop test
    if (!h->ladder.empty() || !h->exact.empty()) HIP_TRY(h, hipStreamSynchronize(h->stream));   // (the previous run returned on the
    destroy_graphs(h); h->graph_tol = tol_arg;                                                   //  counter: its graph may still execute)
  }
  int enq = 0;
  if (!eager && total > 1) {
    // the first sweep goes out as plain launches: its first kernel starts within a few microseconds, where a graph launch
    // costs the host ~20 us before the GPU sees anything (tools/time_run_overhead.py); the graph launches that follow
    // are enqueued while that sweep runs.  (Same kernels, same arguments: bit for bit the graph's.)
    enqueue_sweep(h, tol_arg);
    enq = 1;
  }
  // fixed-iteration runs shorter than three batches: ONE graph of exactly the sweeps left (captured at the first run of
  // that length, kept for the next ones -- a driver that times `run(20)` after a warm-up call of the same length pays
  // one graph launch, hidden behind the eager first sweep, and no graph-to-graph gaps)
  if (!eager && tol_arg < 0.0 && total - enq > 0 && total - enq < 3 * batch) {
    const int rest = total - enq;
    hipGraphExec_t ex = nullptr;
    for (const auto& g : h->exact)
      if (g.first == rest) ex = g.second;
    if (!ex) {
      if (h->exact.size() >= 4) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));           // (it may be the graph the previous run is still draining)
        (void)hipGraphExecDestroy(h->exact.front().second); h->exact.erase(h->exact.begin());
      }
      if (int rc = capture_graph(h, rest, tol_arg, &ex)) return rc;
      h->exact.emplace_back(rest, ex);
    }
    HIP_TRY(h, hipGraphLaunch(ex, h->stream));
    enq = total;
  }
  if (!eager && enq < total && (h->ladder.empty() || h->ladder.front().first != batch))
    if (int rc = capture_ladder(h, batch, tol_arg)) return rc;
  while (enq < total) {
    const int todo = std::min(batch, total - enq);
    if (eager) {
      for (int s = 0; s < todo; ++s) {
        enqueue_sweep(h, tol_arg);
        if (h->opt.time_kernels && h->ev_used + 8 * (size_t)h->V > h->ev.size())
          if (int rc = flush_timing(h)) return rc;
      }
    } else {
      // a full batch is one launch; a remainder a few (binary ladder), SMALLEST rung first: the host cost of a graph launch
      // grows with its nodes, and everything after the first launch is enqueued while the GPU already works
      int left = todo;
      std::vector<hipGraphExec_t> seq;
      for (const auto& rung : h->ladder)
        while (left >= rung.first) { seq.push_back(rung.second); left -= rung.first; }
      for (auto it = seq.rbegin(); it != seq.rend(); ++it) HIP_TRY(h, hipGraphLaunch(*it, h->stream));
    }
    enq += todo;
    if (tol_arg >= 0.0) {     // convergence mode (R/main.r:50-81): look at the (mirrored) device flag between batches
      HIP_TRY(h, hipGetLastError());
      if (int rc = sync_both(h)) return rc;
      if (h->ctl_host->done) break;
    }
  }
  HIP_TRY(h, hipGetLastError());
  if (trace) tp[2] = clk::now();
  if (tol_arg < 0.0 && !h->opt.time_kernels && h->last_owned >= 0 && h->opt.wait_mode == 0) {
    // fixed-iteration runs: wait for the sweep counter the LAST k x k job mirrors into pinned host memory (errors are
    // written, and fenced, before it) instead of a stream synchronisation, whose wake-up costs ~10 us of a 0.9 ms run;
    // the stream may still be draining the last launch's workgroups -- every other entry point synchronises it first.
    // Bounded: after 20 ms without progress the stream synchronisation below takes over.
    const int want = base + total;
    volatile int* counter = &h->ctl_host->sweep;
    int seen = *counter;
    auto t_last = clk::now();
    while (seen != want) {
      const int now = *counter;
      if (now != seen) { seen = now; t_last = clk::now(); }
      else if (std::chrono::duration<double, std::milli>(clk::now() - t_last).count() > 20.0) break;
      for (int p = 0; p < 16; ++p) __builtin_ia32_pause();     // (back off: the core's sibling thread and the memory bus get air)
    }
    if (seen != want)
      if (int rc = sync_both(h)) return rc;
  } else if (int rc = sync_both(h)) return rc;
  if (trace) tp[3] = clk::now();
  if (h->fuse_err && *h->fuse_err) {
    *h->fuse_err = 0;
    h->resume_ok = false;
    return h->fail(RESNMTF_ERR_HIP, "a fused pass launch gave up waiting for its update blocks (pass_fused_kernel): results are invalid");
  }
  const int done_total = h->ctl_host->sweep - base;
  if (tol_arg < 0.0 && done_total != total) return h->fail(RESNMTF_ERR_HIP, "sweep counter mismatch");
  if (all_err && done_total > 0) {
    for (int t = 0; t < done_total; ++t) {        // mean over views (R/main.r:77-78,104-107)
      double sum = 0.0;
      const size_t row = (size_t)((base + t) % h->err_cap) * h->V;
      for (int v = 0; v < h->V; ++v) sum += h->err_host[row + v];
      all_err[t] = sum / (double)h->V;
    }
  }
  if (h->opt.time_kernels)
    if (int rc = flush_timing(h)) return rc;
  if (iters_done) *iters_done = done_total;
  h->resume_ok = true;
  h->ctl_clean = tol_arg < 0.0;        // (a convergence run leaves its stop flag and prev_mean behind)
  h->next_base = base + done_total;
  if (trace) {
    tp[4] = clk::now();
    auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(tp[b] - tp[a]).count(); };
    std::fprintf(stderr, "[resnmtf_run] sweeps %d: prepare %.1f us, enqueue %.1f us, wait %.1f us, read-out %.1f us\n", total,
                 us(0, 1), us(1, 2), us(2, 3), us(3, 4));
  }
  return RESNMTF_OK;
}

int resnmtf_get_factors(resnmtf_handle* h, int v, double* F, double* S, double* G, double* lambda, double* mu) {
  if (int rc = check_view(h, v)) return rc;
  ViewState& vs = h->views[v];
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  std::vector<double> tmp;
  if (F) { tmp.resize((size_t)vs.n * vs.k); HIP_TRY(h, hipMemcpy(tmp.data(), vs.F, tmp.size() * sizeof(double), hipMemcpyDeviceToHost)); to_col_major(tmp, vs.n, vs.k, F); }
  if (S) { tmp.resize((size_t)vs.k * vs.k); HIP_TRY(h, hipMemcpy(tmp.data(), vs.S, tmp.size() * sizeof(double), hipMemcpyDeviceToHost)); to_col_major(tmp, vs.k, vs.k, S); }
  if (G) { tmp.resize((size_t)vs.m * vs.k); HIP_TRY(h, hipMemcpy(tmp.data(), vs.G, tmp.size() * sizeof(double), hipMemcpyDeviceToHost)); to_col_major(tmp, vs.m, vs.k, G); }
  if (lambda || mu) {
    if (!vs.owned) return h->fail(RESNMTF_ERR_STATE, "lambda/mu only exist on the owning handle");
    if (lambda) HIP_TRY(h, hipMemcpy(lambda, vs.lambda, (size_t)vs.k * sizeof(double), hipMemcpyDeviceToHost));
    if (mu) HIP_TRY(h, hipMemcpy(mu, vs.mu, (size_t)vs.k * sizeof(double), hipMemcpyDeviceToHost));
  }
  return RESNMTF_OK;
}

int resnmtf_finalise(resnmtf_handle* h, int v, double* F, double* S, double* G, double* row_clusters,
                     double* col_clusters) {
  if (int rc = check_view(h, v)) return rc;
  ViewState& vs = h->views[v];
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  const size_t nk = (size_t)vs.n * vs.k, mk = (size_t)vs.m * vs.k, kk = (size_t)vs.k * vs.k;
  double* buf = nullptr;   // [cF k][cG k][S kk][Fout nk][rc nk][Gout mk][cc mk]
  int* rel = nullptr;
  const size_t total = 2 * (size_t)vs.k + kk + 2 * nk + 2 * mk;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), total * sizeof(double)));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&rel), (size_t)vs.k * sizeof(int));
  if (e != hipSuccess) { (void)hipFree(buf); return h->fail_hip("hipMalloc", e); }
  double *cF = buf, *cG = cF + vs.k, *So = cG + vs.k, *Fo = So + kk, *rc_ = Fo + nk, *Go = rc_ + nk, *cc = Go + mk;
  hipLaunchKernelGGL(colsum_kernel, dim3(vs.k), dim3(256), 0, h->stream, vs.F, vs.n, vs.k, cF);
  hipLaunchKernelGGL(colsum_kernel, dim3(vs.k), dim3(256), 0, h->stream, vs.G, vs.m, vs.k, cG);
  hipLaunchKernelGGL(finalise_s_kernel, dim3(1), dim3(64), 0, h->stream, vs.S, vs.k, cF, cG, So, rel);
  hipLaunchKernelGGL(finalise_factor_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, h->stream, vs.F, vs.n,
                     vs.k, cF, rel, Fo, rc_);
  hipLaunchKernelGGL(finalise_factor_kernel, dim3((unsigned)((mk + 255) / 256)), dim3(256), 0, h->stream, vs.G, vs.m,
                     vs.k, cG, (const int*)nullptr, Go, cc);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess && S) e = hipMemcpy(S, So, kk * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && F) e = hipMemcpy(F, Fo, nk * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && G) e = hipMemcpy(G, Go, mk * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && row_clusters) e = hipMemcpy(row_clusters, rc_, nk * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && col_clusters) e = hipMemcpy(col_clusters, cc, mk * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  (void)hipFree(rel);
  if (e != hipSuccess) return h->fail_hip("finalise", e);
  return RESNMTF_OK;
}

int resnmtf_factor_device_ptr(resnmtf_handle* h, int v, int which, void** ptr, size_t* bytes) {
  if (int rc = check_view(h, v)) return rc;
  if (!ptr || !bytes) return h->fail(RESNMTF_ERR_INVALID, "ptr/bytes are NULL");
  ViewState& vs = h->views[v];
  switch (which) {
    case RESNMTF_FACTOR_F: *ptr = vs.F; *bytes = (size_t)vs.n * vs.k * sizeof(double); break;
    case RESNMTF_FACTOR_G: *ptr = vs.G; *bytes = (size_t)vs.m * vs.k * sizeof(double); break;
    case RESNMTF_FACTOR_S: *ptr = vs.S; *bytes = (size_t)vs.k * vs.k * sizeof(double); break;
    case RESNMTF_FACTOR_FBLOCK:
      if (!vs.fblk) return h->fail(RESNMTF_ERR_STATE, "no F exchange block: create the handle with replicate_f = 1");
      *ptr = vs.fblk; *bytes = vs.fblk_bytes; break;
    case RESNMTF_FACTOR_FBLOCK_ALL:
      if (!h->fblk_arena) return h->fail(RESNMTF_ERR_STATE, "no F exchange blocks: create the handle with replicate_f = 1");
      *ptr = h->fblk_arena; *bytes = h->fblk_arena_bytes; break;
    case RESNMTF_FACTOR_GBLOCK:
      if (!vs.gblk) return h->fail(RESNMTF_ERR_STATE, "no G exchange block: create the handle with replicate_gs = 1");
      *ptr = vs.gblk; *bytes = vs.gblk_bytes; break;
    case RESNMTF_FACTOR_GBLOCK_ALL:
      if (!h->gblk_arena) return h->fail(RESNMTF_ERR_STATE, "no G exchange blocks: create the handle with replicate_gs = 1");
      *ptr = h->gblk_arena; *bytes = h->gblk_arena_bytes; break;
    case RESNMTF_FACTOR_SBLOCK:
      if (!vs.sblk) return h->fail(RESNMTF_ERR_STATE, "no S exchange block: create the handle with replicate_gs = 1");
      *ptr = vs.sblk; *bytes = h->sblk_stride * sizeof(double); break;
    case RESNMTF_FACTOR_SBLOCK_ALL:
      if (h->sblk_embedded) return h->fail(RESNMTF_ERR_STATE, "the S blocks of this handle sit inside the F blocks (RESNMTF_FACTOR_FBLOCK_ALL moves both)");
      if (!h->sblk_arena) return h->fail(RESNMTF_ERR_STATE, "no S exchange blocks: create the handle with replicate_gs = 1");
      *ptr = h->sblk_arena; *bytes = h->sblk_stride * sizeof(double) * h->V; break;
    case RESNMTF_FACTOR_U_SEND: case RESNMTF_FACTOR_U_RECV: case RESNMTF_FACTOR_FNEW_SEND: case RESNMTF_FACTOR_FNEW_RECV:
    case RESNMTF_FACTOR_T_SEND: case RESNMTF_FACTOR_T_RECV: case RESNMTF_FACTOR_GNEW_SEND: case RESNMTF_FACTOR_GNEW_RECV:
    case RESNMTF_FACTOR_F_SLICE: case RESNMTF_FACTOR_G_SLICE: {
      if (!h->sliced) return h->fail(RESNMTF_ERR_STATE, "no slice buffers: create the handle with slice_chains = 1");
      const size_t V = (size_t)h->V;
      const size_t fbytes = V * h->sl_rows * vs.KP * sizeof(float), gbytes = V * h->sl_cols * vs.KP * sizeof(float);
      switch (which) {
        case RESNMTF_FACTOR_U_SEND: *ptr = h->u_send; *bytes = V * h->u_chunk; break;
        case RESNMTF_FACTOR_U_RECV: *ptr = h->u_recv; *bytes = V * h->u_chunk; break;
        case RESNMTF_FACTOR_T_SEND: *ptr = h->t_send; *bytes = V * h->t_chunk; break;
        case RESNMTF_FACTOR_T_RECV: *ptr = h->t_recv; *bytes = V * h->t_chunk; break;
        case RESNMTF_FACTOR_FNEW_SEND: *ptr = h->f_send; *bytes = fbytes; break;
        case RESNMTF_FACTOR_FNEW_RECV: *ptr = h->f_recv; *bytes = fbytes; break;
        case RESNMTF_FACTOR_GNEW_SEND: *ptr = h->g_send; *bytes = gbytes; break;
        case RESNMTF_FACTOR_GNEW_RECV: *ptr = h->g_recv; *bytes = gbytes; break;
        default: {
          const bool f = which == RESNMTF_FACTOR_F_SLICE;
          const int per = f ? h->sl_rows : h->sl_cols, full = f ? vs.n : vs.m;
          const int begin = std::min(h->opt.slice_index * per, full), len = std::min(per, full - begin);
          *ptr = (f ? vs.F : vs.G) + (size_t)begin * vs.k; *bytes = (size_t)len * vs.k * sizeof(double);
        }
      }
      break;
    }
    default: return h->fail(RESNMTF_ERR_INVALID, "unknown factor selector");
  }
  return RESNMTF_OK;
}

int resnmtf_set_stop_tolerance(resnmtf_handle* h, double tol) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (tol >= 0.0 && !h->opt.replicate_gs)
    return h->fail(RESNMTF_ERR_STATE, "the phase API's stop test runs in the replicated S chain (replicate_gs / slice_chains); resnmtf_run has its own");
  h->phase_tol = tol >= 0.0 ? tol : -1.0;
  return RESNMTF_OK;
}

int resnmtf_loop_state(resnmtf_handle* h, int* sweeps_done, int* done, int* stop_sweep) {
  if (!h) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  SweepCtl c{};
  HIP_TRY(h, hipMemcpy(&c, h->ctl, sizeof(c), hipMemcpyDeviceToHost));
  if (sweeps_done) *sweeps_done = c.sweep;
  if (done) *done = c.done;
  if (stop_sweep) *stop_sweep = c.stop_sweep;
  return RESNMTF_OK;
}

int resnmtf_slice_info(resnmtf_handle* h, int* rows_per_slice, int* cols_per_slice) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (!h->sliced) return h->fail(RESNMTF_ERR_STATE, "not a slice_chains handle");
  if (rows_per_slice) *rows_per_slice = h->sl_rows;
  if (cols_per_slice) *cols_per_slice = h->sl_cols;
  return RESNMTF_OK;
}

// what a rank maps of every other rank: sliced chains -- the four receive buffers and the S block arena; block form -- the
// exchange arenas of the replicated layouts (those the layout has); the arrival counters in both
static void p2p_buffers(resnmtf_handle* h, void* bufs[6]) {
  if (h->block_p2p) {
    bufs[0] = h->fblk_arena; bufs[1] = nullptr; bufs[2] = h->gblk_arena; bufs[3] = nullptr; bufs[4] = h->sblk_arena;
  } else {
    bufs[0] = h->u_recv; bufs[1] = h->f_recv; bufs[2] = h->t_recv; bufs[3] = h->g_recv; bufs[4] = h->sblk_arena;
  }
  bufs[5] = h->p2p_flags;
}

int resnmtf_p2p_export(resnmtf_handle* h, void* handles, size_t capacity, size_t* bytes) {
  if (!h || !bytes) return RESNMTF_ERR_INVALID;
  if (!h->opt.slice_p2p) return h->fail(RESNMTF_ERR_STATE, "not a slice_p2p handle");
  *bytes = 6 * sizeof(hipIpcMemHandle_t);
  if (!handles || capacity < *bytes) return h->fail(RESNMTF_ERR_INVALID, "handle buffer too small");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  void* bufs[6];
  p2p_buffers(h, bufs);
  auto* out = static_cast<hipIpcMemHandle_t*>(handles);
  std::memset(out, 0, *bytes);
  for (int b = 0; b < 6; ++b)
    if (bufs[b]) HIP_TRY(h, hipIpcGetMemHandle(&out[b], bufs[b]));
  return RESNMTF_OK;
}

int resnmtf_p2p_import(resnmtf_handle* h, int rank, const void* handles, size_t bytes) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (!h->opt.slice_p2p) return h->fail(RESNMTF_ERR_STATE, "not a slice_p2p handle");
  if (rank < 0 || rank >= h->V) return h->fail(RESNMTF_ERR_INVALID, "rank out of range");
  resnmtf_handle::Peer& pc = h->peers[(size_t)rank];
  if (pc.imported) return h->fail(RESNMTF_ERR_STATE, "rank already imported");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  void* ptr[6];
  void* own[6];
  p2p_buffers(h, own);
  if (rank == h->opt.slice_index) {
    for (int b = 0; b < 6; ++b) ptr[b] = own[b];
  } else {
    if (!handles || bytes < 6 * sizeof(hipIpcMemHandle_t)) return h->fail(RESNMTF_ERR_INVALID, "handle buffer too small");
    const auto* in = static_cast<const hipIpcMemHandle_t*>(handles);
    for (int b = 0; b < 6; ++b) {
      ptr[b] = nullptr;
      if (!own[b]) continue;              // (every rank has the same layout: a buffer this rank lacks, the peer lacks too)
      HIP_TRY(h, hipIpcOpenMemHandle(&ptr[b], in[b], hipIpcMemLazyEnablePeerAccess));
      pc.opened[b] = ptr[b];
    }
  }
  if (h->block_p2p) {
    pc.fblk_arena = static_cast<char*>(ptr[0]); pc.gblk_arena = static_cast<char*>(ptr[2]);
  } else {
    pc.u_recv = static_cast<char*>(ptr[0]); pc.f_recv = static_cast<float*>(ptr[1]); pc.t_recv = static_cast<char*>(ptr[2]);
    pc.g_recv = static_cast<float*>(ptr[3]);
  }
  pc.sblk = static_cast<double*>(ptr[4]); pc.flags = static_cast<unsigned int*>(ptr[5]);
  pc.imported = true;
  h->p2p_ready = true;
  for (const auto& q : h->peers) h->p2p_ready = h->p2p_ready && q.imported;
  h->prepared = false;
  return RESNMTF_OK;
}

// Every rank calls this at about the same time, after all imports and a host barrier, before resnmtf_prepare.  Host-side
// deadlines everywhere: a node on which peer stores, remote atomics or the stream wait do not work is reported, not hung on.
int resnmtf_p2p_selftest(resnmtf_handle* h, int timeout_ms) {
  if (!h) return RESNMTF_ERR_INVALID;
  if (!h->opt.slice_p2p) return h->fail(RESNMTF_ERR_STATE, "not a slice_p2p handle");
  if (!h->p2p_ready) return h->fail(RESNMTF_ERR_STATE, "slice_p2p: import every rank's buffers first (resnmtf_p2p_import)");
  if (h->p2p_prepared) return h->fail(RESNMTF_ERR_STATE, "resnmtf_p2p_selftest precedes resnmtf_prepare (it writes into the receive buffers)");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  const int V = h->V, r = h->opt.slice_index, kProbeFlag = 8, kWords = 256;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 10000);
  // where rank `from` writes on rank `on` (a place the run prologue overwrites): block form -- the head of view `from`'s U
  // rows in the F arena; sliced form -- the head of chunk `from` of the U receive buffer
  auto region = [&](char* f_arena, char* u_recv, int from) -> unsigned int* {
    if (h->block_p2p) return reinterpret_cast<unsigned int*>(f_arena + (static_cast<const char*>(h->views[(size_t)from].fblk) - static_cast<const char*>(h->fblk_arena)));
    return reinterpret_cast<unsigned int*>(u_recv + (size_t)from * h->u_chunk);
  };
  // host-side waits, all under the one deadline
  auto poll_flag = [&](int flag, unsigned int want, const char* what) -> int {
    unsigned int seen = 0;
    for (;;) {
      HIP_TRY(h, hipMemcpy(&seen, h->p2p_flags + flag, sizeof(seen), hipMemcpyDeviceToHost));
      if (seen >= want) return RESNMTF_OK;
      if (std::chrono::steady_clock::now() > deadline) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "slice_p2p self-test: %u of %u %s within the deadline (peer atomics do not reach this rank)", seen, want, what);
        return h->fail(RESNMTF_ERR_HIP, msg);
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  };
  auto drain_stream = [&](int flag) -> int {
    for (;;) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) return RESNMTF_OK;
      if (q != hipErrorNotReady) return h->fail_hip("hipStreamQuery", q);
      if (std::chrono::steady_clock::now() > deadline) {
        // last resort so that the handle can still be destroyed: satisfy the wait from the host (the error is what the caller acts on)
        const unsigned int all = 0x7FFFFFFFu;
        (void)hipMemcpy(h->p2p_flags + flag, &all, sizeof(all), hipMemcpyHostToDevice);
        return h->fail(RESNMTF_ERR_HIP, "slice_p2p self-test: hipStreamWaitValue32 does not see the arrival counter");
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  };
  int* bad = nullptr; int* bad_dev = nullptr;
  HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&bad), sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  *bad = 0;
  struct Guard { int* p; ~Guard() { if (p) (void)hipHostFree(p); } } guard{bad};
  HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&bad_dev), bad, 0));
  const int kAckFlag = 9;
  // Two rounds with different words.  Each: store to every peer + one arrival on every rank's counter; the host sees the V
  // arrivals; the stream wait of the phases + a kernel that reads what arrived (as in a sweep -- the second round reads lines
  // the first left in this device's caches); an acknowledgement round so that nobody overwrites what a peer has not read yet.
  for (int round = 0; round < 2; ++round) {
    const unsigned int step = 2u * h->probe_epoch + (unsigned)round;
    auto tag_of = [&](int rank) { return 0xA5000000u | ((unsigned)rank << 16) | ((step & 0xFFu) << 8); };
    P2pProbeArgs a{};
    a.tag = tag_of(r);
    for (int c = 0; c < V; ++c)
      if (c != r) a.dst[a.n_dst++] = region(h->peers[(size_t)c].fblk_arena, h->peers[(size_t)c].u_recv, r);
    if (a.n_dst) hipLaunchKernelGGL(p2p_probe_kernel, dim3(1), dim3(kWords), 0, h->stream, a);
    p2p_signal(h, kProbeFlag);
    HIP_TRY(h, hipGetLastError());
    const unsigned int want = (unsigned)V * (step + 1);
    if (int rc = poll_flag(kProbeFlag, want, "arrivals")) return rc;
    HIP_TRY(h, hipStreamWaitValue32(h->stream, h->p2p_flags + kProbeFlag, want, hipStreamWaitValueGte, 0xFFFFFFFFu));
    P2pCheckArgs ck{};
    for (int c = 0; c < V; ++c)
      if (c != r) { ck.src[ck.n_src] = region(static_cast<char*>(h->fblk_arena), h->u_recv, c); ck.tag[ck.n_src++] = tag_of(c); }
    ck.bad = bad_dev;
    if (ck.n_src) hipLaunchKernelGGL(p2p_check_kernel, dim3(1), dim3(kWords), 0, h->stream, ck);
    HIP_TRY(h, hipGetLastError());
    if (int rc = drain_stream(kProbeFlag)) return rc;
    if (*bad) {
      char msg[200];
      std::snprintf(msg, sizeof(msg), "slice_p2p self-test: %d words read by a kernel after the stream wait are not what the peers stored (round %d%s)",
                    *bad, round, round ? ": stale cache lines" : "");
      return h->fail(RESNMTF_ERR_HIP, msg);
    }
    p2p_signal(h, kAckFlag);                                   // "I have read round `round`"
    HIP_TRY(h, hipGetLastError());
    if (int rc = poll_flag(kAckFlag, want, "acknowledgements")) return rc;
  }
  h->probe_epoch += 1;
  for (int c = 0; c < V; ++c)
    if (c != r) HIP_TRY(h, hipMemset(region(static_cast<char*>(h->fblk_arena), h->u_recv, c), 0, kWords * sizeof(unsigned int)));
  return RESNMTF_OK;
}

int resnmtf_kernel_timings(resnmtf_handle* h, double* ms_total, long long* launches, int reset) {
  if (!h || !ms_total || !launches) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = flush_timing(h)) return rc;
  for (int i = 0; i < RESNMTF_TIMED_KINDS; ++i) { ms_total[i] = h->ktime_ms[i]; launches[i] = h->klaunch[i]; }
  if (reset)
    for (int i = 0; i < RESNMTF_TIMED_KINDS; ++i) { h->ktime_ms[i] = 0.0; h->klaunch[i] = 0; }
  return RESNMTF_OK;
}

int resnmtf_view_errors(resnmtf_handle* h, int v, int first, int count, double* out) {
  if (int rc = check_view(h, v)) return rc;
  if (!out || first < 0 || count < 0) return h->fail(RESNMTF_ERR_INVALID, "bad error range");
  if (first + count > h->err_cap) return h->fail(RESNMTF_ERR_INVALID, "range beyond the reserved error buffer");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  // (sweep t of the latest run / of the phases since resnmtf_prepare sits in row (base + t) mod capacity)
  std::vector<double> buf((size_t)h->err_cap * h->V);
  if (count > 0)
    HIP_TRY(h, hipMemcpy(buf.data(), h->err, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int t = 0; t < count; ++t) out[t] = buf[(size_t)((h->sweep_base + first + t) % h->err_cap) * h->V + v];
  return RESNMTF_OK;
}

int resnmtf_synchronize(resnmtf_handle* h) {
  if (!h) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = sync_both(h)) return rc;
  if (h->fuse_err && *h->fuse_err == 2) {
    *h->fuse_err = 0;
    return h->fail(RESNMTF_ERR_HIP, "a peer-store wait gave up (p2p_wait_kernel: the arrivals of an exchange never came): results are invalid");
  }
  if (h->fuse_err && *h->fuse_err) {
    *h->fuse_err = 0;
    return h->fail(RESNMTF_ERR_HIP, "a fused pass launch gave up waiting for its update blocks (pass_fused_kernel): results are invalid");
  }
  return RESNMTF_OK;
}

#ifdef RESNMTF_STAMPS
// diagnostic build only: point the stamp buffer at `buf` ([blocks][16] u64) or detach it (NULL)
int resnmtf_debug_set_stamp_buffer(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : RESNMTF_ERR_HIP;
}
// 1 = only the pass launches stamp, 2 = only the update kernels, 3 = both (they share the columns)
int resnmtf_debug_set_stamp_select(int sel) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sel), &sel, sizeof(sel)) == hipSuccess ? 0 : RESNMTF_ERR_HIP;
}
#endif

int resnmtf_view_image_info(resnmtf_handle* h, int v, int* uses_2byte, double* rel_error) {
  if (int rc = check_view(h, v)) return rc;
  const ViewState& vs = h->views[v];
  if (uses_2byte) *uses_2byte = vs.half ? (vs.u16 ? 2 : 1) : 0;
  if (rel_error) *rel_error = vs.half_capable ? vs.x_relerr : 0.0;
  return RESNMTF_OK;
}

int resnmtf_pass_timings(resnmtf_handle* h, resnmtf_pass_timing* out, int reset) {
  if (!h || !out) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (int rc = flush_timing(h)) return rc;
  *out = h->timing;
  // algorithmic bytes / flops of ONE launch of the first owned view's passes
  for (const auto& v : h->views) {
    if (!v.owned) continue;
    const double n = v.n, m = v.m, k = v.k;
    const double sx = v.half ? 2.0 : 4.0;          // bytes per element of X as stored
    // (a launch that carries the update of its B operand -- pass_fused_kernel -- also moves that update's algorithmic bytes:
    //  product rows 4 len k once, the fp64 factor 8 len k in and out, its f32 copy 4 len k out)
    out->xg_bytes = sx * n * m + 4.0 * (n + m) * k + (can_fuse_update(h, v, 1) ? 24.0 * m * k : 0.0);
    out->xtf_bytes = sx * n * m + 4.0 * (n + m) * k + (can_fuse_update(h, v, 0) && h->all_owned && h->chain_views == 0 ? 24.0 * n * k : 0.0);
    out->xg_flops = 2.0 * n * m * k;
    out->xtf_flops = 2.0 * n * m * k;
    break;
  }
  if (reset) h->timing = resnmtf_pass_timing{};
  return RESNMTF_OK;
}

}  // extern "C"
