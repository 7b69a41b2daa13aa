// resnmtf_hip.hip -- host side of libresnmtf_hip.so: handle, device memory, launch schedule,
// hipGraph capture and the C-ABI declared in include/resnmtf_hip.h.
//
// Schedule of one sweep (R/update_steps.r:272-319) for owned views v = 0..V-1, in order:
//     F_v :  factor_update<F>(U_v)                       U_v = X_v G_v from the previous pass
//     G_v :  pass Xt.F  ->  factor_update<G>(T_v)  ->  pass X.G'   (also G'^T G' and T^T G')
//   then, for v = 0..V-1:
//     S_v :  s_update (S, lambda, mu, error, next F coefficients)
// S_v may trail the other views' F/G updates because no F or G rule reads another view's S and
// the xi coupling only needs S_w (w < v) updated first -- the order inside the trailing loop.
// A run starts with one X.G pass per view (state after set_factors has no U yet).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "resnmtf_hip.h"
#include "resnmtf_kernels.hip.inc"

namespace {

std::string g_create_error;

inline int round_up(int x, int a) { return (x + a - 1) / a * a; }
inline int ceil_div(int x, int a) { return (x + a - 1) / a; }

struct SharedMap {
  bool set = false;        // false or count < 0  ->  NA
  int count = -1;
  int* dev = nullptr;      // [len_v] row of w or -1
};

struct ViewState {
  int n = 0, m = 0, k = 0, KP = 16, NT = 1;
  bool owned = true, has_x = false, has_factors = false;
  int n_pad = 0, m_pad = 0;
  float *X32 = nullptr, *Xt32 = nullptr;
  double* xnorm2 = nullptr;
  double *F = nullptr, *G = nullptr, *S = nullptr, *lambda = nullptr, *mu = nullptr;
  float *F32 = nullptr, *G32 = nullptr, *T32 = nullptr;
  int nsplit_xg = 1, rps_xg = 16, cols_total_xg = 0;
  int nsplit_xtf = 1, rps_xtf = 16, cols_total_xtf = 0;
  float *Pxg = nullptr, *Pxtf = nullptr;
  int rpbF = 16, nblkF = 1, rpbG = 16, nblkG = 1;
  double *colsumF_part = nullptr, *colsumG_part = nullptr;
  double *FtF = nullptr, *Ma_F = nullptr, *Md_F = nullptr;
  std::vector<SharedMap> row_map, col_map;   // indexed by the other view
  UpdateArgs argF{}, argG{};
  SArgs argS{};
  PassArgs passXG{}, passXtF{};
};

}  // namespace

struct resnmtf_handle {
  int V = 0;
  std::vector<ViewState> views;
  resnmtf_options opt{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<double> phi, xi, psi;   // V x V column-major
  SweepCtl* ctl = nullptr;
  double* err = nullptr;              // [err_cap][V]
  double* mean_err = nullptr;         // [err_cap]
  int err_cap = 4096;
  bool prepared = false;
  bool all_owned = true;
  // graphs
  hipGraphExec_t graph_multi = nullptr, graph_one = nullptr;
  int graph_multi_sweeps = 0;
  double graph_tol = -2.0;
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

void free_view(ViewState& v) {
  void* ptrs[] = {v.X32, v.Xt32, v.xnorm2, v.F, v.G, v.S, v.lambda, v.mu, v.F32, v.G32, v.T32, v.Pxg, v.Pxtf,
                  v.colsumF_part, v.colsumG_part, v.FtF, v.Ma_F, v.Md_F};
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

size_t update_smem_bytes(int KP, bool is_g) {
  const int RG = 256 / KP;
  return sizeof(double) * ((size_t)(is_g ? 4 : 2) * KP * KP + 3 * (size_t)RG * KP);
}
size_t s_smem_bytes(int KP) { return sizeof(double) * ((size_t)4 * KP * KP + 256); }

template <int KP>
hipError_t set_smem_attrs() {
  hipError_t e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_update_kernel<KP, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)update_smem_bytes(KP, false));
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_update_kernel<KP, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)update_smem_bytes(KP, true));
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&s_update_kernel<KP>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)s_smem_bytes(KP));
}

void launch_pass(resnmtf_handle* h, const ViewState& v, bool xg) {
  const PassArgs& a = xg ? v.passXG : v.passXtF;
  const int ntiles = a.cols_total / 64;
  const int nsplit = xg ? v.nsplit_xg : v.nsplit_xtf;
  dim3 grid(ntiles, nsplit), block(256);
  const bool timed = h->opt.time_kernels && h->ev_used + 2 <= h->ev.size();
  if (timed) (void)hipEventRecord(h->ev[h->ev_used], h->stream);
  switch (v.NT) {
    case 1: hipLaunchKernelGGL((atb_pass_kernel<1, 8>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((atb_pass_kernel<2, 4>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((atb_pass_kernel<3, 4>), grid, block, 0, h->stream, a); break;
    default: hipLaunchKernelGGL((atb_pass_kernel<4, 4>), grid, block, 0, h->stream, a); break;
  }
  if (timed) {
    (void)hipEventRecord(h->ev[h->ev_used + 1], h->stream);
    h->ev_kind[h->ev_used / 2] = xg ? 0 : 1;
    h->ev_used += 2;
  }
}

void launch_update(resnmtf_handle* h, const ViewState& v, bool is_g) {
  const UpdateArgs& a = is_g ? v.argG : v.argF;
  const int nblk = is_g ? v.nblkG : v.nblkF;
  const size_t smem = update_smem_bytes(v.KP, is_g);
#define LAUNCH_UPD(KPV)                                                                                        \
  if (is_g) hipLaunchKernelGGL((factor_update_kernel<KPV, true>), dim3(nblk), dim3(256), smem, h->stream, a);   \
  else hipLaunchKernelGGL((factor_update_kernel<KPV, false>), dim3(nblk), dim3(256), smem, h->stream, a)
  switch (v.NT) {
    case 1: LAUNCH_UPD(16); break;
    case 2: LAUNCH_UPD(32); break;
    case 3: LAUNCH_UPD(48); break;
    default: LAUNCH_UPD(64); break;
  }
#undef LAUNCH_UPD
}

void launch_s(resnmtf_handle* h, const ViewState& v, int mode, int sweep_offset, bool use_ctl_sweep) {
  SArgs a = v.argS;
  a.mode = mode;
  a.sweep_offset = sweep_offset;
  a.use_ctl_sweep = use_ctl_sweep ? 1 : 0;
  const size_t smem = s_smem_bytes(v.KP);
  switch (v.NT) {
    case 1: hipLaunchKernelGGL((s_update_kernel<16>), dim3(1), dim3(256), smem, h->stream, a); break;
    case 2: hipLaunchKernelGGL((s_update_kernel<32>), dim3(1), dim3(256), smem, h->stream, a); break;
    case 3: hipLaunchKernelGGL((s_update_kernel<48>), dim3(1), dim3(256), smem, h->stream, a); break;
    default: hipLaunchKernelGGL((s_update_kernel<64>), dim3(1), dim3(256), smem, h->stream, a); break;
  }
}

// the three phases of one view (see header comment)
void enqueue_phase_f(resnmtf_handle* h, const ViewState& v) { launch_update(h, v, false); }
void enqueue_phase_g(resnmtf_handle* h, const ViewState& v) {
  launch_pass(h, v, false);
  launch_update(h, v, true);
  launch_pass(h, v, true);
}
void enqueue_phase_s(resnmtf_handle* h, const ViewState& v, int sweep_offset, bool use_ctl) {
  launch_s(h, v, 1, sweep_offset, use_ctl);
}

void enqueue_sweep(resnmtf_handle* h, double tol) {
  for (const auto& v : h->views) {
    enqueue_phase_f(h, v);
    enqueue_phase_g(h, v);
  }
  for (const auto& v : h->views) enqueue_phase_s(h, v, 0, true);
  hipLaunchKernelGGL(end_sweep_kernel, dim3(1), dim3(64), 0, h->stream, h->ctl, h->err, h->V, h->err_cap,
                     h->mean_err, tol);
}

void destroy_graphs(resnmtf_handle* h) {
  if (h->graph_multi) (void)hipGraphExecDestroy(h->graph_multi);
  if (h->graph_one) (void)hipGraphExecDestroy(h->graph_one);
  h->graph_multi = h->graph_one = nullptr;
  h->graph_multi_sweeps = 0;
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
  return RESNMTF_OK;
}

int flush_timing(resnmtf_handle* h) {
  if (h->ev_used == 0) return RESNMTF_OK;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    if (h->ev_kind[i / 2] == 0) { h->timing.xg_ms_total += ms; h->timing.xg_launches++; }
    else { h->timing.xtf_ms_total += ms; h->timing.xtf_launches++; }
  }
  h->ev_used = 0;
  return RESNMTF_OK;
}

// sizes the splits of a streaming pass: ~target workgroups in total, row ranges multiples of 16
void size_pass(int ntiles, int rows_pad, int target, int* nsplit, int* rps) {
  int ns = std::max(1, (target + ntiles / 2) / ntiles);
  ns = std::min(ns, rows_pad / 64);
  ns = std::max(ns, 1);
  int r = round_up(ceil_div(rows_pad, ns), 16);
  *rps = r;
  *nsplit = ceil_div(rows_pad, r);
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
  o->check_every = 8;
  o->target_workgroups = 0;
  o->time_kernels = 0;
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
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_error = "no HIP device available (this library has no CPU fallback)";
    return RESNMTF_ERR_NO_DEVICE;
  }
  if (o.device_id < 0 || o.device_id >= ndev) { g_create_error = "device_id out of range"; return RESNMTF_ERR_INVALID; }
  hipError_t e = hipSetDevice(o.device_id);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return RESNMTF_ERR_HIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, o.device_id) == hipSuccess) {
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
      return RESNMTF_ERR_NO_DEVICE;
    }
  }
  auto* h = new resnmtf_handle();
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
    vs.owned = owned ? owned[v] != 0 : true;
    if (!vs.owned) h->all_owned = false;
    vs.row_map.resize(n_views);
    vs.col_map.resize(n_views);
  }
  int rc = RESNMTF_OK;
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
  if ((e = dev_alloc_zero(&h->err, (size_t)h->err_cap * n_views)) != hipSuccess) return bail(e, "hipMalloc err");
  if ((e = dev_alloc_zero(&h->mean_err, (size_t)h->err_cap)) != hipSuccess) return bail(e, "hipMalloc mean_err");
  for (int v = 0; v < n_views; ++v) {
    ViewState& vs = h->views[v];
    const size_t kk = (size_t)vs.k * vs.k;
    if ((e = dev_alloc_zero(&vs.F, (size_t)vs.n * vs.k)) != hipSuccess) return bail(e, "hipMalloc F");
    if ((e = dev_alloc_zero(&vs.G, (size_t)vs.m * vs.k)) != hipSuccess) return bail(e, "hipMalloc G");
    if ((e = dev_alloc_zero(&vs.S, kk)) != hipSuccess) return bail(e, "hipMalloc S");
    if (!vs.owned) continue;
    if ((e = dev_alloc_zero(&vs.lambda, (size_t)vs.k)) != hipSuccess) return bail(e, "hipMalloc lambda");
    if ((e = dev_alloc_zero(&vs.mu, (size_t)vs.k)) != hipSuccess) return bail(e, "hipMalloc mu");
    if ((e = dev_alloc_zero(&vs.xnorm2, 1)) != hipSuccess) return bail(e, "hipMalloc xnorm2");
    if ((e = dev_alloc_zero(&vs.X32, (size_t)vs.n_pad * vs.m_pad)) != hipSuccess) return bail(e, "hipMalloc X32");
    if ((e = dev_alloc_zero(&vs.Xt32, (size_t)vs.m_pad * vs.n_pad)) != hipSuccess) return bail(e, "hipMalloc Xt32");
    if ((e = dev_alloc_zero(&vs.F32, (size_t)vs.n_pad * 64)) != hipSuccess) return bail(e, "hipMalloc F32");
    if ((e = dev_alloc_zero(&vs.G32, (size_t)vs.m_pad * 64)) != hipSuccess) return bail(e, "hipMalloc G32");
    if ((e = dev_alloc_zero(&vs.T32, (size_t)vs.m_pad * 64)) != hipSuccess) return bail(e, "hipMalloc T32");
    if ((e = dev_alloc_zero(&vs.FtF, kk)) != hipSuccess) return bail(e, "hipMalloc FtF");
    if ((e = dev_alloc_zero(&vs.Ma_F, kk)) != hipSuccess) return bail(e, "hipMalloc Ma_F");
    if ((e = dev_alloc_zero(&vs.Md_F, kk)) != hipSuccess) return bail(e, "hipMalloc Md_F");
    const int target = o.target_workgroups > 0 ? o.target_workgroups : 1024;
    vs.cols_total_xg = vs.n_pad + 128;
    vs.cols_total_xtf = vs.m_pad + 64;
    size_pass(vs.cols_total_xg / 64, vs.m_pad, target, &vs.nsplit_xg, &vs.rps_xg);
    size_pass(vs.cols_total_xtf / 64, vs.n_pad, target, &vs.nsplit_xtf, &vs.rps_xtf);
    if ((e = dev_alloc_zero(&vs.Pxg, (size_t)vs.nsplit_xg * vs.cols_total_xg * vs.KP)) != hipSuccess) return bail(e, "hipMalloc Pxg");
    if ((e = dev_alloc_zero(&vs.Pxtf, (size_t)vs.nsplit_xtf * vs.cols_total_xtf * vs.KP)) != hipSuccess) return bail(e, "hipMalloc Pxtf");
    const int RG = 256 / vs.KP;
    vs.rpbF = round_up(ceil_div(vs.n, 256), RG); vs.nblkF = ceil_div(vs.n, vs.rpbF);
    vs.rpbG = round_up(ceil_div(vs.m, 256), RG); vs.nblkG = ceil_div(vs.m, vs.rpbG);
    if ((e = dev_alloc_zero(&vs.colsumF_part, (size_t)vs.nblkF * vs.k)) != hipSuccess) return bail(e, "hipMalloc colsumF");
    if ((e = dev_alloc_zero(&vs.colsumG_part, (size_t)vs.nblkG * vs.k)) != hipSuccess) return bail(e, "hipMalloc colsumG");
  }
  if ((e = set_smem_attrs<16>()) != hipSuccess) return bail(e, "hipFuncSetAttribute");
  if ((e = set_smem_attrs<32>()) != hipSuccess) return bail(e, "hipFuncSetAttribute");
  if ((e = set_smem_attrs<48>()) != hipSuccess) return bail(e, "hipFuncSetAttribute");
  if ((e = set_smem_attrs<64>()) != hipSuccess) return bail(e, "hipFuncSetAttribute");
  if (o.time_kernels) {
    h->ev.resize(8192);
    h->ev_kind.resize(4096);
    for (auto& evt : h->ev)
      if ((e = hipEventCreate(&evt)) != hipSuccess) return bail(e, "hipEventCreate");
  }
  (void)rc;
  *out = h;
  return RESNMTF_OK;
}

int resnmtf_destroy(resnmtf_handle* h) {
  if (!h) return RESNMTF_OK;
  (void)hipSetDevice(h->opt.device_id);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  destroy_graphs(h);
  for (auto& v : h->views) free_view(v);
  if (h->ctl) (void)hipFree(h->ctl);
  if (h->err) (void)hipFree(h->err);
  if (h->mean_err) (void)hipFree(h->mean_err);
  for (auto& evt : h->ev)
    if (evt) (void)hipEventDestroy(evt);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RESNMTF_OK;
}

int resnmtf_set_view(resnmtf_handle* h, int v, const double* x) {
  if (int rc = check_view(h, v)) return rc;
  if (!x) return h->fail(RESNMTF_ERR_INVALID, "x is NULL");
  ViewState& vs = h->views[v];
  if (!vs.owned) return h->fail(RESNMTF_ERR_STATE, "set_view on a view this handle does not own");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  const size_t count = (size_t)vs.n * vs.m;
  double* staging = nullptr;
  double* partial = nullptr;
  const dim3 grid(ceil_div(vs.n, 32), ceil_div(vs.m, 32));
  const int nparts = grid.x * grid.y;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&staging), count * sizeof(double)));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&partial), (size_t)nparts * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(staging); return h->fail_hip("hipMalloc partial", e); }
  e = hipMemcpyAsync(staging, x, count * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(vs.X32, 0, (size_t)vs.n_pad * vs.m_pad * sizeof(float), h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(vs.Xt32, 0, (size_t)vs.m_pad * vs.n_pad * sizeof(float), h->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(convert_x_kernel, grid, dim3(256), 0, h->stream, staging, vs.n, vs.m, vs.X32, vs.m_pad,
                       vs.Xt32, vs.n_pad, partial);
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, h->stream, partial, nparts, vs.xnorm2);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(staging);
  (void)hipFree(partial);
  if (e != hipSuccess) return h->fail_hip("set_view", e);
  vs.has_x = true;
  return RESNMTF_OK;
}

int resnmtf_set_factors(resnmtf_handle* h, int v, const double* F, const double* S, const double* G,
                        const double* lambda, const double* mu) {
  if (int rc = check_view(h, v)) return rc;
  if (!F || !S || !G) return h->fail(RESNMTF_ERR_INVALID, "F, S and G are required");
  ViewState& vs = h->views[v];
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
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
    hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(vs.n * vs.k, 256)), dim3(256), 0, h->stream, vs.F, vs.n,
                       vs.k, vs.F32);
    hipLaunchKernelGGL(factor_to_f32_kernel, dim3(ceil_div(vs.m * vs.k, 256)), dim3(256), 0, h->stream, vs.G, vs.m,
                       vs.k, vs.G32);
    HIP_TRY(h, hipGetLastError());
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // host vectors go out of scope
  vs.has_factors = true;
  return RESNMTF_OK;
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
  return RESNMTF_OK;
}

int resnmtf_set_shared_rows(resnmtf_handle* h, int v, int w, int count, const int* idx_v, const int* idx_w) {
  return set_shared(h, v, w, count, idx_v, idx_w, true);
}
int resnmtf_set_shared_cols(resnmtf_handle* h, int v, int w, int count, const int* idx_v, const int* idx_w) {
  return set_shared(h, v, w, count, idx_v, idx_w, false);
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
    if (!vs.owned) continue;
    if (!vs.has_x) return h->fail(RESNMTF_ERR_STATE, "set_view missing for an owned view");
    // --- streaming passes
    PassArgs& xg = vs.passXG;
    xg.A0 = vs.Xt32; xg.lda0 = vs.n_pad; xg.ntiles0 = vs.n_pad / 64;
    xg.A1 = vs.G32; xg.A2 = vs.T32; xg.B = vs.G32; xg.P = vs.Pxg;
    xg.cols_total = vs.cols_total_xg; xg.rows_pad = vs.m_pad; xg.rows_per_split = vs.rps_xg; xg.ctl = h->ctl;
    PassArgs& xt = vs.passXtF;
    xt.A0 = vs.X32; xt.lda0 = vs.m_pad; xt.ntiles0 = vs.m_pad / 64;
    xt.A1 = vs.F32; xt.A2 = nullptr; xt.B = vs.F32; xt.P = vs.Pxtf;
    xt.cols_total = vs.cols_total_xtf; xt.rows_pad = vs.n_pad; xt.rows_per_split = vs.rps_xtf; xt.ctl = h->ctl;
    // --- F update (R/update_steps.r:141-165)
    UpdateArgs& f = vs.argF;
    f = UpdateArgs{};
    f.len = vs.n; f.k = vs.k; f.W = vs.F; f.W32 = vs.F32; f.T32 = nullptr;
    f.P = vs.Pxg; f.nsplit = vs.nsplit_xg; f.cols_total = vs.cols_total_xg; f.gram_col0 = 0;
    f.Ma_in = vs.Ma_F; f.Md_in = vs.Md_F; f.S = vs.S; f.lm = vs.lambda; f.gram_out = nullptr;
    f.colsum_part = vs.colsumF_part; f.rows_per_block = vs.rpbF; f.ctl = h->ctl;
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
        c.W = h->views[i].F; c.map = mp.dev; c.weight = wgt; c.n_other = (double)h->views[i].n;
      }
    }
    // --- G update (R/update_steps.r:180-207); branch on the WHOLE psi matrix (:190)
    UpdateArgs& g = vs.argG;
    g = UpdateArgs{};
    g.len = vs.m; g.k = vs.k; g.W = vs.G; g.W32 = vs.G32; g.T32 = vs.T32;
    g.P = vs.Pxtf; g.nsplit = vs.nsplit_xtf; g.cols_total = vs.cols_total_xtf; g.gram_col0 = vs.m_pad;
    g.Ma_in = nullptr; g.Md_in = nullptr; g.S = vs.S; g.lm = vs.mu; g.gram_out = vs.FtF;
    g.colsum_part = vs.colsumG_part; g.rows_per_block = vs.rpbG; g.ctl = h->ctl;
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
        c.W = h->views[i].G; c.map = mp.dev; c.weight = wgt; c.n_other = (double)h->views[i].m;
      }
    }
    // --- S update (R/update_steps.r:220-240); branch on the WHOLE xi matrix (:226)
    SArgs& s = vs.argS;
    s = SArgs{};
    s.k = vs.k; s.mode = 1;
    s.Pxg = vs.Pxg; s.nsplit_xg = vs.nsplit_xg; s.cols_total_xg = vs.cols_total_xg;
    s.gram_col0 = vs.n_pad; s.cross_col0 = vs.n_pad + 64;
    s.FtF = vs.FtF; s.S = vs.S; s.lambda = vs.lambda; s.mu = vs.mu;
    s.colsumF_part = vs.colsumF_part; s.nblkF = vs.nblkF; s.colsumG_part = vs.colsumG_part; s.nblkG = vs.nblkG;
    s.Ma_F = vs.Ma_F; s.Md_F = vs.Md_F; s.xnorm2 = vs.xnorm2;
    s.err = h->err; s.err_stride = V; s.err_col = v; s.err_cap = h->err_cap; s.ctl = h->ctl;
    {
      double sigma = 0.0;
      for (int i = 0; i < V; ++i) sigma += h->xi[(size_t)i + (size_t)v * V];      // sum(xi[, v])  (:231,:233)
      s.restricted = (sum_xi != 0.0) ? 1 : 0;
      s.sigma = sigma;
      s.n_couple = 0;
      for (int i = 0; i < V && s.restricted; ++i) {
        const double wgt = h->xi[(size_t)i + (size_t)v * V];
        if (wgt == 0.0 || i == v) continue;                                        // utils.r:42
        if (h->views[i].k != vs.k) return h->fail(RESNMTF_ERR_INVALID, "xi-coupled views need equal k");
        SCouple& c = s.couple[s.n_couple++];
        c.S = h->views[i].S; c.weight = wgt;
      }
    }
  }
  return RESNMTF_OK;
}

int resnmtf_prepare(resnmtf_handle* h) {
  if (!h) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  if (!h->prepared) {
    destroy_graphs(h);
    if (int rc = build_args(h)) return rc;
    h->prepared = true;
  }
  // run prologue: the first X.G pass and the F coefficients of every owned view
  SweepCtl zero{};
  HIP_TRY(h, hipMemcpyAsync(h->ctl, &zero, sizeof(zero), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // `zero` is a stack object
  for (const auto& v : h->views) {
    if (!v.owned) continue;
    launch_pass(h, v, true);
    launch_s(h, v, 0, 0, false);
  }
  HIP_TRY(h, hipGetLastError());
  return RESNMTF_OK;
}

int resnmtf_phase(resnmtf_handle* h, int v, int phase, int sweep) {
  if (int rc = check_view(h, v)) return rc;
  if (!h->prepared) return h->fail(RESNMTF_ERR_STATE, "resnmtf_prepare has not been called");
  const ViewState& vs = h->views[v];
  if (!vs.owned) return h->fail(RESNMTF_ERR_STATE, "phase on a view this handle does not own");
  if (sweep < 0) return h->fail(RESNMTF_ERR_INVALID, "negative sweep index");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  switch (phase) {
    case RESNMTF_PHASE_F: enqueue_phase_f(h, vs); break;
    case RESNMTF_PHASE_G: enqueue_phase_g(h, vs); break;
    case RESNMTF_PHASE_S: enqueue_phase_s(h, vs, sweep, false); break;
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
  if (n_iters < 0) return h->fail(RESNMTF_ERR_INVALID, "n_iters must be >= 0");
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
  if (int rc = resnmtf_prepare(h)) return rc;
  const bool eager = !h->opt.use_graph || h->opt.time_kernels;
  const int batch = std::max(1, h->opt.check_every);
  if (!eager && (h->graph_tol != tol_arg || !h->graph_one)) {
    destroy_graphs(h);
    if (int rc = capture_graph(h, 1, tol_arg, &h->graph_one)) return rc;
    if (int rc = capture_graph(h, batch, tol_arg, &h->graph_multi)) return rc;
    h->graph_multi_sweeps = batch;
    h->graph_tol = tol_arg;
  }
  int done_total = 0;
  bool converged = false;
  std::vector<double> host_err;
  while (done_total < total && !converged) {
    const int chunk = std::min(total - done_total, h->err_cap);
    // restart the chunk-local sweep counter, keep prev_mean / done
    int zero = 0;
    HIP_TRY(h, hipMemcpyAsync(&h->ctl->sweep, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int enq = 0;
    SweepCtl host_ctl{};
    while (enq < chunk) {
      int todo = std::min(batch, chunk - enq);
      if (eager) {
        for (int s = 0; s < todo; ++s) {
          enqueue_sweep(h, tol_arg);
          if (h->opt.time_kernels && h->ev_used + 8 * (size_t)h->V > h->ev.size())
            if (int rc = flush_timing(h)) return rc;
        }
      } else if (todo == h->graph_multi_sweeps) {
        HIP_TRY(h, hipGraphLaunch(h->graph_multi, h->stream));
      } else {
        for (int s = 0; s < todo; ++s) HIP_TRY(h, hipGraphLaunch(h->graph_one, h->stream));
      }
      enq += todo;
      if (tol_arg >= 0.0) {   // convergence mode: host check between batches
        HIP_TRY(h, hipMemcpyAsync(&host_ctl, h->ctl, sizeof(SweepCtl), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (host_ctl.done) { converged = true; break; }
      }
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(&host_ctl, h->ctl, sizeof(SweepCtl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int done_chunk = host_ctl.sweep;
    if (all_err && done_chunk > 0)
      HIP_TRY(h, hipMemcpy(all_err + done_total, h->mean_err, (size_t)done_chunk * sizeof(double), hipMemcpyDeviceToHost));
    done_total += done_chunk;
    if (host_ctl.done) converged = true;
    if (done_chunk < chunk && !converged) return h->fail(RESNMTF_ERR_HIP, "sweep counter mismatch");
  }
  if (h->opt.time_kernels)
    if (int rc = flush_timing(h)) return rc;
  if (iters_done) *iters_done = done_total;
  return RESNMTF_OK;
}

int resnmtf_get_factors(resnmtf_handle* h, int v, double* F, double* S, double* G, double* lambda, double* mu) {
  if (int rc = check_view(h, v)) return rc;
  ViewState& vs = h->views[v];
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
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
    default: return h->fail(RESNMTF_ERR_INVALID, "unknown factor selector");
  }
  return RESNMTF_OK;
}

int resnmtf_view_errors(resnmtf_handle* h, int v, int first, int count, double* out) {
  if (int rc = check_view(h, v)) return rc;
  if (!out || first < 0 || count < 0) return h->fail(RESNMTF_ERR_INVALID, "bad error range");
  if (count > h->err_cap) return h->fail(RESNMTF_ERR_INVALID, "range longer than the error ring");
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<double> ring((size_t)h->err_cap * h->V);
  HIP_TRY(h, hipMemcpy(ring.data(), h->err, ring.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int t = 0; t < count; ++t) out[t] = ring[(size_t)((first + t) % h->err_cap) * h->V + v];
  return RESNMTF_OK;
}

int resnmtf_synchronize(resnmtf_handle* h) {
  if (!h) return RESNMTF_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->opt.device_id));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
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
    out->xg_bytes = 4.0 * n * m + 4.0 * (n + m) * k;
    out->xtf_bytes = 4.0 * n * m + 4.0 * (n + m) * k;
    out->xg_flops = 2.0 * n * m * k;
    out->xtf_flops = 2.0 * n * m * k;
    break;
  }
  if (reset) h->timing = resnmtf_pass_timing{};
  return RESNMTF_OK;
}

}  // extern "C"
