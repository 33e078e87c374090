// det.hip -- deterministic mode of the column reductions (include/pdae.h: pdae_set_deterministic).
//
// The reference's own reductions (cuDNN batch-norm statistics, cuBLAS split-K, ATen's
// layer_norm backward, models/PointCAE_transformer.py:37-51 / 94-147) make no ordering promise,
// and neither does this library by default: per-block partial sums meet in device-scope float
// atomics.  With a workspace registered, every such site stores its block partials with plain
// stores and one ordered pass adds them -- bit-identical results run to run and graph replay
// to eager launch, at the price of one small launch per reduction.
#include "common.h"

namespace pdae {

// ---- contexts (include/pdae.h: pdae_ctx_create / pdae_ctx_set_current) -----------------------------------------
// Everything mutable the library keeps between calls lives in a Ctx: the deterministic-mode workspace, the parked
// (deferred) reductions, the GEMM arithmetic.  A host thread works on its current context (thread local; the process
// default context until it sets one), so two threads driving two streams -- each with its own context and workspaces --
// share nothing.
constexpr int DEF_MAX = 48;
struct DefJob {
  const float* part;
  float* out[3];
  int n[3];
  int P, width;
};
struct DefJobs {
  DefJob j[DEF_MAX];
};
struct Ctx {
  void* ws = nullptr;                 // deterministic mode
  size_t ws_bytes = 0;
  float* def_ws = nullptr;            // deferred reductions
  size_t def_floats = 0, def_used = 0;
  bool def_on = false, def_hold = false;
  DefJobs def_jobs;
  int def_n = 0;
  int arith = -1;                     // GEMM arithmetic (-1: PDAE_GEMM in the environment decides at first use)
};
static Ctx g_default_ctx;
static thread_local Ctx* t_ctx = nullptr;
static Ctx& ctx() { return t_ctx ? *t_ctx : g_default_ctx; }

int& ctx_gemm_arith() { return ctx().arith; }

bool det_on() { return ctx().ws != nullptr; }

void* det_workspace(size_t bytes, int* rc) {
  *rc = PDAE_OK;
  Ctx& c = ctx();
  if (!c.ws) return nullptr;
  if (bytes > c.ws_bytes) {
    *rc = unsupported("deterministic mode: the registered workspace is too small for this reduction");
    return nullptr;
  }
  return c.ws;
}

template <typename T>
__global__ __launch_bounds__(256) void det_reduce_kernel(int P, int width, const T* __restrict__ part,
                                                         T* __restrict__ o0, int n0, T* __restrict__ o1,
                                                         int n1, T* __restrict__ o2, int n2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  T* out = c < n0 ? (o0 ? o0 + c : nullptr)
                  : (c < n0 + n1 ? (o1 ? o1 + (c - n0) : nullptr) : (o2 ? o2 + (c - n0 - n1) : nullptr));
  if (!out) return;
  // eight independent loads in flight, added in partition order (in fp64: hundreds of partials of mixed sign)
  double t = 0;
  int p = 0;
  for (; p + 8 <= P; p += 8) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(p + k) * width + c];
#pragma unroll
    for (int k = 0; k < 8; ++k) t += v[k];
  }
  for (; p < P; ++p) t += part[(size_t)p * width + c];
  *out += (T)t;
}

int det_reduce(hipStream_t s, int P, int width, const float* part, float* o0, int n0, float* o1, int n1,
               float* o2, int n2) {
  if (P <= 0 || width <= 0) return PDAE_OK;
  hipLaunchKernelGGL(det_reduce_kernel<float>, dim3((width + 255) / 256), dim3(256), 0, s, P, width, part, o0,
                     n0, o1, n1, o2, n2);
  return check_launch("det_reduce");
}

int det_reduce_f64(hipStream_t s, int P, int width, const double* part, double* out) {
  if (P <= 0 || width <= 0) return PDAE_OK;
  hipLaunchKernelGGL(det_reduce_kernel<double>, dim3((width + 255) / 256), dim3(256), 0, s, P, width, part,
                     out, width, (double*)nullptr, 0, (double*)nullptr, 0);
  return check_launch("det_reduce_f64");
}

// ---- deferred reductions ---------------------------------------------------------------------
bool deferred_on() { return ctx().def_on && !ctx().def_hold; }

float* deferred_take(int P, int width, float* o0, int n0, float* o1, int n1, float* o2, int n2) {
  Ctx& c = ctx();
  if (!deferred_on() || c.def_n >= DEF_MAX) return nullptr;
  const size_t need = ((size_t)P * width + 63) & ~(size_t)63;
  if (c.def_used + need > c.def_floats) return nullptr;
  float* part = c.def_ws + c.def_used;
  c.def_used += need;
  DefJob& d = c.def_jobs.j[c.def_n++];
  d.part = part, d.P = P, d.width = width;
  d.out[0] = o0, d.out[1] = o1, d.out[2] = o2;
  d.n[0] = n0, d.n[1] = n1, d.n[2] = n2;
  return part;
}

// grid (column chunks of 32, jobs); block = 32 columns x 8 partial lanes: lane l adds partials l, l + 8, ...
// (eight loads in flight), the lanes are added in lane order, out[c] += the sum (fixed order => reproducible).
__global__ __launch_bounds__(256) void deferred_reduce_kernel(const DefJobs jobs) {
  const DefJob& d = jobs.j[blockIdx.y];
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float t = 0.f;
  if (c < d.width)
    for (int p = l; p < d.P; p += 64) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p + 8 * k < d.P ? d.part[(size_t)(p + 8 * k) * d.width + c] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
  red[l][cl] = t;
  __syncthreads();
  if (l == 0 && c < d.width) {
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][cl];
    float* out = c < d.n[0] ? (d.out[0] ? d.out[0] + c : nullptr)
                            : (c < d.n[0] + d.n[1] ? (d.out[1] ? d.out[1] + (c - d.n[0]) : nullptr)
                                                   : (d.out[2] ? d.out[2] + (c - d.n[0] - d.n[1]) : nullptr));
    if (out) *out += t;
  }
}

}  // namespace pdae

extern "C" int pdae_set_deterministic(void* workspace, size_t bytes) {
  if (workspace && bytes < (1u << 20)) return pdae::bad_arg("set_deterministic: workspace of at least 1 MiB");
  pdae::ctx().ws = workspace;
  pdae::ctx().ws_bytes = workspace ? bytes : 0;
  return PDAE_OK;
}

extern "C" int pdae_deterministic(void) { return pdae::det_on() ? 1 : 0; }

extern "C" int pdae_deferred_begin(void* workspace, size_t bytes) {
  if (!workspace || bytes < (1u << 20)) return pdae::bad_arg("deferred_begin: workspace of at least 1 MiB");
  pdae::Ctx& c = pdae::ctx();
  c.def_ws = static_cast<float*>(workspace);
  c.def_floats = bytes / sizeof(float);
  c.def_used = 0, c.def_n = 0, c.def_on = true, c.def_hold = false;
  return PDAE_OK;
}

extern "C" int pdae_deferred_flush(pdae_stream_t stream) {
  using namespace pdae;
  Ctx& c = ctx();
  c.def_on = false;
  const int rc = rows_wgrad_flush(as_stream(stream));
  if (rc) return rc;
  if (c.def_n == 0) return PDAE_OK;
  int widest = 0;
  for (int i = 0; i < c.def_n; ++i) widest = c.def_jobs.j[i].width > widest ? c.def_jobs.j[i].width : widest;
  hipLaunchKernelGGL(deferred_reduce_kernel, dim3((widest + 31) / 32, c.def_n), dim3(256), 0, as_stream(stream),
                     c.def_jobs);
  c.def_n = 0;
  return check_launch("deferred_flush");
}

extern "C" int pdae_deferred_hold(int hold) {
  pdae::ctx().def_hold = hold != 0;
  return PDAE_OK;
}

extern "C" int pdae_ctx_create(pdae_ctx_t* out) {
  if (!out) return pdae::bad_arg("ctx_create: null pointer");
  *out = reinterpret_cast<pdae_ctx_t>(new pdae::Ctx());
  return PDAE_OK;
}

extern "C" int pdae_ctx_destroy(pdae_ctx_t c) {
  pdae::Ctx* p = reinterpret_cast<pdae::Ctx*>(c);
  if (!p) return pdae::bad_arg("ctx_destroy: null context");
  if (p == pdae::t_ctx) return pdae::bad_arg("ctx_destroy: the context is current on this thread");
  delete p;
  return PDAE_OK;
}

extern "C" int pdae_ctx_set_current(pdae_ctx_t c) {
  pdae::t_ctx = reinterpret_cast<pdae::Ctx*>(c);
  return PDAE_OK;
}

extern "C" pdae_ctx_t pdae_ctx_current(void) { return reinterpret_cast<pdae_ctx_t>(pdae::t_ctx); }
