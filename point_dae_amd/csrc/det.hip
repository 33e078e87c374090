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

static void* g_ws = nullptr;
static size_t g_ws_bytes = 0;

bool det_on() { return g_ws != nullptr; }

void* det_workspace(size_t bytes, int* rc) {
  *rc = PDAE_OK;
  if (!g_ws) return nullptr;
  if (bytes > g_ws_bytes) {
    *rc = unsupported("deterministic mode: the registered workspace is too small for this reduction");
    return nullptr;
  }
  return g_ws;
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
  // eight independent loads in flight, added in partition order
  T t = 0;
  int p = 0;
  for (; p + 8 <= P; p += 8) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(p + k) * width + c];
#pragma unroll
    for (int k = 0; k < 8; ++k) t += v[k];
  }
  for (; p < P; ++p) t += part[(size_t)p * width + c];
  *out += t;
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

}  // namespace pdae

extern "C" int pdae_set_deterministic(void* workspace, size_t bytes) {
  if (workspace && bytes < (1u << 20)) return pdae::bad_arg("set_deterministic: workspace of at least 1 MiB");
  pdae::g_ws = workspace;
  pdae::g_ws_bytes = workspace ? bytes : 0;
  return PDAE_OK;
}

extern "C" int pdae_deterministic(void) { return pdae::det_on() ? 1 : 0; }
