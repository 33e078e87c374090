// interpolate.hip -- feature propagation operators of PointNet++: three_nn, three_interpolate and its
// gradient (reference: extensions/pointnet2/_ext_src/src/interpolate_gpu.cu:12-146, host
// interpolate.cpp; Python wrappers extensions/pointnet2/pointnet2_utils.py:118-204).
//
// The reference launches ONE block per cloud: every thread walks all m known points through global
// memory for its unknown point (three_nn), and the gradient is one global atomicAdd per (channel,
// point, neighbour).  Here:
//   three_nn             256 unknown points per block, the known cloud streamed through LDS in SoA
//                        tiles (broadcast reads), the reference's insertion order and strict `<`
//                        comparisons kept, so distances and indices are bit-identical to it
//   three_interpolate    one thread per output element, p1*w1 + p2*w2 + p3*w3 evaluated left to
//                        right with every operation rounded (the arithmetic contract of pdae.h)
//   three_interpolate_grad  one block per (cloud, channel) row: the scatter-add runs on LDS float
//                        atomics over the row's m slots and the row is written once -- no global
//                        atomics, no zero-fill pass
#include "common.h"

namespace pdae {

constexpr int TNN_TILE = 1024;

__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float* __restrict__ unknown,
                                                       const float* __restrict__ known,
                                                       float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float kx[TNN_TILE], ky[TNN_TILE], kz[TNN_TILE];
  const int b = blockIdx.y;
  unknown += (size_t)b * n * 3, known += (size_t)b * m * 3;
  dist2 += (size_t)b * n * 3, idx += (size_t)b * n * 3;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool on = j < n;
  const float ux = on ? unknown[j * 3 + 0] : 0.f, uy = on ? unknown[j * 3 + 1] : 0.f, uz = on ? unknown[j * 3 + 2] : 0.f;
  // the reference keeps the running bests in double (initial 1e40) and compares the float distance with them
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += TNN_TILE) {
    const int cnt = min(TNN_TILE, m - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      kx[t] = known[(size_t)(k0 + t) * 3 + 0];
      ky[t] = known[(size_t)(k0 + t) * 3 + 1];
      kz[t] = known[(size_t)(k0 + t) * 3 + 2];
    }
    __syncthreads();
    for (int t = 0; t < cnt; ++t) {
      const float x = kx[t], y = ky[t], z = kz[t];
      const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      const int k = k0 + t;
      if (d < best1) {
        best3 = best2, besti3 = besti2;
        best2 = best1, besti2 = besti1;
        best1 = d, besti1 = k;
      } else if (d < best2) {
        best3 = best2, besti3 = besti2;
        best2 = d, besti2 = k;
      } else if (d < best3) {
        best3 = d, besti3 = k;
      }
    }
  }
  if (on) {
    dist2[j * 3 + 0] = (float)best1, dist2[j * 3 + 1] = (float)best2, dist2[j * 3 + 2] = (float)best3;
    idx[j * 3 + 0] = besti1, idx[j * 3 + 1] = besti2, idx[j * 3 + 2] = besti3;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                                const float* __restrict__ points,
                                                                const int* __restrict__ idx,
                                                                const float* __restrict__ weight,
                                                                float* __restrict__ out) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)c * n) return;
  const int l = (int)(i / n), j = (int)(i % n);
  const float* w = weight + ((size_t)b * n + j) * 3;
  const int* id = idx + ((size_t)b * n + j) * 3;
  const float* row = points + ((size_t)b * c + l) * m;
  out[(size_t)b * c * n + i] = row[id[0]] * w[0] + row[id[1]] * w[1] + row[id[2]] * w[2];
}

// one block per (cloud, channel): grad_points[b, l, :] accumulated in LDS, written once
__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(int c, int n, int m,
                                                                     const float* __restrict__ grad_out,
                                                                     const int* __restrict__ idx,
                                                                     const float* __restrict__ weight,
                                                                     float* __restrict__ grad_points) {
  extern __shared__ float acc[];   // [m]
  const int b = blockIdx.y, l = blockIdx.x;
  for (int t = threadIdx.x; t < m; t += 256) acc[t] = 0.f;
  __syncthreads();
  const float* g = grad_out + ((size_t)b * c + l) * n;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float* w = weight + ((size_t)b * n + j) * 3;
    const int* id = idx + ((size_t)b * n + j) * 3;
    const float v = g[j];
    atomicAdd(&acc[id[0]], v * w[0]);
    atomicAdd(&acc[id[1]], v * w[1]);
    atomicAdd(&acc[id[2]], v * w[2]);
  }
  __syncthreads();
  float* dst = grad_points + ((size_t)b * c + l) * m;
  for (int t = threadIdx.x; t < m; t += 256) dst[t] = acc[t];
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                             int32_t* idx, pdae_stream_t stream) {
  if (b < 0 || n < 0 || m < 0) return bad_arg("three_nn: negative size");
  if (b == 0 || n == 0) return PDAE_OK;
  if (!unknown || !dist2 || !idx || (m > 0 && !known)) return bad_arg("three_nn: null pointer");
  if (b > 65535) return unsupported("three_nn: more than 65535 clouds");
  hipLaunchKernelGGL(three_nn_kernel, dim3((n + 255) / 256, b), dim3(256), 0, as_stream(stream), n, m, unknown,
                     known, dist2, idx);
  return check_launch("three_nn");
}

extern "C" int pdae_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx,
                                      const float* weight, float* out, pdae_stream_t stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return bad_arg("three_interpolate: negative size");
  if (b == 0 || c == 0 || n == 0) return PDAE_OK;
  if (m == 0) return bad_arg("three_interpolate: no known points");
  if (!points || !idx || !weight || !out) return bad_arg("three_interpolate: null pointer");
  if (b > 65535) return unsupported("three_interpolate: more than 65535 clouds");
  const long long total = (long long)c * n;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3((unsigned)((total + 255) / 256), b), dim3(256), 0,
                     as_stream(stream), c, m, n, points, idx, weight, out);
  return check_launch("three_interpolate");
}

extern "C" int pdae_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                           const int32_t* idx, const float* weight, float* grad_points,
                                           pdae_stream_t stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return bad_arg("three_interpolate_grad: negative size");
  if (b == 0 || c == 0 || m == 0) return PDAE_OK;
  if (!grad_points || (n > 0 && (!grad_out || !idx || !weight))) return bad_arg("three_interpolate_grad: null pointer");
  if (b > 65535 || c > 65535) return unsupported("three_interpolate_grad: more than 65535 clouds or channels");
  if ((size_t)m * sizeof(float) > 160 * 1024) return unsupported("three_interpolate_grad: more than 40960 known points");
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(three_interpolate_grad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(c, b), dim3(256), (size_t)m * sizeof(float),
                     as_stream(stream), c, n, m, grad_out, idx, weight, grad_points);
  return check_launch("three_interpolate_grad");
}
