// set_abstraction.hip -- the max-pool end of a PointNet++ set-abstraction level
// (pointnet2_modules.PointnetSAModule: SharedMLP -> F.max_pool2d over nsample), fused with the last
// layer's BatchNorm + ReLU, and its backward scatter.
//   bnrelu_group_max   out[g][c] = max_j relu(y[g*ns + j][c] * scale[c] + shift[c]), arg = first j
//                      that attains it (the affine may have a negative scale: it is applied before
//                      the max, row by row) -- one read of y, nothing else stored
//   group_max_scatter_n  dense[g*ns + j][c] = (j == arg[g][c]) ? grad[g][c] : 0
// A thread owns four adjacent channels of one group and walks its ns rows; consecutive threads
// take consecutive channel quads, so every row is read / written in full cache lines.
#include "common.h"

namespace pdae {

__global__ __launch_bounds__(256) void bnrelu_group_max_kernel(long long G, int ns, int C4,
                                                               const float4* __restrict__ y,
                                                               const float4* __restrict__ scale,
                                                               const float4* __restrict__ shift,
                                                               float4* __restrict__ out,
                                                               uchar4* __restrict__ arg) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * C4) return;
  const long long g = i / C4;
  const int q = (int)(i - g * C4);
  const float4 sc = scale[q], sh = shift[q];
  const float4* row = y + g * ns * C4 + q;
  float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);       // relu(.) >= 0 > -1: the first row always wins
  uchar4 ba = make_uchar4(0, 0, 0, 0);
#pragma unroll 8
  for (int j = 0; j < ns; ++j) {
    const float4 v = row[(long long)j * C4];
    float4 t;
    t.x = v.x * sc.x + sh.x, t.y = v.y * sc.y + sh.y, t.z = v.z * sc.z + sh.z, t.w = v.w * sc.w + sh.w;
    t.x = t.x > 0.f ? t.x : 0.f, t.y = t.y > 0.f ? t.y : 0.f;
    t.z = t.z > 0.f ? t.z : 0.f, t.w = t.w > 0.f ? t.w : 0.f;
    if (t.x > best.x) best.x = t.x, ba.x = (unsigned char)j;
    if (t.y > best.y) best.y = t.y, ba.y = (unsigned char)j;
    if (t.z > best.z) best.z = t.z, ba.z = (unsigned char)j;
    if (t.w > best.w) best.w = t.w, ba.w = (unsigned char)j;
  }
  out[i] = best;
  arg[i] = ba;
}

__global__ __launch_bounds__(256) void group_max_scatter_n_kernel(long long G, int ns, int C4,
                                                                  const float4* __restrict__ grad,
                                                                  const uchar4* __restrict__ arg,
                                                                  float4* __restrict__ dense) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * C4) return;
  const long long g = i / C4;
  const int q = (int)(i - g * C4);
  const float4 v = grad[i];
  const uchar4 a = arg[i];
  float4* row = dense + g * ns * C4 + q;
#pragma unroll 8
  for (int j = 0; j < ns; ++j) {
    float4 o;
    o.x = a.x == j ? v.x : 0.f, o.y = a.y == j ? v.y : 0.f;
    o.z = a.z == j ? v.z : 0.f, o.w = a.w == j ? v.w : 0.f;
    row[(long long)j * C4] = o;
  }
}

}  // namespace pdae

using namespace pdae;

static int check_sa(const char* who, long long G, int ns, int C) {
  if (G < 0 || ns <= 0 || ns > 256 || C <= 0 || C % 4 != 0) return bad_arg(who);
  if ((G * (C / 4) + 255) / 256 > 0x7fffffffLL) return unsupported("set abstraction: too many groups");
  return PDAE_OK;
}

extern "C" int pdae_bnrelu_group_max(long long G, int ns, int C, const float* y, const float* scale,
                                     const float* shift, float* out, unsigned char* arg, pdae_stream_t stream) {
  int rc = check_sa("bnrelu_group_max: 1 <= nsample <= 256, C a positive multiple of 4", G, ns, C);
  if (rc || G == 0) return rc;
  if (!y || !scale || !shift || !out || !arg) return bad_arg("bnrelu_group_max: null pointer");
  const long long n = G * (C / 4);
  hipLaunchKernelGGL(bnrelu_group_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), G,
                     ns, C / 4, reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(scale),
                     reinterpret_cast<const float4*>(shift), reinterpret_cast<float4*>(out),
                     reinterpret_cast<uchar4*>(arg));
  return check_launch("bnrelu_group_max");
}

extern "C" int pdae_group_max_scatter_n(long long G, int ns, int C, const float* grad, const unsigned char* arg,
                                        float* dense, pdae_stream_t stream) {
  int rc = check_sa("group_max_scatter_n: 1 <= nsample <= 256, C a positive multiple of 4", G, ns, C);
  if (rc || G == 0) return rc;
  if (!grad || !arg || !dense) return bad_arg("group_max_scatter_n: null pointer");
  const long long n = G * (C / 4);
  hipLaunchKernelGGL(group_max_scatter_n_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     G, ns, C / 4, reinterpret_cast<const float4*>(grad), reinterpret_cast<const uchar4*>(arg),
                     reinterpret_cast<float4*>(dense));
  return check_launch("group_max_scatter_n");
}
