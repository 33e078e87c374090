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

// ---- BatchNorm + ReLU backward THROUGH the max-pool (the last layer of a level) ----------------
// The gradient of a = relu(bn(y)) is non-zero only at the arg-max row of every (group, channel), so
// the dense form (scatter G*ns x C, then two sweeps over it) reads and writes mostly zeros.  Here
//   pool_bn_reduce : S1[c] = sum_g t, S2[c] = sum_g t * xhat(y[g*ns + arg][c]),  t = grad[g][c] * (a > 0)
//                    -- reads grad / arg / out and gathers ONE y element per (g, c); block partials
//                    in fixed order, then plain stores of the per-block sums (no atomics)
//   pool_bn_apply  : dy[g*ns + j][c] = gamma*invstd * ((j == arg) ? t : 0 - S1/R - xhat * S2/R)
//                    -- one read of y, one write of dy
namespace pdae {

constexpr int PB_GROUPS = 256;   // groups per block of the reduction

__global__ __launch_bounds__(256) void pool_bn_reduce_kernel(long long G, int ns, int C4,
                                                             const float4* __restrict__ grad,
                                                             const uchar4* __restrict__ arg,
                                                             const float4* __restrict__ out,
                                                             const float* __restrict__ y,
                                                             const float4* __restrict__ mean,
                                                             const float4* __restrict__ invstd,
                                                             float4* __restrict__ part) {
  extern __shared__ float4 pb_red[];                 // [phases][2][C4]
  const int PH = 256 / C4;
  const int q = threadIdx.x % C4, ph = threadIdx.x / C4;
  const long long g0 = (long long)blockIdx.x * PB_GROUPS;
  const long long g1 = g0 + PB_GROUPS < G ? g0 + PB_GROUPS : G;
  const float4 mu = mean[q], is = invstd[q];
  const int C = C4 * 4;
  // the sums run over up to a million rows of mixed sign (the reference's ATen kernel accumulates them in double on the
  // CPU path the oracle runs): per-thread chains in fp64, the few partials per block / per launch then in fp32 trees
  double s1x = 0., s1y = 0., s1z = 0., s1w = 0., s2x = 0., s2y = 0., s2z = 0., s2w = 0.;
  for (long long g = g0 + ph; g < g1; g += PH) {
    const float4 d = grad[g * C4 + q];
    const float4 o = out[g * C4 + q];                // the pooled relu(bn(y)): > 0 <=> the ReLU passed
    const uchar4 a = arg[g * C4 + q];
    const float* base = y + (size_t)g * ns * C + q * 4;
    const float yx = base[(size_t)a.x * C + 0], yy = base[(size_t)a.y * C + 1];
    const float yz = base[(size_t)a.z * C + 2], yw = base[(size_t)a.w * C + 3];
    const float tx = o.x > 0.f ? d.x : 0.f, ty = o.y > 0.f ? d.y : 0.f;
    const float tz = o.z > 0.f ? d.z : 0.f, tw = o.w > 0.f ? d.w : 0.f;
    s1x += tx, s1y += ty, s1z += tz, s1w += tw;
    s2x += tx * ((yx - mu.x) * is.x), s2y += ty * ((yy - mu.y) * is.y);
    s2z += tz * ((yz - mu.z) * is.z), s2w += tw * ((yw - mu.w) * is.w);
  }
  float4 s1 = make_float4((float)s1x, (float)s1y, (float)s1z, (float)s1w);
  float4 s2 = make_float4((float)s2x, (float)s2y, (float)s2z, (float)s2w);
  pb_red[(ph * 2 + 0) * C4 + q] = s1;
  pb_red[(ph * 2 + 1) * C4 + q] = s2;
  __syncthreads();
  if (ph == 0) {
    for (int k = 1; k < PH; ++k) {
      const float4 u = pb_red[(k * 2 + 0) * C4 + q], v = pb_red[(k * 2 + 1) * C4 + q];
      s1.x += u.x, s1.y += u.y, s1.z += u.z, s1.w += u.w;
      s2.x += v.x, s2.y += v.y, s2.z += v.z, s2.w += v.w;
    }
    part[((size_t)blockIdx.x * 2 + 0) * C4 + q] = s1;
    part[((size_t)blockIdx.x * 2 + 1) * C4 + q] = s2;
  }
}

// S[2][C] = sum of the P partial sets, in order
__global__ __launch_bounds__(256) void pool_bn_finish_kernel(int P, int n, const float* __restrict__ part,
                                                             float* __restrict__ S) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double t = 0.;
  for (int p = 0; p < P; ++p) t += part[(size_t)p * n + i];
  S[i] = (float)t;
}

__global__ __launch_bounds__(256) void pool_bn_apply_kernel(long long G, int ns, int C4, float inv_rows,
                                                            const float4* __restrict__ grad,
                                                            const uchar4* __restrict__ arg,
                                                            const float4* __restrict__ out,
                                                            const float4* __restrict__ y,
                                                            const float4* __restrict__ mean,
                                                            const float4* __restrict__ invstd,
                                                            const float4* __restrict__ gamma,
                                                            const float4* __restrict__ S,
                                                            float4* __restrict__ dy) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= G * C4) return;
  const long long g = i / C4;
  const int q = (int)(i - g * C4);
  const float4 mu = mean[q], is = invstd[q], ga = gamma[q], s1 = S[q], s2 = S[C4 + q];
  const float4 d = grad[i], o = out[i];
  const uchar4 a = arg[i];
  const float tx = o.x > 0.f ? d.x : 0.f, ty = o.y > 0.f ? d.y : 0.f;
  const float tz = o.z > 0.f ? d.z : 0.f, tw = o.w > 0.f ? d.w : 0.f;
  const float kx = ga.x * is.x, ky = ga.y * is.y, kz = ga.z * is.z, kw = ga.w * is.w;
  const float m1x = s1.x * inv_rows, m1y = s1.y * inv_rows, m1z = s1.z * inv_rows, m1w = s1.w * inv_rows;
  const float m2x = s2.x * inv_rows, m2y = s2.y * inv_rows, m2z = s2.z * inv_rows, m2w = s2.w * inv_rows;
  const float4* row = y + g * ns * C4 + q;
  float4* drow = dy + g * ns * C4 + q;
#pragma unroll 8
  for (int j = 0; j < ns; ++j) {
    const float4 v = row[(long long)j * C4];
    float4 r;
    r.x = kx * ((a.x == j ? tx : 0.f) - m1x - ((v.x - mu.x) * is.x) * m2x);
    r.y = ky * ((a.y == j ? ty : 0.f) - m1y - ((v.y - mu.y) * is.y) * m2y);
    r.z = kz * ((a.z == j ? tz : 0.f) - m1z - ((v.z - mu.z) * is.z) * m2z);
    r.w = kw * ((a.w == j ? tw : 0.f) - m1w - ((v.w - mu.w) * is.w) * m2w);
    drow[(long long)j * C4] = r;
  }
}

}  // namespace pdae

extern "C" long long pdae_pool_bn_backward_workspace(long long G, int C) {
  return ((G + pdae::PB_GROUPS - 1) / pdae::PB_GROUPS) * 2 * (long long)C;
}

extern "C" int pdae_pool_bn_backward(long long G, int ns, int C, const float* grad, const unsigned char* arg,
                                     const float* out, const float* y, const float* mean, const float* invstd,
                                     const float* gamma, float* S, float* workspace, float* dy,
                                     pdae_stream_t stream) {
  int rc = check_sa("pool_bn_backward: 1 <= nsample <= 256, C a positive multiple of 4", G, ns, C);
  if (rc) return rc;
  if (256 % (C / 4) != 0 || C > 1024) return unsupported("pool_bn_backward: C/4 must divide 256");
  if (!S) return bad_arg("pool_bn_backward: null pointer");
  hipStream_t s = as_stream(stream);
  if (G == 0) {
    (void)hipMemsetAsync(S, 0, sizeof(float) * 2 * (size_t)C, s);
    return check_launch("pool_bn_backward");
  }
  if (!grad || !arg || !out || !y || !mean || !invstd || !gamma || !workspace || !dy)
    return bad_arg("pool_bn_backward: null pointer");
  const int C4 = C / 4;
  const int P = (int)((G + PB_GROUPS - 1) / PB_GROUPS);
  hipLaunchKernelGGL(pool_bn_reduce_kernel, dim3(P), dim3(256), sizeof(float4) * 2 * 256, s, G, ns, C4,
                     reinterpret_cast<const float4*>(grad), reinterpret_cast<const uchar4*>(arg),
                     reinterpret_cast<const float4*>(out), y, reinterpret_cast<const float4*>(mean),
                     reinterpret_cast<const float4*>(invstd), reinterpret_cast<float4*>(workspace));
  hipLaunchKernelGGL(pool_bn_finish_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, P, 2 * C, workspace, S);
  const long long n = G * C4;
  hipLaunchKernelGGL(pool_bn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, G, ns, C4,
                     1.0f / (float)(G * ns), reinterpret_cast<const float4*>(grad),
                     reinterpret_cast<const uchar4*>(arg), reinterpret_cast<const float4*>(out),
                     reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(mean),
                     reinterpret_cast<const float4*>(invstd), reinterpret_cast<const float4*>(gamma),
                     reinterpret_cast<const float4*>(S), reinterpret_cast<float4*>(dy));
  return check_launch("pool_bn_backward");
}
