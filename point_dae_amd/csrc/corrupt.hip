// corrupt.hip -- in-forward affine corruption of patches and centres.
//
// Semantics: models/PointCAE_transformer.py:680-684 + the tensor corruptions
// of datasets/corrupt_util_tensor.py (:59-116 multiply, :139-342 matmul).  The
// reference runs each of the 1-3 maps as separate broadcast multiply / batched
// matmul kernels over (B,G,k,3) plus 4 add/sub passes and several small H2D
// copies; here everything is one pass: each point is read once and written as
// the ground-truth patch and the transformed patch.
#include "common.h"

namespace pdae {

__device__ __forceinline__ void apply_steps(float& x, float& y, float& z, const float* st,
                                            int nsteps, int b_stride) {
  for (int s = 0; s < nsteps; ++s) {
    const float* p = st + (size_t)s * b_stride;
    if (p[0] == 0.f) {
      x *= p[1];
      y *= p[2];
      z *= p[3];
    } else {  // row vector times matrix: out_j = sum_i v_i R[i][j], i ascending
      const float ox = x * p[1] + y * p[4] + z * p[7];
      const float oy = x * p[2] + y * p[5] + z * p[8];
      const float oz = x * p[3] + y * p[6] + z * p[9];
      x = ox;
      y = oy;
      z = oz;
    }
  }
}

__global__ __launch_bounds__(256) void patch_affine_kernel(
    int b, int g, int k, int nsteps, const float* __restrict__ nbr,
    const float* __restrict__ center, const float* __restrict__ steps, float* __restrict__ gt_nbr,
    float* __restrict__ t_nbr, float* __restrict__ t_center) {
  const long long total = (long long)b * g * k;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long grp = t / k;
  const int bi = (int)(grp / g);
  const float cx = center[grp * 3 + 0], cy = center[grp * 3 + 1], cz = center[grp * 3 + 2];
  const float ax = nbr[t * 3 + 0] + cx, ay = nbr[t * 3 + 1] + cy, az = nbr[t * 3 + 2] + cz;
  const float* st = steps + (size_t)bi * 10;
  float px = ax, py = ay, pz = az, tx = cx, ty = cy, tz = cz;
  apply_steps(px, py, pz, st, nsteps, b * 10);
  apply_steps(tx, ty, tz, st, nsteps, b * 10);
  gt_nbr[t * 3 + 0] = ax - cx;
  gt_nbr[t * 3 + 1] = ay - cy;
  gt_nbr[t * 3 + 2] = az - cz;
  t_nbr[t * 3 + 0] = px - tx;
  t_nbr[t * 3 + 1] = py - ty;
  t_nbr[t * 3 + 2] = pz - tz;
  if (t - grp * k == 0) {
    t_center[grp * 3 + 0] = tx;
    t_center[grp * 3 + 1] = ty;
    t_center[grp * 3 + 2] = tz;
  }
}

}  // namespace pdae

extern "C" int pdae_patch_affine(int b, int g, int k, int nsteps, const float* nbr,
                                 const float* center, const float* steps, float* gt_nbr,
                                 float* t_nbr, float* t_center, pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || g < 0 || k <= 0 || nsteps < 0) return bad_arg("patch_affine: bad size");
  if (b == 0 || g == 0) return PDAE_OK;
  if (!nbr || !center || !gt_nbr || !t_nbr || !t_center || (nsteps > 0 && !steps))
    return bad_arg("patch_affine: null pointer");
  const long long total = (long long)b * g * k;
  hipLaunchKernelGGL(patch_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), b, g, k, nsteps, nbr, center, steps, gt_nbr, t_nbr,
                     t_center);
  return check_launch("patch_affine");
}
