// embed.hip -- the memory-bound passes of the patch embedder's backward
// (Encoder, models/PointCAE_transformer.py:20-51) as single fused sweeps.
//
// PyTorch runs the backward of max-pool / ReLU / BatchNorm (training mode) /
// concat as ~10 separate kernels over (B*G*32, C) tensors; here:
//   group_max_scatter      max-pool backward: grad (G,C) -> dense (G*32,C)
//   group_scatter_add      adds grad (G,C) at the arg-max rows of a (G*32,C) grad
//   bnrelu_backward_reduce S1[c] = sum t, S2[c] = sum t*xhat with
//                          t = dA * (relu mask), xhat = (x - mean) * invstd
//   bnrelu_backward_apply  dX = gamma*invstd * (t - S1/R - xhat*S2/R), in place
//                          over dA, plus the per-group row sums of dX (needed by
//                          the global half of the split concat GEMM)
// All are float4-wide, one pass over each operand.
#include "common.h"

namespace pdae {

// out[g*32 + r][c] = (r == arg[g][c]) ? grad[g][c] : 0
__global__ __launch_bounds__(256) void group_max_scatter_kernel(int G, int C,
                                                                const float* __restrict__ grad,
                                                                const unsigned char* __restrict__ arg,
                                                                float* __restrict__ out) {
  const int c4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int rsub = threadIdx.x >> 6;  // 4 row phases
  const int g = blockIdx.y;
  if (c4 >= C) return;
  const float4 v = *reinterpret_cast<const float4*>(grad + (size_t)g * C + c4);
  const uchar4 a = *reinterpret_cast<const uchar4*>(arg + (size_t)g * C + c4);
  float* base = out + (size_t)g * 32 * C + c4;
#pragma unroll
  for (int r = rsub; r < 32; r += 4) {
    float4 o;
    o.x = a.x == r ? v.x : 0.f;
    o.y = a.y == r ? v.y : 0.f;
    o.z = a.z == r ? v.z : 0.f;
    o.w = a.w == r ? v.w : 0.f;
    *reinterpret_cast<float4*>(base + (size_t)r * C) = o;      // (write-through measured slower here: 24.4 -> 27.6 us)
  }
}

// dst[g*32 + arg[g][c]][c] += grad[g][c]   (one writer per (g,c): no atomics)
__global__ __launch_bounds__(256) void group_scatter_add_kernel(int G, int C,
                                                                const float* __restrict__ grad,
                                                                const unsigned char* __restrict__ arg,
                                                                float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)G * C) return;
  const int g = (int)(i / C), c = (int)(i - (long long)g * C);
  dst[((size_t)g * 32 + arg[i]) * C + c] += grad[i];
}

// column sums over a slab of rows; thread owns 4 columns, 8 row lanes per block
__global__ __launch_bounds__(256) void bnrelu_backward_reduce_kernel(
    int R, int C, const float* __restrict__ dA, const float* __restrict__ X,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ S,
    int rows_per_block, const int* __restrict__ groups, float* __restrict__ det) {
  __shared__ float4 red[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c4 = (blockIdx.x * 32 + cl) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  // (sums over up to a million rows of mixed sign: per-thread chains in fp64, as ATen's CPU kernel accumulates them)
  double a1x = 0., a1y = 0., a1z = 0., a1w = 0., a2x = 0., a2y = 0., a2z = 0., a2w = 0.;
  if (c4 < C) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
    // four rows of loads in flight per thread (one at a time held a CU at ~10 GB/s: 2.7 TB/s over the chip for C = 128);
    // the sums take the rows in the same order as a plain loop
    auto acc = [&](const float4& d, const float4& x) __attribute__((always_inline)) {
      const float tx = (x.x * sc.x + sh.x > 0.f) ? d.x : 0.f;
      const float ty = (x.y * sc.y + sh.y > 0.f) ? d.y : 0.f;
      const float tz = (x.z * sc.z + sh.z > 0.f) ? d.z : 0.f;
      const float tw = (x.w * sc.w + sh.w > 0.f) ? d.w : 0.f;
      a1x += tx, a1y += ty, a1z += tz, a1w += tw;
      a2x += tx * ((x.x - mu.x) * is.x);
      a2y += ty * ((x.y - mu.y) * is.y);
      a2z += tz * ((x.z - mu.z) * is.z);
      a2w += tw * ((x.w - mu.w) * is.w);
    };
    // with a group list, dA holds only the listed groups (compact); the others have dA = 0
    auto xrow = [&](int r) __attribute__((always_inline)) { return groups ? groups[r >> 5] * 32 + (r & 31) : r; };
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      float4 d[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = *reinterpret_cast<const float4*>(dA + (size_t)(r + 8 * u) * C + c4);
        x[u] = *reinterpret_cast<const float4*>(X + (size_t)xrow(r + 8 * u) * C + c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(d[u], x[u]);
    }
    for (; r < r1; r += 8)
      acc(*reinterpret_cast<const float4*>(dA + (size_t)r * C + c4), *reinterpret_cast<const float4*>(X + (size_t)xrow(r) * C + c4));
  }
  const float4 s1 = make_float4((float)a1x, (float)a1y, (float)a1z, (float)a1w);
  const float4 s2 = make_float4((float)a2x, (float)a2y, (float)a2z, (float)a2w);
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl < 2 && c4 < C) {
    float4 t = red[rl][0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 u = red[rl][k][cl];
      t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
    }
    if (det) {   // deterministic mode: row blockIdx.y of the [by][2][C] partials
      *reinterpret_cast<float4*>(det + ((size_t)blockIdx.y * 2 + rl) * C + c4) = t;
    } else {
      float* dst = S + (size_t)rl * C + c4;
      atomicAdd(dst + 0, t.x);
      atomicAdd(dst + 1, t.y);
      atomicAdd(dst + 2, t.z);
      atomicAdd(dst + 3, t.w);
    }
  }
}

// in place: dA <- gamma*invstd*(t - S1/R - xhat*S2/R); gsum[g][c] = sum over the
// group's 32 rows of the result (nullable)
__global__ __launch_bounds__(256) void bnrelu_backward_apply_kernel(
    int G, int C, float* __restrict__ dA, const float* __restrict__ X,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ S, float inv_rows,
    float* __restrict__ gsum, const int* __restrict__ inv_group, float* __restrict__ dX) {
  const int c4 = (blockIdx.x * 32 + (threadIdx.x & 31)) * 4;
  const int g = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (c4 >= C || g >= G) return;
  // inv_group[g] = position of group g in the compact dA, or -1 (its dA is zero)
  const int cg = inv_group ? inv_group[g] : g;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + c4);
  const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
  const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c4);
  const float4 s1 = *reinterpret_cast<const float4*>(S + c4);
  const float4 s2 = *reinterpret_cast<const float4*>(S + C + c4);
  const float kx = ga.x * is.x, ky = ga.y * is.y, kz = ga.z * is.z, kw = ga.w * is.w;
  const float m1x = s1.x * inv_rows, m1y = s1.y * inv_rows, m1z = s1.z * inv_rows, m1w = s1.w * inv_rows;
  const float m2x = s2.x * inv_rows, m2y = s2.y * inv_rows, m2z = s2.z * inv_rows, m2w = s2.w * inv_rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < 32; ++r) {
    const size_t o = ((size_t)g * 32 + r) * C + c4;
    const float4 d = cg >= 0 ? *reinterpret_cast<const float4*>(dA + ((size_t)cg * 32 + r) * C + c4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 x = *reinterpret_cast<const float4*>(X + o);
    float4 y;
    y.x = kx * (((x.x * sc.x + sh.x > 0.f) ? d.x : 0.f) - m1x - ((x.x - mu.x) * is.x) * m2x);
    y.y = ky * (((x.y * sc.y + sh.y > 0.f) ? d.y : 0.f) - m1y - ((x.y - mu.y) * is.y) * m2y);
    y.z = kz * (((x.z * sc.z + sh.z > 0.f) ? d.z : 0.f) - m1z - ((x.z - mu.z) * is.z) * m2z);
    y.w = kw * (((x.w * sc.w + sh.w > 0.f) ? d.w : 0.f) - m1w - ((x.w - mu.w) * is.w) * m2w);
    store_wt4(dX + o, y);
    acc.x += y.x, acc.y += y.y, acc.z += y.z, acc.w += y.w;
  }
  if (gsum) *reinterpret_cast<float4*>(gsum + (size_t)g * C + c4) = acc;
}

// The listed (visible) groups' share of bnrelu_backward_apply: dA holds ONLY the listed groups (compact) and
// is overwritten in place with the gradient of the conv output for those groups; gsum is compact too.  The
// other groups' gradient is BatchNorm's correction alone, dh = u + v * h with the two vectors written by
// bn_correction_kernel -- the caller folds it into small products instead of sweeping those rows
// (patch_embed.py, "masked groups by algebra").
__global__ __launch_bounds__(256) void bnrelu_backward_apply_listed_kernel(
    int n_listed, int C, float* __restrict__ dA, const float* __restrict__ X,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ S, float inv_rows,
    float* __restrict__ gsum, const int* __restrict__ groups, int gsum_by_group) {
  const int c4 = (blockIdx.x * 32 + (threadIdx.x & 31)) * 4;
  const int cg = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (c4 >= C || cg >= n_listed) return;
  const int g = groups[cg];
  const float4 sc = *reinterpret_cast<const float4*>(scale + c4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + c4);
  const float4 mu = *reinterpret_cast<const float4*>(mean + c4);
  const float4 is = *reinterpret_cast<const float4*>(invstd + c4);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c4);
  const float4 s1 = *reinterpret_cast<const float4*>(S + c4);
  const float4 s2 = *reinterpret_cast<const float4*>(S + C + c4);
  const float kx = ga.x * is.x, ky = ga.y * is.y, kz = ga.z * is.z, kw = ga.w * is.w;
  const float m1x = s1.x * inv_rows, m1y = s1.y * inv_rows, m1z = s1.z * inv_rows, m1w = s1.w * inv_rows;
  const float m2x = s2.x * inv_rows, m2y = s2.y * inv_rows, m2z = s2.z * inv_rows, m2w = s2.w * inv_rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < 32; ++r) {
    float* dp = dA + ((size_t)cg * 32 + r) * C + c4;
    const float4 d = *reinterpret_cast<const float4*>(dp);
    const float4 x = *reinterpret_cast<const float4*>(X + ((size_t)g * 32 + r) * C + c4);
    float4 y;
    y.x = kx * (((x.x * sc.x + sh.x > 0.f) ? d.x : 0.f) - m1x - ((x.x - mu.x) * is.x) * m2x);
    y.y = ky * (((x.y * sc.y + sh.y > 0.f) ? d.y : 0.f) - m1y - ((x.y - mu.y) * is.y) * m2y);
    y.z = kz * (((x.z * sc.z + sh.z > 0.f) ? d.z : 0.f) - m1z - ((x.z - mu.z) * is.z) * m2z);
    y.w = kw * (((x.w * sc.w + sh.w > 0.f) ? d.w : 0.f) - m1w - ((x.w - mu.w) * is.w) * m2w);
    store_wt4(dp, y);
    acc.x += y.x, acc.y += y.y, acc.z += y.z, acc.w += y.w;
  }
  if (gsum) *reinterpret_cast<float4*>(gsum + (size_t)(gsum_by_group ? g : cg) * C + c4) = acc;
}

// uv[0][c] = u = -k m1 + k m2 s mu,  uv[1][c] = v = -k m2 s   (k = gamma s, m = S / rows):
// for a row whose activation gradient is zero, dh = u + v * h.
__global__ void bn_correction_kernel(int C, const float* __restrict__ S, float inv_rows,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, float* __restrict__ uv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float k = gamma[c] * invstd[c];
  const float m1 = S[c] * inv_rows, m2 = S[C + c] * inv_rows;
  const float v = -(k * m2) * invstd[c];
  uv[c] = -(k * m1) - v * mean[c];
  uv[C + c] = v;
}

// row sums of the conv-output gradient of the MASKED groups (dh = u + v * h on all 32 rows):
// dgb[groups[cg]] = v * hs[cg] + 32 * xe[cg], hs = sum of the group's 32 conv outputs without the group bias
// term (fsum W^T), xe = u + v * gb  (so that 32 u + v (hs + 32 gb) = v hs + 32 xe)
__global__ __launch_bounds__(256) void masked_group_sums_kernel(int n_listed, int C4, const float4* __restrict__ hs,
                                                                const float4* __restrict__ xe,
                                                                const float4* __restrict__ v,
                                                                const int* __restrict__ groups,
                                                                float4* __restrict__ dgb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_listed * C4) return;
  const int cg = (int)(i / C4), q = (int)(i - (long long)cg * C4);
  const float4 h = hs[i], x = xe[i], vv = v[q];
  float4 o;
  o.x = vv.x * h.x + 32.f * x.x, o.y = vv.y * h.y + 32.f * x.y;
  o.z = vv.z * h.z + 32.f * x.z, o.w = vv.w * h.w + 32.f * x.w;
  dgb[(size_t)groups[cg] * C4 + q] = o;
}

// out[cg][c] = sum over the 32 rows of group groups[cg] of X (thread = 4 channels of one listed group)
__global__ __launch_bounds__(256) void group_sum_listed_kernel(int n_listed, int C4, const float4* __restrict__ X,
                                                               const int* __restrict__ groups,
                                                               float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_listed * C4) return;
  const int cg = (int)(i / C4), q = (int)(i - (long long)cg * C4);
  const float4* row = X + (size_t)groups[cg] * 32 * C4 + q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int r = 0; r < 32; ++r) {
    const float4 v = row[(size_t)r * C4];
    a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
  }
  out[i] = a;
}

// First conv of the embedder (K = 3, first_conv[0]) with BatchNorm's batch statistics:
//   y[m][c] = ((x0*w[c][0] + x1*w[c][1]) + x2*w[c][2]) + b[c];  stats[0][c] += sum_m y,
//   stats[1][c] += sum_m y*y  (fp32 per-thread partials over 64 rows, then fp64 atomics).
// Thread = 4 adjacent channels of one row phase; weights stay in registers; the pass is
// bound by the (R, C) store.
constexpr int C1_ROWS = 512;   // rows per block
__global__ __launch_bounds__(256) void conv1_stats_kernel(int R, int C, const float* __restrict__ x,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, double* __restrict__ stats,
                                                          double* __restrict__ det) {
  extern __shared__ float red[];             // [phases][2][C]
  const int q = C >> 2;                      // channel quads
  const int phases = 256 / q;                // row phases per block
  const int cq = threadIdx.x % q, ph = threadIdx.x / q;
  const int c = cq * 4;
  float w[4][3], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[j][0] = W[(c + j) * 3 + 0], w[j][1] = W[(c + j) * 3 + 1], w[j][2] = W[(c + j) * 3 + 2];
    b[j] = bias ? bias[c + j] : 0.f;
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int mbeg = blockIdx.x * C1_ROWS, mend = min(R, mbeg + C1_ROWS);
  if (ph < phases)
    for (int m = mbeg + ph; m < mend; m += phases) {
      const float x0 = x[(size_t)m * 3], x1 = x[(size_t)m * 3 + 1], x2 = x[(size_t)m * 3 + 2];
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = ((x0 * w[j][0] + x1 * w[j][1]) + x2 * w[j][2]) + b[j];
        s1[j] += v[j];
        s2[j] += v[j] * v[j];
      }
      *reinterpret_cast<float4*>(y + (size_t)m * C + c) = make_float4(v[0], v[1], v[2], v[3]);   // (write-through: 36.5 -> 41.6 us)
    }
  if (ph < phases) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[(ph * 2 + 0) * C + c + j] = s1[j];
      red[(ph * 2 + 1) * C + c + j] = s2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    double t = 0.0;
    for (int p = 0; p < phases; ++p) t += (double)red[p * 2 * C + i];
    if (det) det[(size_t)blockIdx.x * 2 * C + i] = t;
    else atomicAdd(stats + i, t);
  }
}

// Training-mode nn.BatchNorm1d bookkeeping in one launch: batch mean / biased variance
// from the accumulated sums (fp64 sums, or P fp32 partial sets), running estimates
// (unbiased variance, momentum), the update counter, and the affine y = x*scale + shift.
__global__ void bn_finalize_kernel(int C, double rows, const double* __restrict__ st64,
                                   const float* __restrict__ part, int P,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ rmean,
                                   float* __restrict__ rvar, long long* __restrict__ counter,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && counter) *counter += 1;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  if (st64) {
    s1 = st64[c], s2 = st64[C + c];
  } else {
    for (int p = 0; p < P; ++p) s1 += (double)part[(p * 2 + 0) * C + c], s2 += (double)part[(p * 2 + 1) * C + c];
  }
  const double mu = s1 / rows;
  double var = s2 / rows - mu * mu;
  if (var < 0.0) var = 0.0;
  const float m = (float)mu, v = (float)var;
  const float is = 1.0f / sqrtf(v + eps);
  if (rmean) rmean[c] = (1.0f - momentum) * rmean[c] + momentum * m;
  if (rvar) rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)(var * (rows / (rows > 1.0 ? rows - 1.0 : 1.0)));
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - m * sc;
  mean[c] = m;
  invstd[c] = is;
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_group_max_scatter(int G, int C, const float* grad, const unsigned char* arg,
                                      float* out, pdae_stream_t stream) {
  if (G < 0 || C <= 0 || C % 4 != 0) return bad_arg("group_max_scatter: C must be a positive multiple of 4");
  if (G == 0) return PDAE_OK;
  if (!grad || !arg || !out) return bad_arg("group_max_scatter: null pointer");
  if (G > 65535 * 32) return unsupported("group_max_scatter: too many groups");
  hipLaunchKernelGGL(group_max_scatter_kernel, dim3((C / 4 + 63) / 64, G), dim3(256), 0,
                     as_stream(stream), G, C, grad, arg, out);
  return check_launch("group_max_scatter");
}

extern "C" int pdae_group_scatter_add(int G, int C, const float* grad, const unsigned char* arg,
                                      float* dst, pdae_stream_t stream) {
  if (G < 0 || C <= 0) return bad_arg("group_scatter_add: bad size");
  if (G == 0) return PDAE_OK;
  if (!grad || !arg || !dst) return bad_arg("group_scatter_add: null pointer");
  const long long n = (long long)G * C;
  hipLaunchKernelGGL(group_scatter_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     as_stream(stream), G, C, grad, arg, dst);
  return check_launch("group_scatter_add");
}

namespace pdae {
int bnrelu_backward_sums(hipStream_t s, int Rsum, int C, const float* dA, const float* X, const float* scale, const float* shift,
                         const float* mean, const float* invstd, float* S, const int* groups) {
  (void)hipMemsetAsync(S, 0, sizeof(float) * 2 * (size_t)C, s);
  if (Rsum <= 0) return check_launch("bnrelu_backward_sums");
  int rows = 512;
  int by = (Rsum + rows - 1) / rows;
  if (by > 16384) {
    rows = (Rsum + 16383) / 16384;
    by = (Rsum + rows - 1) / rows;
  }
  int rc = PDAE_OK;
  float* det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)by * 2 * C, &rc));
  if (rc) return rc;
  hipLaunchKernelGGL(bnrelu_backward_reduce_kernel, dim3((C / 4 + 31) / 32, by), dim3(256), 0, s, Rsum, C, dA, X, scale,
                     shift, mean, invstd, S, rows, groups, det);
  if (det && (rc = det_reduce(s, by, 2 * C, det, S, 2 * C))) return rc;
  return check_launch("bnrelu_backward_sums");
}
}  // namespace pdae

// The apply halves of pdae_bnrelu_backward / pdae_bnrelu_backward_listed for a caller that already holds the sums S (from
// pdae_rows_gemm_bnrelu_stats: the data-gradient GEMM that produced dA left them): same arguments, S is an INPUT.  dA may
// already carry the ReLU mask (the apply re-evaluates it: idempotent).
extern "C" int pdae_bnrelu_backward_apply(int G, int C, float* dA, const float* X, const float* scale,
                                          const float* shift, const float* mean, const float* invstd,
                                          const float* gamma, const float* S, float* gsum, int n_listed,
                                          const int32_t* groups, const int32_t* inv_group, float* dX,
                                          pdae_stream_t stream) {
  if (G < 0 || C <= 0 || C % 4 != 0) return bad_arg("bnrelu_backward_apply: C must be a positive multiple of 4");
  if (G == 0) return PDAE_OK;
  if (!dA || !X || !scale || !shift || !mean || !invstd || !gamma || !S) return bad_arg("bnrelu_backward_apply: null pointer");
  if ((groups == nullptr) != (inv_group == nullptr) || (groups && !dX))
    return bad_arg("bnrelu_backward_apply: groups, inv_group and dX go together");
  if (!groups) dX = dA;
  (void)n_listed;
  hipLaunchKernelGGL(bnrelu_backward_apply_kernel, dim3((C / 4 + 31) / 32, (G + 7) / 8), dim3(256), 0, as_stream(stream),
                     G, C, dA, X, scale, shift, mean, invstd, gamma, S, 1.0f / (float)(G * 32), gsum, inv_group, dX);
  return check_launch("bnrelu_backward_apply");
}

extern "C" int pdae_bnrelu_backward_listed_apply(int G, int C, float* dA, const float* X, const float* scale,
                                                 const float* shift, const float* mean, const float* invstd,
                                                 const float* gamma, const float* S, float* gsum, int gsum_by_group,
                                                 float* uv, int n_listed, const int32_t* groups, pdae_stream_t stream) {
  if (G < 0 || n_listed < 0 || n_listed > G || C <= 0 || C % 4 != 0)
    return bad_arg("bnrelu_backward_listed_apply: C must be a positive multiple of 4, n_listed <= G");
  if (!S || !scale || !shift || !mean || !invstd || !gamma || (n_listed && (!dA || !X || !groups)))
    return bad_arg("bnrelu_backward_listed_apply: null pointer");
  hipStream_t s = as_stream(stream);
  const float inv_rows = G > 0 ? 1.0f / (float)((long long)G * 32) : 0.f;
  if (n_listed > 0)
    hipLaunchKernelGGL(bnrelu_backward_apply_listed_kernel, dim3((C / 4 + 31) / 32, (n_listed + 7) / 8), dim3(256), 0,
                       s, n_listed, C, dA, X, scale, shift, mean, invstd, gamma, const_cast<float*>(S), inv_rows, gsum, groups,
                       gsum_by_group);
  if (uv)
    hipLaunchKernelGGL(bn_correction_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, S, inv_rows, gamma, mean,
                       invstd, uv);
  return check_launch("bnrelu_backward_listed_apply");
}

extern "C" int pdae_bnrelu_backward(int G, int C, float* dA, const float* X, const float* scale,
                                    const float* shift, const float* mean, const float* invstd,
                                    const float* gamma, float* S, float* gsum, int n_listed,
                                    const int32_t* groups, const int32_t* inv_group, float* dX,
                                    pdae_stream_t stream) {
  if (G < 0 || C <= 0 || C % 4 != 0) return bad_arg("bnrelu_backward: C must be a positive multiple of 4");
  if (!S) return bad_arg("bnrelu_backward: null pointer");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(S, 0, sizeof(float) * 2 * (size_t)C, s);
  if (G == 0) return check_launch("bnrelu_backward");
  if (!dA || !X || !scale || !shift || !mean || !invstd || !gamma)
    return bad_arg("bnrelu_backward: null pointer");
  if ((groups == nullptr) != (inv_group == nullptr) || (groups && !dX))
    return bad_arg("bnrelu_backward: groups, inv_group and dX go together");
  if (!groups) dX = dA;                       // in place over all groups
  const int R = G * 32;
  const int Rsum = groups ? n_listed * 32 : R;  // rows that carry a non-zero dA
  int rows = 512;
  int by = (Rsum + rows - 1) / rows;
  if (by > 16384) {
    rows = (Rsum + 16383) / 16384;
    by = (Rsum + rows - 1) / rows;
  }
  if (Rsum > 0) {
    int rc = PDAE_OK;
    float* det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)by * 2 * C, &rc));
    if (rc) return rc;
    hipLaunchKernelGGL(bnrelu_backward_reduce_kernel, dim3((C / 4 + 31) / 32, by), dim3(256), 0, s, Rsum,
                       C, dA, X, scale, shift, mean, invstd, S, rows, groups, det);
    if (det && (rc = det_reduce(s, by, 2 * C, det, S, 2 * C))) return rc;
  }
  hipLaunchKernelGGL(bnrelu_backward_apply_kernel, dim3((C / 4 + 31) / 32, (G + 7) / 8), dim3(256), 0, s,
                     G, C, dA, X, scale, shift, mean, invstd, gamma, S, 1.0f / (float)R, gsum, inv_group,
                     dX);
  return check_launch("bnrelu_backward");
}

extern "C" int pdae_embed_conv1_stats(int R, int C, const float* x, const float* W, const float* bias,
                                      float* y, double* stats, pdae_stream_t stream) {
  if (R < 0 || C <= 0 || C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0)
    return bad_arg("embed_conv1_stats: C/4 must divide 256");
  if (R == 0) return PDAE_OK;
  if (!x || !W || !y || !stats) return bad_arg("embed_conv1_stats: null pointer");
  const int phases = 256 / (C / 4), blocks = (R + C1_ROWS - 1) / C1_ROWS;
  int rc = PDAE_OK;
  double* det = static_cast<double*>(det_workspace(sizeof(double) * (size_t)blocks * 2 * C, &rc));
  if (rc) return rc;
  hipLaunchKernelGGL(conv1_stats_kernel, dim3(blocks), dim3(256), sizeof(float) * phases * 2 * C,
                     as_stream(stream), R, C, x, W, bias, y, stats, det);
  if (det) return det_reduce_f64(as_stream(stream), blocks, 2 * C, det, stats);
  return check_launch("embed_conv1_stats");
}

extern "C" int pdae_bn_finalize(int C, long long rows, const double* stats64, const float* partials, int P,
                                const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, long long* num_batches_tracked,
                                float* scale, float* shift, float* mean, float* invstd,
                                pdae_stream_t stream) {
  if (C <= 0 || rows <= 0) return bad_arg("bn_finalize: C and rows must be positive");
  if ((!stats64 && (!partials || P <= 0)) || !gamma || !beta || !scale || !shift || !mean || !invstd)
    return bad_arg("bn_finalize: null pointer");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), C,
                     (double)rows, stats64, partials, P, gamma, beta, eps, momentum, running_mean,
                     running_var, num_batches_tracked, scale, shift, mean, invstd);
  return check_launch("bn_finalize");
}

extern "C" int pdae_bnrelu_backward_listed(int G, int C, float* dA, const float* X, const float* scale,
                                           const float* shift, const float* mean, const float* invstd,
                                           const float* gamma, float* S, float* gsum, int gsum_by_group,
                                           float* uv, int n_listed, const int32_t* groups,
                                           pdae_stream_t stream) {
  if (G < 0 || n_listed < 0 || n_listed > G || C <= 0 || C % 4 != 0)
    return bad_arg("bnrelu_backward_listed: C must be a positive multiple of 4, n_listed <= G");
  if (!S) return bad_arg("bnrelu_backward_listed: null pointer");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(S, 0, sizeof(float) * 2 * (size_t)C, s);
  if (!scale || !shift || !mean || !invstd || !gamma || (n_listed && (!dA || !X || !groups)))
    return bad_arg("bnrelu_backward_listed: null pointer");
  const int Rsum = n_listed * 32;
  int rc = PDAE_OK;
  if (Rsum > 0) {
    int rows = 512;
    int by = (Rsum + rows - 1) / rows;
    if (by > 16384) {
      rows = (Rsum + 16383) / 16384;
      by = (Rsum + rows - 1) / rows;
    }
    float* det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)by * 2 * C, &rc));
    if (rc) return rc;
    hipLaunchKernelGGL(bnrelu_backward_reduce_kernel, dim3((C / 4 + 31) / 32, by), dim3(256), 0, s, Rsum, C, dA, X,
                       scale, shift, mean, invstd, S, rows, groups, det);
    if (det && (rc = det_reduce(s, by, 2 * C, det, S, 2 * C))) return rc;
  }
  const float inv_rows = G > 0 ? 1.0f / (float)((long long)G * 32) : 0.f;
  if (n_listed > 0)
    hipLaunchKernelGGL(bnrelu_backward_apply_listed_kernel, dim3((C / 4 + 31) / 32, (n_listed + 7) / 8), dim3(256), 0,
                       s, n_listed, C, dA, X, scale, shift, mean, invstd, gamma, S, inv_rows, gsum, groups, gsum_by_group);
  if (uv)
    hipLaunchKernelGGL(bn_correction_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, S, inv_rows, gamma, mean,
                       invstd, uv);
  return check_launch("bnrelu_backward_listed");
}

extern "C" int pdae_group_sum_listed(int n_listed, int C, const float* X, const int32_t* groups, float* out,
                                     pdae_stream_t stream) {
  if (n_listed < 0 || C <= 0 || C % 4 != 0) return bad_arg("group_sum_listed: C must be a positive multiple of 4");
  if (n_listed == 0) return PDAE_OK;
  if (!X || !groups || !out) return bad_arg("group_sum_listed: null pointer");
  const long long n = (long long)n_listed * (C / 4);
  hipLaunchKernelGGL(group_sum_listed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     n_listed, C / 4, reinterpret_cast<const float4*>(X), groups, reinterpret_cast<float4*>(out));
  return check_launch("group_sum_listed");
}

extern "C" int pdae_masked_group_sums(int n_listed, int C, const float* hs, const float* xe, const float* v,
                                      const int32_t* groups, float* dgb, pdae_stream_t stream) {
  if (n_listed < 0 || C <= 0 || C % 4 != 0) return bad_arg("masked_group_sums: C must be a positive multiple of 4");
  if (n_listed == 0) return PDAE_OK;
  if (!hs || !xe || !v || !groups || !dgb) return bad_arg("masked_group_sums: null pointer");
  const long long n = (long long)n_listed * (C / 4);
  hipLaunchKernelGGL(masked_group_sums_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     n_listed, C / 4, reinterpret_cast<const float4*>(hs), reinterpret_cast<const float4*>(xe),
                     reinterpret_cast<const float4*>(v), groups, reinterpret_cast<float4*>(dgb));
  return check_launch("masked_group_sums");
}

// Weight gradient of the embedder's first conv (K = 3): dW[c][k] = sum_r d[r][c] * x[r][k], one pass over d.
// On the grouped MFMA kernel this N = 128, K = 3(+1) product cost 78 + 44 us (a 128x128 tile for 3 useful
// columns, 512 partial tiles to add); here a thread owns four channels of a row phase and walks its rows, the
// phases meet in LDS and every block stores one partial [3][C] (the caller adds the partials in block order).
namespace pdae {
constexpr int C1B_ROWS = 1024;
__global__ __launch_bounds__(256) void conv1_backward_weight_kernel(int R, int C4, const float4* __restrict__ d,
                                                                    const float* __restrict__ x,
                                                                    float4* __restrict__ part) {
  extern __shared__ float4 c1b_red[];             // [phases][3][C4]
  const int PH = 256 / C4;
  const int q = threadIdx.x % C4, ph = threadIdx.x / C4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
  const int r0 = blockIdx.x * C1B_ROWS, r1 = min(R, r0 + C1B_ROWS);
#pragma unroll 4
  for (int r = r0 + ph; r < r1; r += PH) {
    const float4 g = d[(size_t)r * C4 + q];
    const float x0 = x[(size_t)r * 3], x1 = x[(size_t)r * 3 + 1], x2 = x[(size_t)r * 3 + 2];
    a0.x += g.x * x0, a0.y += g.y * x0, a0.z += g.z * x0, a0.w += g.w * x0;
    a1.x += g.x * x1, a1.y += g.y * x1, a1.z += g.z * x1, a1.w += g.w * x1;
    a2.x += g.x * x2, a2.y += g.y * x2, a2.z += g.z * x2, a2.w += g.w * x2;
  }
  c1b_red[(ph * 3 + 0) * C4 + q] = a0;
  c1b_red[(ph * 3 + 1) * C4 + q] = a1;
  c1b_red[(ph * 3 + 2) * C4 + q] = a2;
  __syncthreads();
  if (ph == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float4 t = c1b_red[k * C4 + q];
      for (int p = 1; p < PH; ++p) {
        const float4 u = c1b_red[(p * 3 + k) * C4 + q];
        t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
      }
      part[((size_t)blockIdx.x * 3 + k) * C4 + q] = t;
    }
  }
}
}  // namespace pdae

extern "C" int pdae_embed_conv1_backward_weight_parts(int R) { return (R + pdae::C1B_ROWS - 1) / pdae::C1B_ROWS; }

extern "C" int pdae_embed_conv1_backward_weight(int R, int C, const float* d, const float* x, float* part,
                                                pdae_stream_t stream) {
  if (R < 0 || C <= 0 || C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0)
    return bad_arg("embed_conv1_backward_weight: C/4 must divide 256");
  if (R == 0) return PDAE_OK;
  if (!d || !x || !part) return bad_arg("embed_conv1_backward_weight: null pointer");
  hipLaunchKernelGGL(conv1_backward_weight_kernel, dim3((R + C1B_ROWS - 1) / C1B_ROWS), dim3(256),
                     sizeof(float4) * 3 * 256, as_stream(stream), R, C / 4, reinterpret_cast<const float4*>(d), x,
                     reinterpret_cast<float4*>(part));
  return check_launch("embed_conv1_backward_weight");
}
