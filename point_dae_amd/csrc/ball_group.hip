// ball_group.hip -- ball query, grouping and gather (+ their gradients).
//
// Semantics: extensions/pointnet2/_ext_src/src/ball_query_gpu.cu:12-47,
// group_points_gpu.cu:11-67, sampling_gpu.cu:11-50 of the reference (restated
// in oracle/pdae_oracle.c).  The reference runs ONE block per cloud: a thread
// per centre scans all N points serially (ball query), and grouping walks
// nsample serially per thread.  Here:
//   * ball query: one wave per centre over an SoA copy of the cloud in LDS;
//     64 candidates per step, hits ranked with ballot + mbcnt so the output
//     keeps the reference's ascending-index order, early exit at nsample.
//   * grouping: flat (point,sample) index on the lanes, channel loop inside;
//     16-byte stores, the 2-4 KB feature row it gathers from stays in L1.
//   * gradients: scatter-add through LDS float atomics per (cloud, channel),
//     then plain stores -- no global atomics and no memset pass.
#include "common.h"

namespace pdae {

// ---- ball query -------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, float radius, int nsample,
                                                         int cpw,
                                                         const float* __restrict__ new_xyz_all,
                                                         const float* __restrict__ xyz_all,
                                                         int32_t* __restrict__ idx_all) {
  extern __shared__ float soa[];  // x[n] y[n] z[n]
  float* sx = soa;
  float* sy = soa + n;
  float* sz = soa + 2 * n;
  const int bi = blockIdx.y;
  const float* xyz = xyz_all + (size_t)bi * n * 3;
  for (int i = threadIdx.x; i < n; i += 256) {
    sx[i] = xyz[i * 3 + 0];
    sy[i] = xyz[i * 3 + 1];
    sz[i] = xyz[i * 3 + 2];
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  const float radius2 = radius * radius;
  const int j0 = (blockIdx.x * 4 + wave) * cpw;
  const int j1 = min(m, j0 + cpw);
  for (int j = j0; j < j1; ++j) {
    const float* c = new_xyz_all + ((size_t)bi * m + j) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    int32_t* out = idx_all + ((size_t)bi * m + j) * nsample;
    int cnt = 0;
    int first = 0;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += kWave) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < n) {
        // (new - x)^2 summed x,y,z as in ball_query_gpu.cu:33-35
        const float d2 = sqdist(cx, cy, cz, sx[k], sy[k], sz[k]);
        hit = d2 < radius2;
      }
      const unsigned long long mask = __ballot(hit);
      if (mask == 0) continue;
      if (cnt == 0) first = k0 + (int)__builtin_ctzll(mask);
      const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi(
                                (unsigned)(mask >> 32),
                                __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      if (hit && pos < nsample) out[pos] = k;
      cnt += __popcll(mask);
    }
    if (cnt > nsample) cnt = nsample;
    // pad with the first hit; an empty ball yields zeros (first = 0)
    for (int l = cnt + lane; l < nsample; l += kWave) out[l] = first;
  }
}

// ---- grouping ---------------------------------------------------------------
// out[b,c,t] = points[b,c,idx[b,t]], t flat over (npoints*nsample); 4 t per lane.
__global__ __launch_bounds__(256) void group_points_kernel(int c, int n, int total,
                                                           const float* __restrict__ points_all,
                                                           const int32_t* __restrict__ idx_all,
                                                           float* __restrict__ out_all) {
  const int bi = blockIdx.y;
  const float* points = points_all + (size_t)bi * n * c;
  const int32_t* idx = idx_all + (size_t)bi * total;
  float* out = out_all + (size_t)bi * total * c;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= total) return;
  if (t + 3 < total && (total & 3) == 0) {
    const int4 id = *reinterpret_cast<const int4*>(idx + t);
    for (int l = 0; l < c; ++l) {
      const float* row = points + (size_t)l * n;
      const float4 v = make_float4(row[id.x], row[id.y], row[id.z], row[id.w]);
      *reinterpret_cast<float4*>(out + (size_t)l * total + t) = v;
    }
  } else {
    for (int u = t; u < min(total, t + 4); ++u) {
      const int id = idx[u];
      for (int l = 0; l < c; ++l) out[(size_t)l * total + u] = points[(size_t)l * n + id];
    }
  }
}

// grad_points[b,c,:] = scatter-add of grad_out[b,c,t] at idx[b,t]; one
// workgroup per (channel, cloud), accumulation in LDS.
__global__ __launch_bounds__(256) void group_points_grad_kernel(
    int c, int n, int total, const float* __restrict__ grad_out_all,
    const int32_t* __restrict__ idx_all, float* __restrict__ grad_points_all) {
  extern __shared__ float acc[];  // n
  const int l = blockIdx.x, bi = blockIdx.y;
  const float* go = grad_out_all + ((size_t)bi * c + l) * total;
  const int32_t* idx = idx_all + (size_t)bi * total;
  float* gp = grad_points_all + ((size_t)bi * c + l) * n;
  for (int i = threadIdx.x; i < n; i += 256) acc[i] = 0.f;
  __syncthreads();
  for (int t = threadIdx.x; t < total; t += 256) atomicAdd(&acc[idx[t]], go[t]);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) gp[i] = acc[i];
}

// large-n fallback of the gradient: zero-fill + global atomics
__global__ void group_points_grad_atomic_kernel(int c, int n, int total,
                                                const float* __restrict__ grad_out_all,
                                                const int32_t* __restrict__ idx_all,
                                                float* __restrict__ grad_points_all) {
  const int l = blockIdx.y, bi = blockIdx.z;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const float v = grad_out_all[((size_t)bi * c + l) * total + t];
  atomicAdd(grad_points_all + ((size_t)bi * c + l) * n + idx_all[(size_t)bi * total + t], v);
}


// ---------------------------------------------------------------------------------------------
// QueryAndGroup in ROW layout (extensions/pointnet2/pointnet2_utils.py:345-361: grouping_operation of the coordinates,
// minus the centre, concatenated with grouping_operation of the features) as ONE pass that writes the set-abstraction
// MLP's input rows directly:
//     out[(b np + p) ns + s] = [ xyz[b, idx[b,p,s]] - new_xyz[b,p] | 0 | features[b N + idx[b,p,s], :C] ]   (4 + C floats)
// (the zero keeps K a multiple of 4 for the row GEMMs).  The PyTorch form was index_select + subtract + zeros + cat +
// index_select: 0.56 ms at the second level of cfg2; its backward an index_add with global float atomics, 0.50 ms.
__global__ __launch_bounds__(256) void sa_group_rows_kernel(long long total4, int Q, int N, int np, int ns,
                                                            const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                            const int32_t* __restrict__ idx, const float* __restrict__ feat,
                                                            float4* __restrict__ out) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;     // float4 index over (row, q)
  if (f >= total4) return;
  const long long r = f / Q;
  const int q = (int)(f - r * Q);
  const long long centre = r / ns;                                   // b np + p
  const long long b = centre / np;
  const long long src = b * N + idx[r];
  float4 v;
  if (q == 0) {
    const float* a = xyz + src * 3;
    const float* c = new_xyz + centre * 3;
    v = make_float4(a[0] - c[0], a[1] - c[1], a[2] - c[2], 0.f);
  } else {
    v = *reinterpret_cast<const float4*>(feat + src * (long long)(4 * (Q - 1)) + 4 * (q - 1));
  }
  out[f] = v;
}

// Gradient to the features: dfeat[b N + j, c] = sum over the rows (p, s) of cloud b with idx = j of dout[row, 4 + c].
// LDS FLOAT atomics are the wrong tool here: ds_add_f32 retires about one lane every two cycles per CU -- an accumulator
// array in LDS fed by 134 M of them took 650 us at cfg2's second level, 119 us with the adds made plain (racy) stores.
// So the rows of a cloud are first ORDERED BY SOURCE POINT with a counting sort in LDS (integer atomics on 8 k items:
// histogram, prefix sum, scatter), then every point's rows are summed in registers by one sub-group reading whole
// 4 C-byte feature rows, and written once: no float atomics, no accumulators in LDS, no zero-fill of dfeat.  SPLIT blocks
// per cloud share the (cheap) sort and take a range of points each.
template <int SPLIT>
__global__ __launch_bounds__(256) void sa_group_rows_grad_kernel(int N, int rows, int C, int W,
                                                                 const int32_t* __restrict__ idx,
                                                                 const float* __restrict__ dout, float* __restrict__ dfeat) {
  extern __shared__ int lds_i[];                 // cnt[N] | off[N + 1] | list[rows]
  int* cnt = lds_i;
  int* off = lds_i + N;
  int* list = off + N + 1;
  const int b = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT;
  const int tid = threadIdx.x;
  const int32_t* ib = idx + (long long)b * rows;
  for (int i = tid; i < N; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int r = tid; r < rows; r += 256) atomicAdd(&cnt[ib[r]], 1);
  __syncthreads();
  // exclusive prefix sum of cnt (N <= 4096): one wave, N / 64 entries per lane + a shuffle scan of the lane totals
  if (tid < 64) {
    const int per = (N + 63) / 64, j0 = tid * per;
    int sum = 0;
    for (int k = 0; k < per; ++k)
      if (j0 + k < N) sum += cnt[j0 + k];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (tid >= d) incl += t;
    }
    int run = incl - sum;
    for (int k = 0; k < per; ++k)
      if (j0 + k < N) {
        off[j0 + k] = run;
        run += cnt[j0 + k];
      }
    if (tid == 63) off[N] = incl;
  }
  __syncthreads();
  for (int i = tid; i < N; i += 256) cnt[i] = off[i];           // cnt becomes the scatter cursor
  __syncthreads();
  for (int r = tid; r < rows; r += 256) list[atomicAdd(&cnt[ib[r]], 1)] = r;
  __syncthreads();
  // ---- sums: LPR lanes (one channel quad each) per point, SUBS points in flight
  const int LPR = C / 4, SUBS = 256 / LPR;
  const int lane = tid % LPR, sub = tid / LPR;
  if (sub >= SUBS) return;
  const int jbeg = (int)((long long)N * part / SPLIT), jend = (int)((long long)N * (part + 1) / SPLIT);
  const float* base = dout + (long long)b * rows * W + 4 + 4 * lane;
  constexpr int UN = 8;
  for (int j = jbeg + sub; j < jend; j += SUBS) {
    const int k0 = off[j], k1 = off[j + 1];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k0; k < k1; k += UN) {
      float4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long long)list[min(k + u, k1 - 1)] * W);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (k + u < k1) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
    }
    *reinterpret_cast<float4*>(dfeat + ((long long)b * N + j) * C + 4 * lane) = s;
  }
}

}  // namespace pdae

extern "C" int pdae_ball_query(int b, int n, int m, float radius, int nsample,
                               const float* new_xyz, const float* xyz, int32_t* idx,
                               pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return bad_arg("ball_query: negative size");
  if (b == 0 || m == 0 || nsample == 0) return PDAE_OK;
  if (!new_xyz || !xyz || !idx) return bad_arg("ball_query: null pointer");
  if (b > 65535) return unsupported("ball_query: b > 65535");
  const size_t lds = (size_t)n * 3 * sizeof(float);
  if (lds > 150 * 1024) return unsupported("ball_query: n > 12800 not implemented");
  hipStream_t s = as_stream(stream);
  if (n == 0) {  // nothing in range: the reference leaves its zero-initialised idx
    if (hipMemsetAsync(idx, 0, sizeof(int32_t) * (size_t)b * m * nsample, s) != hipSuccess)
      return check_launch("ball_query");
    return PDAE_OK;
  }
  int cpw = 8;
  while (cpw > 1 && (long long)b * ((m + cpw - 1) / cpw) < 4096) cpw >>= 1;
  const int waves = (m + cpw - 1) / cpw;
  hipLaunchKernelGGL(ball_query_kernel, dim3((waves + 3) / 4, b), dim3(256), lds, s, n, m, radius,
                     nsample, cpw, new_xyz, xyz, idx);
  return check_launch("ball_query");
}

extern "C" int pdae_group_points(int b, int c, int n, int npoints, int nsample,
                                 const float* points, const int32_t* idx, float* out,
                                 pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0)
    return bad_arg("group_points: negative size");
  const long long total = (long long)npoints * nsample;
  if (b == 0 || c == 0 || total == 0) return PDAE_OK;
  if (!points || !idx || !out) return bad_arg("group_points: null pointer");
  if (b > 65535 || total > (1ll << 30)) return unsupported("group_points: size");
  hipStream_t s = as_stream(stream);
  const int blocks = (int)((total + 1023) / 1024);
  hipLaunchKernelGGL(group_points_kernel, dim3(blocks, b), dim3(256), 0, s, c, n, (int)total,
                     points, idx, out);
  return check_launch("group_points");
}

extern "C" int pdae_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                      const float* grad_out, const int32_t* idx,
                                      float* grad_points, pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0)
    return bad_arg("group_points_grad: negative size");
  if (b == 0 || c == 0 || n == 0) return PDAE_OK;
  if (!grad_points) return bad_arg("group_points_grad: null pointer");
  const long long total = (long long)npoints * nsample;
  hipStream_t s = as_stream(stream);
  if (total == 0) {
    (void)hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, s);
    return check_launch("group_points_grad");
  }
  if (!grad_out || !idx) return bad_arg("group_points_grad: null pointer");
  if (b > 65535 || c > 65535 || total > (1ll << 30)) return unsupported("group_points_grad: size");
  if ((size_t)n * sizeof(float) <= 128 * 1024) {
    hipLaunchKernelGGL(group_points_grad_kernel, dim3(c, b), dim3(256), (size_t)n * sizeof(float),
                       s, c, n, (int)total, grad_out, idx, grad_points);
  } else {
    (void)hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, s);
    hipLaunchKernelGGL(group_points_grad_atomic_kernel, dim3((unsigned)((total + 255) / 256), c, b),
                       dim3(256), 0, s, c, n, (int)total, grad_out, idx, grad_points);
  }
  return check_launch("group_points_grad");
}

// gather_points is grouping with nsample = 1 (sampling_gpu.cu:11-23 vs
// group_points_gpu.cu:11-31 compute the same thing for a (b, m) index).
extern "C" int pdae_gather_points(int b, int c, int n, int npoints, const float* points,
                                  const int32_t* idx, float* out, pdae_stream_t stream) {
  return pdae_group_points(b, c, n, npoints, 1, points, idx, out, stream);
}

extern "C" int pdae_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out,
                                       const int32_t* idx, float* grad_points,
                                       pdae_stream_t stream) {
  return pdae_group_points_grad(b, c, n, npoints, 1, grad_out, idx, grad_points, stream);
}


// Row-layout QueryAndGroup (pointnet2_utils.py:345-361) for the set-abstraction levels: see sa_group_rows_kernel.
extern "C" int pdae_sa_group_rows(int B, int N, int np, int ns, int C, const float* xyz, const float* new_xyz,
                                  const int32_t* idx, const float* features, float* out, pdae_stream_t stream) {
  using namespace pdae;
  if (B < 0 || N <= 0 || np < 0 || ns < 0 || C < 0) return bad_arg("sa_group_rows: bad size");
  if (C % 4 != 0) return unsupported("sa_group_rows: the feature width must be a multiple of 4");
  const long long rows = (long long)B * np * ns;
  if (rows == 0) return PDAE_OK;
  if (!xyz || !new_xyz || !idx || !out || (C > 0 && !features)) return bad_arg("sa_group_rows: null pointer");
  const int Q = 1 + C / 4;
  const long long total4 = rows * Q;
  if (total4 > (1ll << 31) * 255) return unsupported("sa_group_rows: size");
  hipLaunchKernelGGL(sa_group_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), total4, Q,
                     N, np, ns, xyz, new_xyz, idx, features, reinterpret_cast<float4*>(out));
  return check_launch("sa_group_rows");
}

extern "C" int pdae_sa_group_rows_grad(int B, int N, int np, int ns, int C, const int32_t* idx, const float* dout,
                                       float* dfeatures, pdae_stream_t stream) {
  using namespace pdae;
  if (B < 0 || N <= 0 || np < 0 || ns < 0 || C <= 0) return bad_arg("sa_group_rows_grad: bad size");
  if (C % 4 != 0 || C > 1024) return unsupported("sa_group_rows_grad: the feature width must be a multiple of 4, at most 1024");
  if (B == 0) return PDAE_OK;
  if (!dfeatures) return bad_arg("sa_group_rows_grad: null pointer");
  hipStream_t s = as_stream(stream);
  const long long rows = (long long)np * ns;
  if (rows == 0) {
    (void)hipMemsetAsync(dfeatures, 0, sizeof(float) * (size_t)B * N * C, s);
    return check_launch("sa_group_rows_grad");
  }
  if (!idx || !dout) return bad_arg("sa_group_rows_grad: null pointer");
  const size_t lds = sizeof(int) * ((size_t)2 * N + 1 + (size_t)rows);
  if (N > 4096 || lds > 150 * 1024) return unsupported("sa_group_rows_grad: a cloud's points / rows do not fit LDS");
  constexpr int SPLIT = 4;
  if ((long long)B * SPLIT > 0x7fffffffLL) return unsupported("sa_group_rows_grad: batch size");
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_group_rows_grad_kernel<SPLIT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(sa_group_rows_grad_kernel<SPLIT>, dim3(B * SPLIT), dim3(256), lds, s, N, (int)rows, C, 4 + C, idx, dout,
                     dfeatures);
  return check_launch("sa_group_rows_grad");
}
