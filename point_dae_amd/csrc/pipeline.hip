// pipeline.hip -- loader-side corruption on the device: `dropout_local`.
//
// Reference: datasets/corrupt_util.py:590-612 (corrupt_dropout_local), run per item by 8 DataLoader
// workers on the host (ShapeNet55Dataset.__getitem__ :90-119): draw a drop ratio in [0.1, 0.5) and
// 1..7 clusters whose sizes sum to ratio * P; for each cluster shuffle the cloud, take its first
// point as the seed, argsort ALL points by distance to the seed and cut off the K nearest.  At the
// ~9 k clouds/s of the training step that is 7 argsorts of an 8192-point cloud per item, 60 k
// argsorts per second, on host cores.
//
// Here the whole batch runs in one launch, one 1024-thread block per cloud, the cloud in registers
// (P / 1024 points per thread).  The random draws are INPUTS (seed ranks and cluster sizes, drawn by
// the host side with the reference's distributions), so the kernel is a pure function that the
// oracle restates and the live reference pins (tests/golden/make_pipeline_fixtures.py):
//   seed        = the r-th surviving point in index order (a shuffle's first element is a uniform
//                 draw among the survivors; the order of the survivors does not matter downstream:
//                 a random subset is taken next)          -- block-wide prefix count
//   K nearest   = exact selection without a sort: binary search over the 32-bit pattern of the
//                 squared distance (non-negative floats order like their bits) for the smallest
//                 threshold with at least K survivors at or below it; points below it go, points
//                 equal to it go in index order until K are gone   -- 33 block-wide counts
// Squared distances as the reference computes them: sum over the columns of (p - seed)^2.
#include "common.h"

namespace pdae {

constexpr int DL_THREADS = 1024, DL_MAXPPT = 16, DL_MAXCL = 8;

__device__ __forceinline__ int block_sum_1024(int v, int* lds) {
  // wave sums (DPP-free shuffles), then 16 partials through LDS; returns the block total to all
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < DL_THREADS / 64; ++w) t += lds[w];
  return t;
}

// exclusive prefix of v over the block (thread order) -- wave scan + wave offsets
__device__ __forceinline__ int block_excl_scan_1024(int v, int* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, kWave);
    if (lane >= o) inc += u;
  }
  __syncthreads();
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < w; ++k) off += lds[k];
  return off + inc - v;
}

template <int PPT>
__global__ __launch_bounds__(DL_THREADS) void dropout_local_kernel(int P, const float* __restrict__ xyz,
                                                                   const int* __restrict__ nclusters,
                                                                   const int* __restrict__ seed_rank,
                                                                   const int* __restrict__ sizes,
                                                                   unsigned char* __restrict__ alive_out) {
  __shared__ int red[DL_THREADS / 64];
  __shared__ float seed[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* pts = xyz + (size_t)b * P * 3;
  // thread t owns points t * PPT .. t * PPT + PPT - 1 (index order = thread order, then slot order)
  float px[PPT], py[PPT], pz[PPT];
  bool alive[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid * PPT + i;
    alive[i] = k < P;
    px[i] = alive[i] ? pts[(size_t)k * 3 + 0] : 0.f;
    py[i] = alive[i] ? pts[(size_t)k * 3 + 1] : 0.f;
    pz[i] = alive[i] ? pts[(size_t)k * 3 + 2] : 0.f;
  }
  const int nc = min(nclusters[b], DL_MAXCL);
  for (int c = 0; c < nc; ++c) {
    const int K = sizes[b * DL_MAXCL + c];
    const int rank = seed_rank[b * DL_MAXCL + c];
    // ---- the seed: the rank-th survivor
    int mine = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) mine += alive[i] ? 1 : 0;
    const int before = block_excl_scan_1024(mine, red);
    if (rank >= before && rank < before + mine) {
      int r = rank - before;
#pragma unroll
      for (int i = 0; i < PPT; ++i)
        if (alive[i]) {
          if (r == 0) seed[0] = px[i], seed[1] = py[i], seed[2] = pz[i];
          --r;
        }
    }
    __syncthreads();
    const float sx = seed[0], sy = seed[1], sz = seed[2];
    unsigned key[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = px[i] - sx, dy = py[i] - sy, dz = pz[i] - sz;
      const float d = dx * dx + dy * dy + dz * dz;
      key[i] = alive[i] ? __float_as_uint(d) : 0xffffffffu;
    }
    if (K <= 0) continue;
    // ---- smallest threshold T with count(key <= T) >= K (the K-th smallest key)
    unsigned lo = 0, hi = 0x7f800000u;           // all finite non-negative floats
    while (lo < hi) {
      const unsigned mid = lo + ((hi - lo) >> 1);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < PPT; ++i) cnt += key[i] <= mid ? 1 : 0;
      if (block_sum_1024(cnt, red) >= K) hi = mid;
      else lo = mid + 1;
    }
    const unsigned T = lo;
    int below = 0, equal = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) below += key[i] < T ? 1 : 0, equal += key[i] == T ? 1 : 0;
    const int nbelow = block_sum_1024(below, red);
    const int eq_before = block_excl_scan_1024(equal, red);
    int quota = K - nbelow - eq_before;          // ties at T go in index order
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (key[i] < T) alive[i] = false;
      else if (key[i] == T) {
        if (quota > 0) alive[i] = false;
        --quota;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid * PPT + i;
    if (k < P) alive_out[(size_t)b * P + k] = alive[i] ? 1 : 0;
  }
}


// ---------------------------------------------------------------------------------------------
// The other loader-side stages (datasets/corrupt_util.py, ShapeNet55Dataset.py:67-119), each a pure function of the
// cloud and of the draws the host side makes with the reference's distributions (the live reference pins every one
// of them through tests/golden/make_loader_fixtures.py):
//   norm_affine   'norm' (_pc_normalize :7-17: centroid, max norm) -> up to three affine maps y = x M + t applied one
//                 after the other as 'affine_r3' does (:1062-1070: translate :139-140, scale_nonorm :91-92, rotate
//                 :262-263, reflection :408-409, shear :425-428) -> 'jitter' (:179-191: + sigma * noise)
//   add_global    :830-841 / :42-56: points uniform in the unit ball from three uniforms each, appended
//   add_local     :844-870: Gaussian clusters around cloud points, pulled back into the unit sphere, appended
//   density       :875-897: distance-gated drop from a random viewpoint -> alive mask
//   subset        ShapeNet.random_sample :76-88: the n smallest random keys among the survivors, in key order
// One 1024-thread block per cloud for the stages that reduce over the cloud (norm, subset); the cloud stays in
// registers.  fp32 throughout; the centroid is accumulated in fp64 (the reference's float32 np.mean adds the rows
// one by one: its own rounding is ~4e-7, which is what the parity tolerance covers).

template <int PPT>
__global__ __launch_bounds__(DL_THREADS) void norm_affine_kernel(int P, int out_stride, int normalise,
                                                                 const float* __restrict__ xyz,
                                                                 const int* __restrict__ nmaps,
                                                                 const float* __restrict__ maps,
                                                                 const float* __restrict__ sigma,
                                                                 const float* __restrict__ noise,
                                                                 float* __restrict__ out) {
  __shared__ double red[3][DL_THREADS / 64];
  __shared__ float redm[DL_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* pts = xyz + (size_t)b * P * 3;
  float px[PPT], py[PPT], pz[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * DL_THREADS;          // thread-interleaved: consecutive lanes read consecutive points
    const bool in = k < P;
    px[i] = in ? pts[(size_t)k * 3 + 0] : 0.f;
    py[i] = in ? pts[(size_t)k * 3 + 1] : 0.f;
    pz[i] = in ? pts[(size_t)k * 3 + 2] : 0.f;
  }
  if (normalise) {
    double sx = 0., sy = 0., sz = 0.;
#pragma unroll
    for (int i = 0; i < PPT; ++i) sx += px[i], sy += py[i], sz += pz[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      sx += __shfl_xor(sx, o, kWave), sy += __shfl_xor(sy, o, kWave), sz += __shfl_xor(sz, o, kWave);
    }
    if (lane == 0) red[0][w] = sx, red[1][w] = sy, red[2][w] = sz;
    __syncthreads();
    double cx = 0., cy = 0., cz = 0.;
#pragma unroll
    for (int q = 0; q < DL_THREADS / 64; ++q) cx += red[0][q], cy += red[1][q], cz += red[2][q];
    const float mx = (float)(cx / P), my = (float)(cy / P), mz = (float)(cz / P);
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int k = tid + i * DL_THREADS;
      px[i] -= mx, py[i] -= my, pz[i] -= mz;
      if (k < P) m = fmaxf(m, sqrtf(px[i] * px[i] + py[i] * py[i] + pz[i] * pz[i]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
    if (lane == 0) redm[w] = m;
    __syncthreads();
    m = 0.f;
#pragma unroll
    for (int q = 0; q < DL_THREADS / 64; ++q) m = fmaxf(m, redm[q]);
#pragma unroll
    for (int i = 0; i < PPT; ++i) px[i] /= m, py[i] /= m, pz[i] /= m;
  }
  const int nm = nmaps ? min(nmaps[b], 3) : 0;
  for (int q = 0; q < nm; ++q) {
    const float* M = maps + ((size_t)b * 3 + q) * 12;
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7],
                m22 = M[8], t0 = M[9], t1 = M[10], t2 = M[11];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float x = px[i], y = py[i], z = pz[i];
      px[i] = ((x * m00 + y * m10) + z * m20) + t0;      // row vector times matrix, as np.dot(pointcloud, R)
      py[i] = ((x * m01 + y * m11) + z * m21) + t1;
      pz[i] = ((x * m02 + y * m12) + z * m22) + t2;
    }
  }
  const float sg = sigma ? sigma[b] : 0.f;
  float* dst = out + (size_t)b * out_stride * 3;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * DL_THREADS;
    if (k >= P) continue;
    float x = px[i], y = py[i], z = pz[i];
    if (noise && sg != 0.f) {
      const float* nz = noise + ((size_t)b * P + k) * 3;
      x += sg * nz[0], y += sg * nz[1], z += sg * nz[2];
    }
    dst[(size_t)k * 3 + 0] = x, dst[(size_t)k * 3 + 1] = y, dst[(size_t)k * 3 + 2] = z;
  }
}

// appended points: rows [p0, p0 + count[b]) of cloud b
__global__ __launch_bounds__(256) void add_global_kernel(int nmax, int stride, int p0, const int* __restrict__ count,
                                                         const float* __restrict__ u, float* __restrict__ xyz) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= min(count[b], nmax)) return;
  const float* q = u + ((size_t)b * nmax + j) * 3;
  const float radius = powf(q[0], 1.0f / 3.0f), theta = acosf(q[1]), phi = q[2];
  float* d = xyz + ((size_t)b * stride + p0 + j) * 3;
  const float st = sinf(theta);
  d[0] = radius * st * cosf(phi), d[1] = radius * st * sinf(phi), d[2] = radius * cosf(theta);
}

__global__ __launch_bounds__(256) void add_local_kernel(int nmax, int stride, int p0, const int* __restrict__ count,
                                                        const int* __restrict__ seed, const float* __restrict__ sigma,
                                                        const float* __restrict__ noise, float* __restrict__ xyz) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= min(count[b], nmax)) return;
  const size_t e = (size_t)b * nmax + j;
  const float* c = xyz + ((size_t)b * stride + seed[e]) * 3;
  const float sg = sigma[e];
  float x = c[0] + sg * noise[e * 3 + 0], y = c[1] + sg * noise[e * 3 + 1], z = c[2] + sg * noise[e * 3 + 2];
  const float d2 = (x * x + y * y) + z * z;
  if (d2 > 1.f) x /= d2, y /= d2, z /= d2;                 // (:866-867: divided by the SQUARED norm, as the reference)
  float* d = xyz + ((size_t)b * stride + p0 + j) * 3;
  d[0] = x, d[1] = y, d[2] = z;
}

__global__ __launch_bounds__(256) void density_kernel(int P, int stride, const float* __restrict__ xyz,
                                                      const float* __restrict__ view, const float* __restrict__ gate,
                                                      const float* __restrict__ r, unsigned char* __restrict__ alive) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= P) return;
  const float* p = xyz + ((size_t)b * stride + k) * 3;
  const float vx = view[b * 3], vy = view[b * 3 + 1], vz = view[b * 3 + 2];
  const float dv = sqrtf((vx * vx + vy * vy) + vz * vz);          // |v| (= 1 up to rounding), :884-886
  const float dx = p[0] - vx, dy = p[1] - vy, dz = p[2] - vz;
  float d = sqrtf((dx * dx + dy * dy) + dz * dz);
  d = (d - (dv - 1.f)) / ((dv + 1.f) - (dv - 1.f));
  const bool keep = d * gate[b] < r[(size_t)b * P + k];
  unsigned char* a = alive + (size_t)b * stride + k;
  *a = (*a && keep) ? 1 : 0;
}

// the n smallest keys among the alive points of a cloud, written in ascending key order (ties by index).
// Fewer than n survivors: the rest repeats them cyclically in that order.
template <int PPT>
__global__ __launch_bounds__(DL_THREADS) void subset_kernel(int P, int stride, int n, const float* __restrict__ xyz,
                                                            const unsigned char* __restrict__ alive,
                                                            const float* __restrict__ keys, float* __restrict__ out) {
  extern __shared__ unsigned sel[];                // [2][n]: key bits, point index of the selected
  __shared__ int red[DL_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  unsigned key[PPT];
  int mine = 0;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid * PPT + i;                   // index order = thread order, then slot order
    const bool in = k < P && (!alive || alive[(size_t)b * stride + k]);
    key[i] = in ? __float_as_uint(fmaxf(keys[(size_t)b * stride + k], 0.f)) : 0xffffffffu;
    mine += in ? 1 : 0;
  }
  const int total = block_sum_1024(mine, red);
  const int K = min(n, total);
  if (K == 0) {                                    // nothing survived: zeros
    for (int j = tid; j < n * 3; j += DL_THREADS) out[(size_t)b * n * 3 + j] = 0.f;
    return;
  }
  unsigned lo = 0, hi = 0x7f800000u;
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) cnt += key[i] <= mid ? 1 : 0;
    if (block_sum_1024(cnt, red) >= K) hi = mid;
    else lo = mid + 1;
  }
  const unsigned T = lo;
  int below = 0, equal = 0;
#pragma unroll
  for (int i = 0; i < PPT; ++i) below += key[i] < T ? 1 : 0, equal += key[i] == T ? 1 : 0;
  const int nbelow = block_sum_1024(below, red);
  const int eq_before = block_excl_scan_1024(equal, red);
  int quota = K - nbelow - eq_before;
  bool take[PPT];
  int ntake = 0;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    take[i] = key[i] < T;
    if (key[i] == T) {
      take[i] = quota > 0;
      --quota;
    }
    ntake += take[i] ? 1 : 0;
  }
  int pos = block_excl_scan_1024(ntake, red);
#pragma unroll
  for (int i = 0; i < PPT; ++i)
    if (take[i]) sel[pos] = key[i], sel[n + pos] = (unsigned)(tid * PPT + i), ++pos;
  __syncthreads();
  // rank of every selected element among the K selected (LDS broadcast reads), then the gather
  for (int j = tid; j < K; j += DL_THREADS) {
    const unsigned kj = sel[j], ij = sel[n + j];
    int rank = 0;
    for (int q = 0; q < K; ++q) {
      const unsigned kq = sel[q];
      rank += (kq < kj || (kq == kj && sel[n + q] < ij)) ? 1 : 0;
    }
    const float* p = xyz + ((size_t)b * stride + ij) * 3;
    for (int o = rank; o < n; o += K) {            // o > rank only when there are fewer than n survivors
      float* d = out + ((size_t)b * n + o) * 3;
      d[0] = p[0], d[1] = p[1], d[2] = p[2];
    }
  }
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_dropout_local(int b, int p, const float* xyz, const int32_t* nclusters,
                                  const int32_t* seed_rank, const int32_t* sizes, unsigned char* alive,
                                  pdae_stream_t stream) {
  if (b < 0 || p < 0) return bad_arg("dropout_local: negative size");
  if (b == 0 || p == 0) return PDAE_OK;
  if (!xyz || !nclusters || !seed_rank || !sizes || !alive) return bad_arg("dropout_local: null pointer");
  if (p > DL_THREADS * DL_MAXPPT) return unsupported("dropout_local: more than 16384 points per cloud");
  hipStream_t s = as_stream(stream);
  const int ppt = (p + DL_THREADS - 1) / DL_THREADS;
  if (ppt <= 2) hipLaunchKernelGGL(dropout_local_kernel<2>, dim3(b), dim3(DL_THREADS), 0, s, p, xyz, nclusters, seed_rank, sizes, alive);
  else if (ppt <= 4) hipLaunchKernelGGL(dropout_local_kernel<4>, dim3(b), dim3(DL_THREADS), 0, s, p, xyz, nclusters, seed_rank, sizes, alive);
  else if (ppt <= 8) hipLaunchKernelGGL(dropout_local_kernel<8>, dim3(b), dim3(DL_THREADS), 0, s, p, xyz, nclusters, seed_rank, sizes, alive);
  else hipLaunchKernelGGL(dropout_local_kernel<16>, dim3(b), dim3(DL_THREADS), 0, s, p, xyz, nclusters, seed_rank, sizes, alive);
  return check_launch("dropout_local");
}


extern "C" int pdae_pipeline_norm_affine(int b, int p, int out_stride, int normalise, const float* xyz,
                                         const int32_t* nmaps, const float* maps, const float* sigma,
                                         const float* noise, float* out, pdae_stream_t stream) {
  if (b < 0 || p < 0 || out_stride < p) return bad_arg("pipeline_norm_affine: b, p >= 0 and out_stride >= p required");
  if (b == 0 || p == 0) return PDAE_OK;
  if (!xyz || !out || (nmaps && !maps) || (noise && !sigma)) return bad_arg("pipeline_norm_affine: null pointer");
  if (p > DL_THREADS * DL_MAXPPT) return unsupported("pipeline_norm_affine: more than 16384 points per cloud");
  hipStream_t s = as_stream(stream);
  const int ppt = (p + DL_THREADS - 1) / DL_THREADS;
#define PDAE_NA(PPT) hipLaunchKernelGGL(norm_affine_kernel<PPT>, dim3(b), dim3(DL_THREADS), 0, s, p, out_stride, normalise, xyz, nmaps, maps, sigma, noise, out)
  if (ppt <= 2) PDAE_NA(2);
  else if (ppt <= 4) PDAE_NA(4);
  else if (ppt <= 8) PDAE_NA(8);
  else PDAE_NA(16);
#undef PDAE_NA
  return check_launch("pipeline_norm_affine");
}

extern "C" int pdae_pipeline_add_global(int b, int nmax, int stride, int p0, const int32_t* count, const float* u,
                                        float* xyz, pdae_stream_t stream) {
  if (b < 0 || nmax < 0 || p0 < 0 || stride < p0 + nmax) return bad_arg("pipeline_add_global: bad size");
  if (b == 0 || nmax == 0) return PDAE_OK;
  if (!count || !u || !xyz) return bad_arg("pipeline_add_global: null pointer");
  if (b > 65535) return unsupported("pipeline_add_global: b > 65535");
  hipLaunchKernelGGL(add_global_kernel, dim3((nmax + 255) / 256, b), dim3(256), 0, as_stream(stream), nmax, stride, p0,
                     count, u, xyz);
  return check_launch("pipeline_add_global");
}

extern "C" int pdae_pipeline_add_local(int b, int nmax, int stride, int p0, const int32_t* count, const int32_t* seed,
                                       const float* sigma, const float* noise, float* xyz, pdae_stream_t stream) {
  if (b < 0 || nmax < 0 || p0 < 0 || stride < p0 + nmax) return bad_arg("pipeline_add_local: bad size");
  if (b == 0 || nmax == 0) return PDAE_OK;
  if (!count || !seed || !sigma || !noise || !xyz) return bad_arg("pipeline_add_local: null pointer");
  if (b > 65535) return unsupported("pipeline_add_local: b > 65535");
  hipLaunchKernelGGL(add_local_kernel, dim3((nmax + 255) / 256, b), dim3(256), 0, as_stream(stream), nmax, stride, p0,
                     count, seed, sigma, noise, xyz);
  return check_launch("pipeline_add_local");
}

extern "C" int pdae_pipeline_density(int b, int p, int stride, const float* xyz, const float* view, const float* gate,
                                     const float* r, unsigned char* alive, pdae_stream_t stream) {
  if (b < 0 || p < 0 || stride < p) return bad_arg("pipeline_density: bad size");
  if (b == 0 || p == 0) return PDAE_OK;
  if (!xyz || !view || !gate || !r || !alive) return bad_arg("pipeline_density: null pointer");
  if (b > 65535) return unsupported("pipeline_density: b > 65535");
  hipLaunchKernelGGL(density_kernel, dim3((p + 255) / 256, b), dim3(256), 0, as_stream(stream), p, stride, xyz, view, gate,
                     r, alive);
  return check_launch("pipeline_density");
}

extern "C" int pdae_pipeline_subset(int b, int p, int stride, int n, const float* xyz, const unsigned char* alive,
                                    const float* keys, float* out, pdae_stream_t stream) {
  if (b < 0 || p < 0 || n < 0 || stride < p) return bad_arg("pipeline_subset: bad size");
  if (b == 0 || n == 0) return PDAE_OK;
  if (!xyz || !keys || !out) return bad_arg("pipeline_subset: null pointer");
  if (p > DL_THREADS * DL_MAXPPT) return unsupported("pipeline_subset: more than 16384 points per cloud");
  if (n > 16384) return unsupported("pipeline_subset: more than 16384 points per subset");
  hipStream_t s = as_stream(stream);
  const size_t lds = sizeof(unsigned) * 2 * (size_t)n;
  const int ppt = (p + DL_THREADS - 1) / DL_THREADS;
#define PDAE_SS(PPT) hipLaunchKernelGGL(subset_kernel<PPT>, dim3(b), dim3(DL_THREADS), lds, s, p, stride, n, xyz, alive, keys, out)
  if (ppt <= 2) PDAE_SS(2);
  else if (ppt <= 4) PDAE_SS(4);
  else if (ppt <= 8) PDAE_SS(8);
  else PDAE_SS(16);
#undef PDAE_SS
  return check_launch("pipeline_subset");
}
