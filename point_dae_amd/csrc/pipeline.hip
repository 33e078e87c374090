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
