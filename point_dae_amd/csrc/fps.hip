// fps.hip -- farthest point sampling for gfx950.
//
// Semantics: extensions/pointnet2/_ext_src/src/sampling_gpu.cu:72-176 of the
// reference (restated in oracle/pdae_oracle.c fps_one).  Design is not the
// reference's: one workgroup per cloud keeps the cloud AND the running
// min-distances in registers for all m-1 serial iterations (the reference
// re-reads both from global memory every iteration), the per-iteration argmax
// is a single 64-bit key max -- DPP butterfly inside a wave, one LDS slot per
// wave across waves, ONE barrier per iteration (double-buffered slots) instead
// of the reference's 10 __syncthreads.
//
// Key = (bits(d2) << 32) | ((0xFFFF - rank(k)) << 16) | k, where rank(k) is the
// position of point k in the reference's tie order: its bs-thread block scans
// k = tid, tid+bs, ... with strict '>' (first k wins inside a thread) and the
// shared-memory tree pairs (tid, tid+s), s = bs/2..1, keeping the lower slot,
// so between threads the winner has a 0 at the lowest bit where the tids
// differ: ascending BIT-REVERSED (k % bs), then ascending k / bs.  d2 >= 0
// always, so its bit pattern orders like the float.  Skipped points
// (x*x+y*y+z*z <= 1e-3, evaluated float-vs-double like the source) and padding
// get key 0; if every point is skipped the reference returns index 0 and so
// does key 0.
#include <cmath>

#include "common.h"

namespace pdae {

template <int T, int P>
__global__ __launch_bounds__(T) void fps_kernel(int n, int m, int bs_log2, int cols,
                                                const float* __restrict__ dataset,
                                                int32_t* __restrict__ idxs,
                                                float* __restrict__ centres, int use_lds) {
  constexpr int W = T / kWave;
  extern __shared__ float lds_xyz[];  // n*3 floats when use_lds
  __shared__ unsigned long long slots[2][W > 1 ? W : 1];

  const int tid = threadIdx.x;
  const float* cloud = dataset + (size_t)blockIdx.x * n * 3;
  idxs += (size_t)blockIdx.x * m;
  if (centres) centres += (size_t)blockIdx.x * m * 3;

  float px[P], py[P], pz[P], temp[P];
  unsigned low[P];
  const int bs_mask = (1 << bs_log2) - 1;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int k = tid + i * T;
    float x = 0.f, y = 0.f, z = 0.f;
    bool valid = false;
    if (k < n) {
      x = cloud[k * 3 + 0];
      y = cloud[k * 3 + 1];
      z = cloud[k * 3 + 2];
      const float mag = (x * x) + (y * y) + (z * z);
      valid = !((double)mag <= 1e-3);
      if (use_lds) {
        lds_xyz[k * 3 + 0] = x;
        lds_xyz[k * 3 + 1] = y;
        lds_xyz[k * 3 + 2] = z;
      }
    }
    px[i] = x;
    py[i] = y;
    pz[i] = z;
    // skipped / padded points: temp = 0 keeps d2 = +0 and low = 0 -> key 0
    temp[i] = valid ? 1e10f : 0.f;
    const unsigned rev =
        bs_log2 ? (__brev((unsigned)(k & bs_mask)) >> (32 - bs_log2)) : 0u;
    const unsigned rank = rev * (unsigned)cols + (unsigned)(k >> bs_log2);
    low[i] = valid ? (((0xFFFFu - rank) << 16) | (unsigned)k) : 0u;
  }
  if (m <= 0) return;
  int old = 0;
  if (tid == 0) {
    idxs[0] = 0;
    if (centres) {
      centres[0] = cloud[0];
      centres[1] = cloud[1];
      centres[2] = cloud[2];
    }
  }
  if (use_lds) __syncthreads();

  for (int j = 1; j < m; ++j) {
    float x1, y1, z1;
    if (use_lds) {
      x1 = lds_xyz[old * 3 + 0];
      y1 = lds_xyz[old * 3 + 1];
      z1 = lds_xyz[old * 3 + 2];
    } else {
      x1 = cloud[old * 3 + 0];
      y1 = cloud[old * 3 + 1];
      z1 = cloud[old * 3 + 2];
    }
    unsigned long long best = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const float d = sqdist(px[i], py[i], pz[i], x1, y1, z1);
      const float d2 = fminf(d, temp[i]);
      temp[i] = d2;
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(d2) << 32) | low[i];
      best = max_u64(best, key);
    }
    best = wave_max_u64(best);
    if (W > 1) {
      if (lane_id() == 0) slots[j & 1][tid / kWave] = best;
      __syncthreads();
      unsigned long long r = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) r = max_u64(r, slots[j & 1][w]);
      best = r;
    }
    old = (int)(best & 0xFFFFull);
    if (tid == 0) {
      idxs[j] = old;
      if (centres) {
        centres[j * 3 + 0] = cloud[old * 3 + 0];
        centres[j * 3 + 1] = cloud[old * 3 + 1];
        centres[j * 3 + 2] = cloud[old * 3 + 2];
      }
    }
  }
}

// cuda_utils.h:15-21 opt_n_threads -- the reference's block size decides the
// tie order, so it is computed the same way (double log on the host).
static int ref_block_size(int work_size) {
  const int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

template <int T, int P>
static void launch_fps(int b, int n, int m, int bs_log2, int cols, const float* dataset,
                       int32_t* idxs, float* centres, hipStream_t s) {
  const size_t lds_bytes = (size_t)n * 3 * sizeof(float);
  const int use_lds = lds_bytes <= 96 * 1024;
  hipLaunchKernelGGL((fps_kernel<T, P>), dim3(b), dim3(T), use_lds ? lds_bytes : 0, s, n, m,
                     bs_log2, cols, dataset, idxs, centres, use_lds);
}

}  // namespace pdae

extern "C" int pdae_furthest_point_sampling(int b, int n, int m, const float* dataset,
                                            int32_t* idxs, float* centres,
                                            pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || n <= 0 || m < 0) return bad_arg("fps: b>=0, n>0, m>=0 required");
  if (b == 0 || m == 0) return PDAE_OK;
  if (!dataset || !idxs) return bad_arg("fps: null pointer");
  if (n > 32768) return unsupported("fps: n > 32768 not implemented");
  const int bs = ref_block_size(n);
  int bs_log2 = 0;
  while ((1 << bs_log2) < bs) ++bs_log2;
  const int cols = (n + bs - 1) / bs;
  hipStream_t s = as_stream(stream);
  // registers hold T*P >= n points; 4 points per lane keeps one wave per SIMD
  // busy for N=1024 (T=256) while leaving the barrier at 4 waves.
  if (n <= 64) launch_fps<64, 1>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 128) launch_fps<64, 2>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 256) launch_fps<128, 2>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 512) launch_fps<256, 2>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 1024) launch_fps<256, 4>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 2048) launch_fps<512, 4>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 4096) launch_fps<1024, 4>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 8192) launch_fps<1024, 8>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else if (n <= 16384) launch_fps<512, 32>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  else launch_fps<1024, 32>(b, n, m, bs_log2, cols, dataset, idxs, centres, s);
  return check_launch("fps");
}
