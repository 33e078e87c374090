// knn.hip -- k nearest neighbours (+ fused grouping) for gfx950.
//
// Semantics: KNN_CUDA 0.2 `knn(ref, query, k)` (third-party, restated in
// oracle/pdae_oracle.c oracle_knn): the k smallest squared distances in
// ascending order, the earlier index first on equal distances, int64 indices,
// returned distance = sqrtf(d2).  Optionally also writes ref[idx] - query, the
// gather + centre subtraction of Group.forward
// (models/PointCAE_transformer.py:79-85).
//
// Design (nothing like the reference's distance matrix + per-thread insertion
// sort, which KNN_CUDA additionally drives from a Python loop over the batch):
//   * one wave per query, QPW consecutive queries of a cloud per wave; the
//     cloud lives in registers (PPL points per lane) and is reused by all QPW
//     queries, so HBM sees each cloud once per wave.
//   * selection is exact on the 64-bit key (bits(d2) << 32 | index) -- unique
//     keys make "ascending, earlier index first" a plain ascending sort.
//       1. every lane takes the min of its PPL distances; the k-th smallest of
//          those 64 lane-minima bounds the k-th smallest distance from above
//          (k <= 64 lanes each contribute one point at or below it);
//       2. candidates with key <= bound are compacted (ballot + mbcnt) into a
//          small LDS staging ring; typically ~1.3 k survive out of N;
//       3. every 64 staged candidates are bitonic-sorted across the lanes and
//          merged (bitonic merge) into the running sorted best-64, after which
//          the bound tightens to best[k-1];
//       4. lanes 0..k-1 hold the answer in order.
#include "common.h"
#include "wave_select.h"

namespace pdae {

template <int PPL>  // points per lane held in registers; n <= 64*PPL
__global__ __launch_bounds__(256) void knn_kernel(int n, int g, int k, int qpw,
                                                  const float* __restrict__ ref_all,
                                                  const float* __restrict__ query_all,
                                                  int64_t* __restrict__ idx_all,
                                                  float* __restrict__ dist_all,
                                                  float* __restrict__ nbr_all) {
  __shared__ unsigned long long stage_all[4][128];
  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  unsigned long long* stage = stage_all[wave];

  const int bi = blockIdx.y;
  const float* ref = ref_all + (size_t)bi * n * 3;
  const float* query = query_all + (size_t)bi * g * 3;
  const int q0 = (blockIdx.x * 4 + wave) * qpw;
  if (q0 >= g) return;
  const int q1 = min(g, q0 + qpw);

  float px[PPL], py[PPL], pz[PPL];
#pragma unroll
  for (int i = 0; i < PPL; ++i) {
    const int p = i * kWave + lane;
    const bool in = p < n;
    px[i] = in ? ref[p * 3 + 0] : 0.f;
    py[i] = in ? ref[p * 3 + 1] : 0.f;
    pz[i] = in ? ref[p * 3 + 2] : 0.f;
  }

  for (int q = q0; q < q1; ++q) {
    const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    float d[PPL];
    float lane_min = __builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      // KNN_CUDA accumulates (ref - query)^2 over the dims in order from 0
      d[i] = sqdist(px[i], py[i], pz[i], qx, qy, qz);
      if (i * kWave + lane < n) lane_min = fminf(lane_min, d[i]);
    }
    const float t = wave_kth_smallest(lane_min, k);
    KnnSelect st;
    st.best = kKeyMax;
    st.bound = ((unsigned long long)__float_as_uint(t) << 32) | 0xffffffffull;
    st.staged = 0;
    st.have_best = false;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const int p = i * kWave + lane;
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(d[i]) << 32) | (unsigned)p;
      knn_offer(st, stage, key, p < n, k);
    }
    if (st.staged > 0 || !st.have_best) knn_flush(st, stage, st.staged, k);

    if (lane < k) {
      const int id = (int)(st.best & 0xffffffffull);
      const float d2 = __uint_as_float((unsigned)(st.best >> 32));
      const size_t o = ((size_t)bi * g + q) * k + lane;
      idx_all[o] = id;
      if (dist_all) dist_all[o] = sqrtf(d2);
      if (nbr_all) {
        nbr_all[o * 3 + 0] = ref[id * 3 + 0] - qx;
        nbr_all[o * 3 + 1] = ref[id * 3 + 1] - qy;
        nbr_all[o * 3 + 2] = ref[id * 3 + 2] - qz;
      }
    }
  }
}

// Large clouds: points streamed from global memory (L2-resident) per query.
__global__ __launch_bounds__(256) void knn_stream_kernel(int n, int g, int k, int qpw,
                                                         const float* __restrict__ ref_all,
                                                         const float* __restrict__ query_all,
                                                         int64_t* __restrict__ idx_all,
                                                         float* __restrict__ dist_all,
                                                         float* __restrict__ nbr_all) {
  __shared__ unsigned long long stage_all[4][128];
  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  unsigned long long* stage = stage_all[wave];
  const int bi = blockIdx.y;
  const float* ref = ref_all + (size_t)bi * n * 3;
  const float* query = query_all + (size_t)bi * g * 3;
  const int q0 = (blockIdx.x * 4 + wave) * qpw;
  if (q0 >= g) return;
  const int q1 = min(g, q0 + qpw);
  for (int q = q0; q < q1; ++q) {
    const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    float lane_min = __builtin_huge_valf();
    for (int p = lane; p < n; p += kWave)
      lane_min = fminf(lane_min, sqdist(ref[p * 3 + 0], ref[p * 3 + 1], ref[p * 3 + 2], qx, qy, qz));
    const float t = wave_kth_smallest(lane_min, k);
    KnnSelect st;
    st.best = kKeyMax;
    st.bound = ((unsigned long long)__float_as_uint(t) << 32) | 0xffffffffull;
    st.staged = 0;
    st.have_best = false;
    for (int p0 = 0; p0 < n; p0 += kWave) {
      const int p = p0 + lane;
      const bool in = p < n;
      const float dd =
          in ? sqdist(ref[p * 3 + 0], ref[p * 3 + 1], ref[p * 3 + 2], qx, qy, qz) : 0.f;
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)p;
      knn_offer(st, stage, key, in, k);
    }
    if (st.staged > 0 || !st.have_best) knn_flush(st, stage, st.staged, k);
    if (lane < k) {
      const int id = (int)(st.best & 0xffffffffull);
      const float d2 = __uint_as_float((unsigned)(st.best >> 32));
      const size_t o = ((size_t)bi * g + q) * k + lane;
      idx_all[o] = id;
      if (dist_all) dist_all[o] = sqrtf(d2);
      if (nbr_all) {
        nbr_all[o * 3 + 0] = ref[id * 3 + 0] - qx;
        nbr_all[o * 3 + 1] = ref[id * 3 + 1] - qy;
        nbr_all[o * 3 + 2] = ref[id * 3 + 2] - qz;
      }
    }
  }
}

}  // namespace pdae

extern "C" int pdae_knn(int b, int n, int g, int k, const float* ref, const float* query,
                        int64_t* idx, float* dist, float* nbr, pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || n <= 0 || g < 0 || k <= 0) return bad_arg("knn: b>=0, n>0, g>=0, k>0 required");
  if (k > n) return bad_arg("knn: k > n");
  if (k > 64) return unsupported("knn: k > 64 not implemented");
  if (b == 0 || g == 0) return PDAE_OK;
  if (!ref || !query || !idx) return bad_arg("knn: null pointer");
  if (b > 65535) return unsupported("knn: b > 65535");
  hipStream_t s = as_stream(stream);
  // queries per wave: enough waves to cover the chip (1024 SIMDs) when the
  // batch allows it, while amortising the register load of the cloud.
  int qpw = 8;
  while (qpw > 1 && (long long)b * ((g + qpw - 1) / qpw) < 2048) qpw >>= 1;
  const int waves = (g + qpw - 1) / qpw;
  dim3 grid((waves + 3) / 4, b), block(256);
#define PDAE_KNN_LAUNCH(PPL) \
  hipLaunchKernelGGL((knn_kernel<PPL>), grid, block, 0, s, n, g, k, qpw, ref, query, idx, dist, nbr)
  if (n <= 64) PDAE_KNN_LAUNCH(1);
  else if (n <= 128) PDAE_KNN_LAUNCH(2);
  else if (n <= 256) PDAE_KNN_LAUNCH(4);
  else if (n <= 512) PDAE_KNN_LAUNCH(8);
  else if (n <= 1024) PDAE_KNN_LAUNCH(16);
  else if (n <= 2048) PDAE_KNN_LAUNCH(32);
  else
    hipLaunchKernelGGL(knn_stream_kernel, grid, block, 0, s, n, g, k, qpw, ref, query, idx, dist,
                       nbr);
#undef PDAE_KNN_LAUNCH
  return check_launch("knn");
}
