// wave_select.h -- exact k-smallest selection of 64-bit keys by one wave (knn.hip's design notes, steps 1-4):
// lane minima bound the k-th smallest from above, survivors are compacted into an LDS staging ring, every 64 staged
// keys are bitonic-sorted across the lanes and merged into the running sorted best-64.  Shared by knn.hip (xyz
// neighbours) and dgcnn.hip (feature-space neighbours from Gram rows).
#pragma once
#include "common.h"

namespace pdae {

constexpr unsigned long long kKeyMax = ~0ull;

// ascending bitonic sort of one key per lane over the 64 lanes
__device__ __forceinline__ unsigned long long wave_sort_u64(unsigned long long v) {
  const int lane = lane_id();
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      const unsigned long long o = shfl_u64(v, lane ^ stride);
      const bool up = ((lane & size) == 0);          // ascending block?
      const bool lower = ((lane & stride) == 0);      // lower partner?
      const bool take_min = (up == lower);
      v = take_min ? min_u64(v, o) : max_u64(v, o);
    }
  }
  return v;
}

// a, b ascending over the lanes -> the 64 smallest of the union, ascending
__device__ __forceinline__ unsigned long long wave_merge_low_u64(unsigned long long a,
                                                                 unsigned long long b) {
  const int lane = lane_id();
  unsigned long long v = min_u64(a, shfl_u64(b, 63 - lane));  // bitonic, 64 smallest
#pragma unroll
  for (int stride = 32; stride >= 1; stride >>= 1) {
    const unsigned long long o = shfl_u64(v, lane ^ stride);
    v = ((lane & stride) == 0) ? min_u64(v, o) : max_u64(v, o);
  }
  return v;
}

struct KnnSelect {
  unsigned long long best;   // lane j: j-th smallest key so far
  unsigned long long bound;  // wave-uniform: keys above it cannot be in the answer
  int staged;                // wave-uniform count of staged candidates
  bool have_best;
};

// Sort + merge one staged chunk (up to 64 keys taken from stage[0..63]).
__device__ __forceinline__ void knn_flush(KnnSelect& st, unsigned long long* stage, int count,
                                          int k) {
  const int lane = lane_id();
  unsigned long long v = lane < count ? stage[lane] : kKeyMax;
  v = wave_sort_u64(v);
  st.best = st.have_best ? wave_merge_low_u64(st.best, v) : v;
  st.have_best = true;
  const unsigned long long kth = shfl_u64(st.best, k - 1);
  st.bound = min_u64(st.bound, kth);
}

// Offer one candidate per lane (pred false = none).  stage holds 128 keys.
__device__ __forceinline__ void knn_offer(KnnSelect& st, unsigned long long* stage,
                                          unsigned long long key, bool pred, int k) {
  pred = pred && (key <= st.bound);
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return;
  const int pos = st.staged + (int)__builtin_amdgcn_mbcnt_hi(
                                  (unsigned)(mask >> 32),
                                  __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  if (pred) stage[pos] = key;
  st.staged += __popcll(mask);
  if (st.staged >= 64) {
    knn_flush(st, stage, 64, k);
    // move the overflow (< 64 keys) to the front
    const int lane = lane_id();
    const int rest = st.staged - 64;
    const unsigned long long mv = lane < rest ? stage[64 + lane] : 0;
    if (lane < rest) stage[lane] = mv;
    st.staged = rest;
  }
}

// k-th smallest (1-based k) of one float per lane, wave-uniform result.
__device__ __forceinline__ float wave_kth_smallest(float mine, int k) {
  int cnt_le = 0;
#pragma unroll 8
  for (int l = 0; l < 64; ++l) {
    const float other = __builtin_bit_cast(
        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), l));
    cnt_le += (other <= mine) ? 1 : 0;
  }
  // smallest value that has at least k lane values at or below it
  return wave_min_f32(cnt_le >= k ? mine : __builtin_huge_valf());
}

}  // namespace pdae
