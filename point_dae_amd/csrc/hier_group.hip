// hier_group.hip -- the index bookkeeping of Point-M2AE's multi-scale token pyramid.
//
// Reference: models/Point_M2AE_modules.py:219-248 (Group: FPS centres + kNN patches, returns the
// neighbourhoods, the centres AND the flat neighbour indices `idx + b * N`) and models/Point_M2AE.py:
// 245-263 (three Group levels, level i > 0 groups the centres of level i - 1), :107-117 (multi-scale
// masking: a coarse token's visibility is pushed down to the finer tokens it was grouped from), :132
// (token merging: the finer level's visible tokens gathered by the same flat indices).
//
// FPS and kNN are the kernels of fps.hip / knn.hip; what is left per level is byte / index work on
// (B, G, k) int64 indices -- HBM-bound, one pass each, written as two small kernels instead of the
// reference's arange / add / view / boolean-multiply / scatter chain (6 elementwise launches per level).
#include "common.h"

namespace pdae {

// idx (b, g*k) int64, local to each cloud  ->  flat (b*g*k) = idx + cloud * n   (16-byte accesses)
__global__ __launch_bounds__(256) void flatten_group_index_kernel(long long total2, int per_cloud2, long long n,
                                                                  const longlong2* __restrict__ idx,
                                                                  longlong2* __restrict__ flat) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total2) return;
  const long long base = (i / per_cloud2) * n;
  longlong2 v = idx[i];
  v.x += base, v.y += base;
  flat[i] = v;
}

__global__ __launch_bounds__(256) void flatten_group_index_tail_kernel(long long total, long long per_cloud, long long n,
                                                                       const long long* __restrict__ idx,
                                                                       long long* __restrict__ flat) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < total) flat[i] = idx[i] + (i / per_cloud) * n;
}

// Multi-scale masking, one level down (Point_M2AE.py:112-117):
//     idx_masked = ~parent_masked[:, None] * idx            # masked parents contribute index 0 (!)
//     child_masked = ones(C).scatter(0, idx_masked, 0)
// i.e. every child of a VISIBLE parent becomes visible, and -- the reference's quirk, kept -- flat child 0
// becomes visible as soon as any parent is masked (its zeroed indices all land on element 0).
// child_masked is pre-filled with 1 by the entry; all writers store 0, so the races are benign.
__global__ __launch_bounds__(256) void mask_propagate_kernel(long long total, int k,
                                                             const unsigned char* __restrict__ parent_masked,
                                                             const long long* __restrict__ flat_idx,
                                                             unsigned char* __restrict__ child_masked) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const bool masked = parent_masked[i / k] != 0;
  child_masked[masked ? 0 : flat_idx[i]] = 0;
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_flatten_group_index(int b, int n, int gk, const int64_t* idx, int64_t* flat, pdae_stream_t stream) {
  if (b < 0 || n <= 0 || gk < 0) return bad_arg("flatten_group_index: b >= 0, n > 0, g*k >= 0 required");
  if (b == 0 || gk == 0) return PDAE_OK;
  if (!idx || !flat) return bad_arg("flatten_group_index: null pointer");
  hipStream_t s = as_stream(stream);
  const long long total = (long long)b * gk;
  if (gk % 2 == 0 && ((uintptr_t)idx % 16 == 0) && ((uintptr_t)flat % 16 == 0)) {
    const long long t2 = total / 2;
    hipLaunchKernelGGL(flatten_group_index_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, t2, gk / 2,
                       (long long)n, reinterpret_cast<const longlong2*>(idx), reinterpret_cast<longlong2*>(flat));
  } else {
    hipLaunchKernelGGL(flatten_group_index_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total,
                       (long long)gk, (long long)n, reinterpret_cast<const long long*>(idx),
                       reinterpret_cast<long long*>(flat));
  }
  return check_launch("flatten_group_index");
}

extern "C" int pdae_mask_propagate(int parents, int k, int children, const uint8_t* parent_masked,
                                   const int64_t* flat_idx, uint8_t* child_masked, pdae_stream_t stream) {
  if (parents < 0 || k <= 0 || children <= 0) return bad_arg("mask_propagate: parents >= 0, k > 0, children > 0 required");
  if (!child_masked) return bad_arg("mask_propagate: null pointer");
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(child_masked, 1, (size_t)children, s) != hipSuccess) return check_launch("mask_propagate");
  if (parents == 0) return PDAE_OK;
  if (!parent_masked || !flat_idx) return bad_arg("mask_propagate: null pointer");
  const long long total = (long long)parents * k;
  hipLaunchKernelGGL(mask_propagate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, k,
                     parent_masked, reinterpret_cast<const long long*>(flat_idx), child_masked);
  return check_launch("mask_propagate");
}
