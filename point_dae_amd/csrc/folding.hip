// folding.hip -- the first layer of the FoldingNet stage of Point_CAE_PointNetv2
// (models/PointCAE_pointnetv2.py:157-167: folding2[0] on [grid(2) | coarse point(3) | global
// feature(1024)] for every one of 16 grid cells x 1024 coarse points x B clouds).
//
// The reference materialises the (B, 1029, 16384) input and runs one Conv1d over it.  The conv is
// linear in the three column blocks, so its output is a per-cloud term a[b] (global feature, bias),
// a per-coarse-point term p[b,c] and a per-grid-cell term gd[g] -- three small GEMMs -- and the
// (B*1024*16, 512) activation is
//     h[(b,c,g), :] = relu((a[b] + p[b,c]) + gd[g])
// written here in ONE pass (4.3 GB at B = 128: the pass is the HBM write).  Backward: the masked
// gradient dpre[(b,c,g), :] arrives from the next layer's data-gradient GEMM (its epilogue applies
// the ReLU mask); one pass over it produces dp[b,c] = sum_g dpre and per-block partials of
// dgd[g] = sum_{b,c} dpre (added in block order by the caller: no atomics); da[b] = sum_c dp[b,c]
// is a reduction of the small dp.
#include <cstdlib>

#include "common.h"

namespace pdae {

// one float4 of the output per thread; consecutive threads walk a row
__global__ __launch_bounds__(256) void fold_input_kernel(long long n4, int coarse, int cells, int C4,
                                                         const float4* __restrict__ a,
                                                         const float4* __restrict__ p,
                                                         const float4* __restrict__ gd,
                                                         float4* __restrict__ h) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long row = i / C4;
  const int q = (int)(i - row * C4);
  const long long bc = row / cells;
  const int g = (int)(row - bc * cells);
  const long long b = bc / coarse;
  const float4 va = a[b * C4 + q], vp = p[bc * C4 + q], vg = gd[(long long)g * C4 + q];
  float4 o;
  o.x = (va.x + vp.x) + vg.x, o.y = (va.y + vp.y) + vg.y;
  o.z = (va.z + vp.z) + vg.z, o.w = (va.w + vp.w) + vg.w;
  o.x = o.x > 0.f ? o.x : 0.f, o.y = o.y > 0.f ? o.y : 0.f;
  o.z = o.z > 0.f ? o.z : 0.f, o.w = o.w > 0.f ? o.w : 0.f;
  h[i] = o;
}

// The same pass with the per-pair work hoisted (round 6): a thread owns one channel quad of one (cloud, coarse point) pair,
// adds a[b] + p[b,c] ONCE and walks the pair's cells -- 1 + 1/cells loads per store instead of 3, `cells` stores in flight
// per thread; the one-float4-per-thread form above ran the 4.3 GB write at 3.4 TB/s (a million 256-thread blocks).
__global__ __launch_bounds__(256) void fold_input_pairs_kernel(long long pairs4, int coarse, int cells, int C4,
                                                               const float4* __restrict__ a, const float4* __restrict__ p,
                                                               const float4* __restrict__ gd, float4* __restrict__ h) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pairs4) return;
  const long long bc = i / C4;
  const int q = (int)(i - bc * C4);
  const long long b = bc / coarse;
  const float4 va = a[b * C4 + q], vp = p[i];
  const float4 ap = make_float4(va.x + vp.x, va.y + vp.y, va.z + vp.z, va.w + vp.w);
  float4* dst = h + bc * cells * C4 + q;
#pragma unroll 4
  for (int g = 0; g < cells; ++g) {
    const float4 vg = gd[(long long)g * C4 + q];
    float4 o = make_float4(ap.x + vg.x, ap.y + vg.y, ap.z + vg.z, ap.w + vg.w);
    o.x = o.x > 0.f ? o.x : 0.f, o.y = o.y > 0.f ? o.y : 0.f;
    o.z = o.z > 0.f ? o.z : 0.f, o.w = o.w > 0.f ? o.w : 0.f;
    dst[(long long)g * C4] = o;
  }
}

// the same first layer with a PER-ROW term instead of the per-cloud / per-cell ones (the published variant's second
// folding stage, PointCAE_transformer.py:1050-1059: the first fold's points enter the second): h[r] = relu(row[r] + p[r / cells])
__global__ __launch_bounds__(256) void fold_input_rows_kernel(long long n4, int cells, int C4, const float4* __restrict__ row,
                                                              const float4* __restrict__ p, float4* __restrict__ h) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long r = i / C4;
  const int q = (int)(i - r * C4);
  const float4 vr = row[i], vp = p[(r / cells) * C4 + q];
  float4 o = make_float4(vr.x + vp.x, vr.y + vp.y, vr.z + vp.z, vr.w + vp.w);
  o.x = o.x > 0.f ? o.x : 0.f, o.y = o.y > 0.f ? o.y : 0.f;
  o.z = o.z > 0.f ? o.z : 0.f, o.w = o.w > 0.f ? o.w : 0.f;
  h[i] = o;
}

// block = FG_PAIRS consecutive (cloud, coarse point) pairs; thread = one channel quad x one group of
// 256 / C4 ... rows.  Layout: tid % C4 = channel quad, tid / C4 = row phase; a phase owns the cells
// g = phase, phase + PH, ... so its dgd accumulators stay in registers across the block's pairs.
constexpr int FG_PAIRS = 64;
template <int CPT>   // cells per thread
__global__ __launch_bounds__(256) void fold_input_grad_kernel(long long pairs, int cells, int C4,
                                                              const float4* __restrict__ dpre,
                                                              float4* __restrict__ dp,
                                                              float4* __restrict__ dgd_part) {
  extern __shared__ float4 red[];                 // [phases][C4]
  const int PH = 256 / C4;                        // row phases (C4 need not divide 256: the threads of an
  const int q = threadIdx.x % C4, ph = threadIdx.x / C4;   // incomplete last phase only keep the barriers)
  const bool active = ph < PH;
  const long long p0 = (long long)blockIdx.x * FG_PAIRS;
  const long long p1 = p0 + FG_PAIRS < pairs ? p0 + FG_PAIRS : pairs;
  float4 acc[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long pr = p0; pr < p1; ++pr) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int g = ph + k * PH;
      const float4 v = (active && g < cells) ? dpre[(pr * cells + g) * C4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[k].x += v.x, acc[k].y += v.y, acc[k].z += v.z, acc[k].w += v.w;
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    // the pair's sum over its cells: the phases meet in LDS, added in phase order
    if (active) red[ph * C4 + q] = s;
    __syncthreads();
    if (ph == 0) {
      float4 t = red[q];
      for (int k = 1; k < PH; ++k) {
        const float4 u = red[k * C4 + q];
        t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
      }
      dp[pr * C4 + q] = t;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int g = ph + k * PH;
    if (active && g < cells) dgd_part[((long long)blockIdx.x * cells + g) * C4 + q] = acc[k];
  }
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_fold_input(int clouds, int coarse, int cells, int C, const float* a, const float* p,
                               const float* gd, float* h, pdae_stream_t stream) {
  if (clouds < 0 || coarse <= 0 || cells <= 0 || C <= 0 || C % 4 != 0)
    return bad_arg("fold_input: C must be a positive multiple of 4");
  if (clouds == 0) return PDAE_OK;
  if (!a || !p || !gd || !h) return bad_arg("fold_input: null pointer");
  const long long n4 = (long long)clouds * coarse * cells * (C / 4);
  if ((n4 + 255) / 256 > 0x7fffffffLL) return unsupported("fold_input: too many elements");
  static const bool per_element = [] { const char* e = getenv("PDAE_FOLD_INPUT"); return e && e[0] == 'e'; }();   // (A/B)
  if (per_element) {
    hipLaunchKernelGGL(fold_input_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4,
                       coarse, cells, C / 4, reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(p),
                       reinterpret_cast<const float4*>(gd), reinterpret_cast<float4*>(h));
  } else {
    const long long pairs4 = (long long)clouds * coarse * (C / 4);
    hipLaunchKernelGGL(fold_input_pairs_kernel, dim3((unsigned)((pairs4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       pairs4, coarse, cells, C / 4, reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(p),
                       reinterpret_cast<const float4*>(gd), reinterpret_cast<float4*>(h));
  }
  return check_launch("fold_input");
}

extern "C" int pdae_fold_input_rows(long long pairs, int cells, int C, const float* row, const float* p, float* h,
                                    pdae_stream_t stream) {
  if (pairs < 0 || cells <= 0 || C <= 0 || C % 4 != 0) return bad_arg("fold_input_rows: C must be a positive multiple of 4");
  if (pairs == 0) return PDAE_OK;
  if (!row || !p || !h) return bad_arg("fold_input_rows: null pointer");
  const long long n4 = pairs * cells * (C / 4);
  if ((n4 + 255) / 256 > 0x7fffffffLL) return unsupported("fold_input_rows: too many elements");
  hipLaunchKernelGGL(fold_input_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4, cells,
                     C / 4, reinterpret_cast<const float4*>(row), reinterpret_cast<const float4*>(p),
                     reinterpret_cast<float4*>(h));
  return check_launch("fold_input_rows");
}

extern "C" int pdae_fold_input_grad_parts(int clouds, int coarse) {
  const long long pairs = (long long)clouds * coarse;
  return (int)((pairs + FG_PAIRS - 1) / FG_PAIRS);
}

extern "C" int pdae_fold_input_grad(int clouds, int coarse, int cells, int C, const float* dpre, float* dp,
                                    float* dgd_part, pdae_stream_t stream) {
  if (clouds < 0 || coarse <= 0 || cells <= 0 || C <= 0 || C % 4 != 0 || C > 1024)
    return bad_arg("fold_input_grad: C must be a positive multiple of 4, at most 1024");
  if (clouds == 0) return PDAE_OK;
  if (!dpre || !dp || !dgd_part) return bad_arg("fold_input_grad: null pointer");
  const int C4 = C / 4, PH = 256 / C4;
  const int cpt = (cells + PH - 1) / PH;
  const long long pairs = (long long)clouds * coarse;
  const unsigned grid = (unsigned)((pairs + FG_PAIRS - 1) / FG_PAIRS);
  const size_t lds = sizeof(float4) * 256;
  hipStream_t s = as_stream(stream);
  auto a4 = reinterpret_cast<const float4*>(dpre);
  auto b4 = reinterpret_cast<float4*>(dp);
  auto c4 = reinterpret_cast<float4*>(dgd_part);
  if (cpt <= 1) hipLaunchKernelGGL(fold_input_grad_kernel<1>, dim3(grid), dim3(256), lds, s, pairs, cells, C4, a4, b4, c4);
  else if (cpt <= 2) hipLaunchKernelGGL(fold_input_grad_kernel<2>, dim3(grid), dim3(256), lds, s, pairs, cells, C4, a4, b4, c4);
  else if (cpt <= 4) hipLaunchKernelGGL(fold_input_grad_kernel<4>, dim3(grid), dim3(256), lds, s, pairs, cells, C4, a4, b4, c4);
  else if (cpt <= 8) hipLaunchKernelGGL(fold_input_grad_kernel<8>, dim3(grid), dim3(256), lds, s, pairs, cells, C4, a4, b4, c4);
  else if (cpt <= 16) hipLaunchKernelGGL(fold_input_grad_kernel<16>, dim3(grid), dim3(256), lds, s, pairs, cells, C4, a4, b4, c4);
  else if (cpt <= 20) hipLaunchKernelGGL(fold_input_grad_kernel<20>, dim3(grid), dim3(256), lds, s, pairs, cells, C4, a4, b4, c4);
  else return unsupported("fold_input_grad: more than 20 cells per row phase");
  return check_launch("fold_input_grad");
}

// ---- backward of the stage's last layer (512 -> 3, zero-padded to 4 outputs) in ONE pass over h2 ----------
// dy (rows, 4) is the gradient of the offsets, h2 (rows, C) the kept ReLU output of the middle layer, W (4, C):
//   d2[r][c]   = h2[r][c] > 0 ? sum_j dy[r][j] W[j][c] : 0        (gradient of the middle layer's pre-activation)
//   dW[j][c]   = sum_r dy[r][j] h2[r][c]                           (per-block partials, added in block order)
// The GEMM formulation reads h2 twice (the K = 4 data-gradient product with the ReLU mask, the N = 4 weight
// gradient on 128-wide MFMA tiles: 1.35 + 2.2 ms at 2.1 M rows); this pass is its 8.6 GB of traffic.
namespace pdae {

constexpr int FO_ROWS = 1024;   // rows per block
__global__ __launch_bounds__(256) void fold_out_backward_kernel(long long rows, int C4, const float4* __restrict__ dy,
                                                                const float4* __restrict__ h2,
                                                                const float4* __restrict__ W,
                                                                float4* __restrict__ d2, float4* __restrict__ part) {
  extern __shared__ float4 fo_red[];              // [phases][4][C4]
  const int PH = 256 / C4;
  const int q = threadIdx.x % C4, ph = threadIdx.x / C4;
  const bool active = ph < PH;
  const float4 w0 = W[q], w1 = W[C4 + q], w2 = W[2 * C4 + q], w3 = W[3 * C4 + q];
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  const long long r0 = (long long)blockIdx.x * FO_ROWS;
  const long long r1 = r0 + FO_ROWS < rows ? r0 + FO_ROWS : rows;
#pragma unroll 4
  for (long long r = active ? r0 + ph : r1; r < r1; r += PH) {
    const float4 g = dy[r];                       // the row's four output gradients (same address across q: broadcast)
    const float4 h = h2[r * C4 + q];
    float4 o;
    o.x = h.x > 0.f ? ((g.x * w0.x + g.y * w1.x) + (g.z * w2.x + g.w * w3.x)) : 0.f;
    o.y = h.y > 0.f ? ((g.x * w0.y + g.y * w1.y) + (g.z * w2.y + g.w * w3.y)) : 0.f;
    o.z = h.z > 0.f ? ((g.x * w0.z + g.y * w1.z) + (g.z * w2.z + g.w * w3.z)) : 0.f;
    o.w = h.w > 0.f ? ((g.x * w0.w + g.y * w1.w) + (g.z * w2.w + g.w * w3.w)) : 0.f;
    d2[r * C4 + q] = o;
    a0.x += g.x * h.x, a0.y += g.x * h.y, a0.z += g.x * h.z, a0.w += g.x * h.w;
    a1.x += g.y * h.x, a1.y += g.y * h.y, a1.z += g.y * h.z, a1.w += g.y * h.w;
    a2.x += g.z * h.x, a2.y += g.z * h.y, a2.z += g.z * h.z, a2.w += g.z * h.w;
    a3.x += g.w * h.x, a3.y += g.w * h.y, a3.z += g.w * h.z, a3.w += g.w * h.w;
  }
  if (active) {
    fo_red[(ph * 4 + 0) * C4 + q] = a0;
    fo_red[(ph * 4 + 1) * C4 + q] = a1;
    fo_red[(ph * 4 + 2) * C4 + q] = a2;
    fo_red[(ph * 4 + 3) * C4 + q] = a3;
  }
  __syncthreads();
  if (ph == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 t = fo_red[j * C4 + q];
      for (int k = 1; k < PH; ++k) {
        const float4 u = fo_red[(k * 4 + j) * C4 + q];
        t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
      }
      part[((size_t)blockIdx.x * 4 + j) * C4 + q] = t;
    }
  }
}

}  // namespace pdae

extern "C" int pdae_fold_out_backward_parts(long long rows) { return (int)((rows + pdae::FO_ROWS - 1) / pdae::FO_ROWS); }

extern "C" int pdae_fold_out_backward(long long rows, int C, const float* dy, const float* h2, const float* W,
                                      float* d2, float* part, pdae_stream_t stream) {
  if (rows < 0 || C <= 0 || C % 4 != 0 || C > 1024)
    return bad_arg("fold_out_backward: C must be a positive multiple of 4, at most 1024");
  if (rows == 0) return PDAE_OK;
  if (!dy || !h2 || !W || !d2 || !part) return bad_arg("fold_out_backward: null pointer");
  const int C4 = C / 4;
  const long long blocks = (rows + FO_ROWS - 1) / FO_ROWS;
  if (blocks > 0x7fffffffLL) return unsupported("fold_out_backward: too many rows");
  hipLaunchKernelGGL(fold_out_backward_kernel, dim3((unsigned)blocks), dim3(256), sizeof(float4) * 4 * 256,
                     as_stream(stream), rows, C4, reinterpret_cast<const float4*>(dy),
                     reinterpret_cast<const float4*>(h2), reinterpret_cast<const float4*>(W),
                     reinterpret_cast<float4*>(d2), reinterpret_cast<float4*>(part));
  return check_launch("fold_out_backward");
}
