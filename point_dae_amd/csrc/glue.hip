// glue.hip -- three small launches that replace groups of framework (ATen) launches in the graphed step (round 6;
// VERDICT r5 item 6).  Inside the step's graph every node costs ~4.5-5 us whatever it does, so each entry below is worth
// the nodes it removes:
//   pdae_partials_sum_t    out[c][k] = sum_p part[p][k][c] in a fixed order: the first conv's weight gradient from its per-block
//                          partials (was at::sum + a transposing copy: 16.5 + 4.6 us)
//   pdae_multi_copy        n (src -> dst, count) float copies in ONE launch: the gather of the gradients that autograd left
//                          outside the flat buffer (was torch._foreach_copy_: two multi_tensor_apply launches)
//                          [a 2-D source with a row stride is allowed per entry: the (128, 3) view of a (128, 4) tile]
//   pdae_assemble_tokens   decoder input rows [visible tokens | M copies of the mask token] per sample (was expand + cat:
//                          PointCAE_transformer.py:700-703); its backward splits the gradient into the two contiguous parts
//   pdae_embed_split_conv3_weight / pdae_embed_masked_prep / pdae_embed_dw3_assemble
//                          the patch embedder's small element-wise steps around conv3's split weight (patch_embed.py), each
//                          one launch where ATen issued two to four
#include "common.h"

namespace pdae {

// 16 columns x 16 partial-lanes per block: lane l sums partials l, l + 16, ... in ascending order, the 16 lane sums are added
// in ascending order (a fixed order: the result does not depend on the launch)
__global__ __launch_bounds__(256) void partials_sum_t_kernel(int P, int K, int C, const float* __restrict__ part,
                                                             float* __restrict__ out) {
  __shared__ float red[16][17];
  const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + col;                   // i = k * C + c
  const int n = K * C;
  float t = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int p = lane; p < P; p += 16) t += part[(size_t)p * n + i];
  }
  red[lane][col] = t;
  __syncthreads();
  if (lane == 0 && i < n) {
    float r = red[0][col];
#pragma unroll
    for (int l = 1; l < 16; ++l) r += red[l][col];
    out[(size_t)(i % C) * K + i / C] = r;
  }
}

constexpr int MC_MAX = 128;                                // tensors per launch (the by-value argument: 3.6 KB of the 4 KB kernarg)
constexpr int MC_UNIT = 2048;                              // elements per block
struct MultiCopyArgs {
  int n;
  int unit0[MC_MAX];                                       // first unit of tensor i (prefix sums of the unit counts)
  const float* src[MC_MAX];
  float* dst[MC_MAX];
  int count[MC_MAX];
  unsigned short cols[MC_MAX], src_ld[MC_MAX];             // cols == 0: contiguous; else src element i at (i / cols) * src_ld + i % cols
};

__global__ __launch_bounds__(256) void multi_copy_kernel(const MultiCopyArgs a) {
  int lo = 0, hi = a.n - 1;                                // the last tensor whose first unit is <= blockIdx.x (block-uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.unit0[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const int t = lo, u = blockIdx.x - a.unit0[t];
  const int beg = u * MC_UNIT, end = min(a.count[t], beg + MC_UNIT);
  const float* s = a.src[t];
  float* d = a.dst[t];
  const int cols = a.cols[t], ld = a.src_ld[t];
  if (!s) {                                                // no source: zero fill
    for (int i = beg + threadIdx.x; i < end; i += 256) d[i] = 0.f;
  } else if (cols) {
    for (int i = beg + threadIdx.x; i < end; i += 256) d[i] = s[(i / cols) * ld + i % cols];
  } else if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
    const int e4 = beg + ((end - beg) & ~3);
    for (int i = beg + threadIdx.x * 4; i < e4; i += 1024) *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(s + i);
    for (int i = e4 + threadIdx.x; i < end; i += 256) d[i] = s[i];
  } else {
    for (int i = beg + threadIdx.x; i < end; i += 256) d[i] = s[i];
  }
}

// out[(b, t)] = t < Tv ? vis[(b, t)] : token, one float4 per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(long long n4, int C4, int G, int Tv, const float4* __restrict__ vis,
                                                              const float4* __restrict__ token, float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long row = i / C4;
  const int q = (int)(i - row * C4), t = (int)(row % G);
  const long long b = row / G;
  out[i] = t < Tv ? vis[(b * Tv + t) * C4 + q] : token[q];
}

// the backward of assemble_tokens: dout (B, G, C) -> dvis (B, Tv, C) and dmask (B, G - Tv, C), both contiguous (the mask
// token's gradient is the column sum of dmask: pdae_colsum)
__global__ __launch_bounds__(256) void assemble_tokens_grad_kernel(long long n4, int C4, int G, int Tv, const float4* __restrict__ dout,
                                                                   float4* __restrict__ dvis, float4* __restrict__ dmask) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long row = i / C4;
  const int q = (int)(i - row * C4), t = (int)(row % G);
  const long long b = row / G;
  const float4 v = dout[i];
  if (t < Tv) dvis[(b * Tv + t) * C4 + q] = v;
  else dmask[(b * (G - Tv) + (t - Tv)) * C4 + q] = v;
}

// conv3's weight w (N, 2 K2) [global half | local half] -> wg (N, K2), wl (N, K2) and wlt = wl^T (K2, N)
__global__ __launch_bounds__(256) void split_conv3_weight_kernel(int N, int K2, const float* __restrict__ w, float* __restrict__ wg,
                                                                 float* __restrict__ wl, float* __restrict__ wlt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * 2 * K2) return;
  const int n = i / (2 * K2), k = i - n * 2 * K2;
  const float v = w[i];
  if (k < K2) {
    wg[n * K2 + k] = v;
  } else {
    wl[n * K2 + k - K2] = v;
    if (wlt) wlt[(size_t)(k - K2) * N + n] = v;
  }
}

// the element-wise operands of the masked-groups algebra (patch_embed.py _masked_by_algebra):
//   xe[g][c] = u[c] + gb[masked[g]][c] * v[c]   (Gm, C3)        wv[c][k] = wl[c][k] * v[c]   (C3, C2)
__global__ __launch_bounds__(256) void masked_prep_kernel(int Gm, int C3, int C2, const float* __restrict__ uv, const float* __restrict__ gb,
                                                          const int* __restrict__ masked, const float* __restrict__ wl,
                                                          float4* __restrict__ xe, float4* __restrict__ wv) {
  const int q3 = C3 / 4, q2 = C2 / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < Gm * q3) {
    const int g = i / q3, c = (i - g * q3) * 4;
    const float4 b = *reinterpret_cast<const float4*>(gb + (size_t)masked[g] * C3 + c);
    const float4 u = *reinterpret_cast<const float4*>(uv + c), v = *reinterpret_cast<const float4*>(uv + C3 + c);
    xe[i] = make_float4(u.x + b.x * v.x, u.y + b.y * v.y, u.z + b.z * v.z, u.w + b.w * v.w);
  } else if (i < Gm * q3 + C3 * q2) {
    const int j = i - Gm * q3, c = j / q2;
    const float4 w = reinterpret_cast<const float4*>(wl)[j];
    const float v = uv[C3 + c];
    wv[j] = make_float4(w.x * v, w.y * v, w.z * v, w.w * v);
  }
}

// conv3's weight gradient in the parameter's layout: dw3 (C3, 2 C2) = [dwg | dwl + v (.) wgram + xterm]  (v == null: [dwg | dwl])
__global__ __launch_bounds__(256) void dw3_assemble_kernel(int C3, int C2, const float4* __restrict__ dwg, const float4* __restrict__ dwl,
                                                           const float* __restrict__ v, const float4* __restrict__ wgram,
                                                           const float4* __restrict__ xterm, float4* __restrict__ dw3) {
  const int q2 = C2 / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;            // one float4 of dw3
  if (i >= C3 * 2 * q2) return;
  const int c = i / (2 * q2), k = i - c * 2 * q2;
  if (k < q2) {
    dw3[i] = dwg[c * q2 + k];
  } else {
    const int j = c * q2 + k - q2;
    float4 d = dwl[j];
    if (v) {
      const float s = v[c];
      const float4 g = wgram[j], x = xterm[j];
      d = make_float4((d.x + s * g.x) + x.x, (d.y + s * g.y) + x.y, (d.z + s * g.z) + x.z, (d.w + s * g.w) + x.w);
    }
    dw3[i] = d;
  }
}

// out (R2, C2) = in (R, C; row stride ld) in its top-left corner, zeros elsewhere (F.pad of a small weight / a column slice of one /
// a narrow activation: was a fill + a copy)
__global__ __launch_bounds__(256) void pad2d_kernel(long long n, int R, int C, int ld, int C2, const float* __restrict__ in,
                                                    float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long r = i / C2;
  const int c = (int)(i - r * C2);
  out[i] = (r < R && c < C) ? in[r * ld + c] : 0.f;
}

// out[b][c] = max_t x[b][t][c] + mean_t x[b][t][c] (the published variant's global feature, models/PointCAE_transformer.py:1024:
// x.max(dim=1)[0] + x.mean(1)); arg[b][c] = the first t of the maximum.  Backward: dx[b][t][c] = g[b][c] (1 / T + [t == arg]).
__global__ __launch_bounds__(256) void max_plus_mean_kernel(int n, int T, int C, const float* __restrict__ x, float* __restrict__ out,
                                                            unsigned char* __restrict__ arg) {
  const int i = blockIdx.x * 256 + threadIdx.x;            // i = b * C + c
  if (i >= n) return;
  const int b = i / C, c = i - b * C;
  const float* p = x + (size_t)b * T * C + c;
  float m = p[0], s = p[0];
  int a = 0;
  for (int t = 1; t < T; ++t) {
    const float v = p[(size_t)t * C];
    s += v;
    if (v > m) m = v, a = t;
  }
  out[i] = m + s / (float)T;
  arg[i] = (unsigned char)a;
}

__global__ __launch_bounds__(256) void max_plus_mean_grad_kernel(long long n, int T, int C, const float* __restrict__ g,
                                                                 const unsigned char* __restrict__ arg, float* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // i = (b * T + t) * C + c
  if (i >= n) return;
  const int c = (int)(i % C);
  const long long bt = i / C;
  const int t = (int)(bt % T);
  const long long b = bt / T;
  const float gv = g[b * C + c];
  dx[i] = gv / (float)T + (arg[b * C + c] == t ? gv : 0.f);
}

// out (R, C) = [p0 | p1 | ...]: up to four row-major pieces side by side, piece q = cols[q] columns read through row stride ld[q]
// (the gradient of a weight that was used as column blocks: was a zero fill + a copy per block and the adds of autograd)
struct HcatArgs {
  int n, R, C;
  const float* src[4];
  int cols[4], ld[4], at[4];
};
__global__ __launch_bounds__(256) void hcat_kernel(const HcatArgs a, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)a.R * a.C) return;
  const long long r = i / a.C;
  const int c = (int)(i - r * a.C);
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (q < a.n && a.src[q] && c >= a.at[q] && c < a.at[q] + a.cols[q]) v = a.src[q][r * a.ld[q] + (c - a.at[q])];   // (no source: zeros)
  out[i] = v;
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_hcat(int n, int R, const float* const* src, const int* cols, const int* ld, float* out, pdae_stream_t stream) {
  if (n <= 0 || n > 4 || R < 0 || !src || !cols || !ld) return bad_arg("hcat: 1..4 pieces");
  HcatArgs a = {};
  a.n = n, a.R = R;
  int at = 0;
  for (int q = 0; q < n; ++q) {
    if (cols[q] <= 0 || (src[q] && ld[q] < cols[q])) return bad_arg("hcat: bad piece");
    a.src[q] = src[q], a.cols[q] = cols[q], a.ld[q] = ld[q], a.at[q] = at;
    at += cols[q];
  }
  a.C = at;
  if (R == 0) return PDAE_OK;
  if (!out) return bad_arg("hcat: null pointer");
  const long long n_el = (long long)R * at;
  if ((n_el + 255) / 256 > 0x7fffffffLL) return unsupported("hcat: too many elements");
  hipLaunchKernelGGL(hcat_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, as_stream(stream), a, out);
  return check_launch("hcat");
}

extern "C" int pdae_max_plus_mean(int B, int T, int C, const float* x, float* out, unsigned char* arg, pdae_stream_t stream) {
  if (B < 0 || T <= 0 || T > 255 || C <= 0 || (long long)B * C >= (1LL << 31)) return bad_arg("max_plus_mean: 1 <= T <= 255");
  if (B == 0) return PDAE_OK;
  if (!x || !out || !arg) return bad_arg("max_plus_mean: null pointer");
  hipLaunchKernelGGL(max_plus_mean_kernel, dim3((B * C + 255) / 256), dim3(256), 0, as_stream(stream), B * C, T, C, x, out, arg);
  return check_launch("max_plus_mean");
}

extern "C" int pdae_max_plus_mean_grad(int B, int T, int C, const float* g, const unsigned char* arg, float* dx, pdae_stream_t stream) {
  if (B < 0 || T <= 0 || T > 255 || C <= 0) return bad_arg("max_plus_mean_grad: 1 <= T <= 255");
  if (B == 0) return PDAE_OK;
  if (!g || !arg || !dx) return bad_arg("max_plus_mean_grad: null pointer");
  const long long n = (long long)B * T * C;
  if ((n + 255) / 256 > 0x7fffffffLL) return unsupported("max_plus_mean_grad: too many elements");
  hipLaunchKernelGGL(max_plus_mean_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n, T, C, g, arg, dx);
  return check_launch("max_plus_mean_grad");
}

extern "C" int pdae_pad2d(int R, int C, int ld, int R2, int C2, const float* in, float* out, pdae_stream_t stream) {
  if (R < 0 || C < 0 || ld < C || R2 < R || C2 < C || C2 <= 0) return bad_arg("pad2d: 0 <= R <= R2, 0 <= C <= C2, C <= ld, C2 > 0");
  if (R2 == 0) return PDAE_OK;
  if (!out || (R > 0 && C > 0 && !in)) return bad_arg("pad2d: null pointer");
  const long long n = (long long)R2 * C2;
  if ((n + 255) / 256 > 0x7fffffffLL) return unsupported("pad2d: too many elements");
  hipLaunchKernelGGL(pad2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n, R, C, ld, C2, in, out);
  return check_launch("pad2d");
}

extern "C" int pdae_partials_sum_t(int P, int K, int C, const float* part, float* out, pdae_stream_t stream) {
  if (P < 0 || K <= 0 || C <= 0) return bad_arg("partials_sum_t: bad size");
  if ((P > 0 && !part) || !out) return bad_arg("partials_sum_t: null pointer");
  hipLaunchKernelGGL(partials_sum_t_kernel, dim3((K * C + 15) / 16), dim3(256), 0, as_stream(stream), P, K, C, part, out);
  return check_launch("partials_sum_t");
}

extern "C" int pdae_multi_copy(int n, const float* const* src, float* const* dst, const long long* counts, const int* cols,
                               const int* src_ld, pdae_stream_t stream) {
  if (n < 0) return bad_arg("multi_copy: n < 0");
  if (n == 0) return PDAE_OK;
  if (!src || !dst || !counts || (!cols != !src_ld)) return bad_arg("multi_copy: null pointer");
  for (int i0 = 0; i0 < n; i0 += MC_MAX) {
    MultiCopyArgs a = {};
    a.n = n - i0 < MC_MAX ? n - i0 : MC_MAX;
    long long units = 0;
    for (int i = 0; i < a.n; ++i) {
      const long long c = counts[i0 + i];
      if (c < 0 || c >= (1LL << 31) || (c > 0 && !dst[i0 + i])) return bad_arg("multi_copy: bad entry");
      const int w = cols ? cols[i0 + i] : 0, ld = cols ? src_ld[i0 + i] : 0;
      if (w < 0 || w > 65535 || ld < w || ld > 65535) return bad_arg("multi_copy: 0 <= cols <= src_ld <= 65535");
      a.unit0[i] = (int)units, a.src[i] = src[i0 + i], a.dst[i] = dst[i0 + i], a.count[i] = (int)c;
      a.cols[i] = (unsigned short)(w == ld ? 0 : w), a.src_ld[i] = (unsigned short)ld;
      units += (c + MC_UNIT - 1) / MC_UNIT;
    }
    if (units >= (1LL << 31)) return bad_arg("multi_copy: too many elements for one launch");
    if (units > 0) hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)units), dim3(256), 0, as_stream(stream), a);
  }
  return check_launch("multi_copy");
}

extern "C" int pdae_assemble_tokens(int B, int G, int Tv, int C, const float* vis, const float* token, float* out,
                                    pdae_stream_t stream) {
  if (B < 0 || G <= 0 || Tv < 0 || Tv > G || C <= 0 || C % 4 != 0) return bad_arg("assemble_tokens: C % 4 == 0, 0 <= Tv <= G");
  if (B == 0) return PDAE_OK;
  if ((Tv > 0 && !vis) || !token || !out) return bad_arg("assemble_tokens: null pointer");
  const long long n4 = (long long)B * G * (C / 4);
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4, C / 4, G, Tv,
                     reinterpret_cast<const float4*>(vis), reinterpret_cast<const float4*>(token), reinterpret_cast<float4*>(out));
  return check_launch("assemble_tokens");
}

extern "C" int pdae_assemble_tokens_grad(int B, int G, int Tv, int C, const float* dout, float* dvis, float* dmask,
                                         pdae_stream_t stream) {
  if (B < 0 || G <= 0 || Tv < 0 || Tv > G || C <= 0 || C % 4 != 0) return bad_arg("assemble_tokens_grad: C % 4 == 0, 0 <= Tv <= G");
  if (B == 0) return PDAE_OK;
  if (!dout || (Tv > 0 && !dvis) || (Tv < G && !dmask)) return bad_arg("assemble_tokens_grad: null pointer");
  const long long n4 = (long long)B * G * (C / 4);
  hipLaunchKernelGGL(assemble_tokens_grad_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4, C / 4, G,
                     Tv, reinterpret_cast<const float4*>(dout), reinterpret_cast<float4*>(dvis), reinterpret_cast<float4*>(dmask));
  return check_launch("assemble_tokens_grad");
}

extern "C" int pdae_embed_split_conv3_weight(int N, int K2, const float* w, float* wg, float* wl, float* wlt, pdae_stream_t stream) {
  if (N <= 0 || K2 <= 0 || (long long)N * 2 * K2 >= (1LL << 31)) return bad_arg("embed_split_conv3_weight: bad size");
  if (!w || !wg || !wl) return bad_arg("embed_split_conv3_weight: null pointer");
  hipLaunchKernelGGL(split_conv3_weight_kernel, dim3((N * 2 * K2 + 255) / 256), dim3(256), 0, as_stream(stream), N, K2, w, wg, wl, wlt);
  return check_launch("embed_split_conv3_weight");
}

extern "C" int pdae_embed_masked_prep(int Gm, int C3, int C2, const float* uv, const float* gb, const int* masked, const float* wl,
                                      float* xe, float* wv, pdae_stream_t stream) {
  if (Gm < 0 || C3 <= 0 || C2 <= 0 || C3 % 4 != 0 || C2 % 4 != 0) return bad_arg("embed_masked_prep: widths must be multiples of 4");
  if ((long long)Gm * C3 + (long long)C3 * C2 >= (1LL << 31)) return bad_arg("embed_masked_prep: too large");
  if (!uv || !wl || !wv || (Gm > 0 && (!gb || !masked || !xe))) return bad_arg("embed_masked_prep: null pointer");
  const int n = Gm * (C3 / 4) + C3 * (C2 / 4);
  hipLaunchKernelGGL(masked_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), Gm, C3, C2, uv, gb, masked, wl,
                     reinterpret_cast<float4*>(xe), reinterpret_cast<float4*>(wv));
  return check_launch("embed_masked_prep");
}

extern "C" int pdae_embed_dw3_assemble(int C3, int C2, const float* dwg, const float* dwl, const float* v, const float* wgram,
                                       const float* xterm, float* dw3, pdae_stream_t stream) {
  if (C3 <= 0 || C2 <= 0 || C2 % 4 != 0 || (long long)C3 * 2 * C2 >= (1LL << 31)) return bad_arg("embed_dw3_assemble: C2 % 4 == 0");
  if (!dwg || !dwl || !dw3 || (v && (!wgram || !xterm))) return bad_arg("embed_dw3_assemble: null pointer");
  const int n = C3 * 2 * (C2 / 4);
  hipLaunchKernelGGL(dw3_assemble_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), C3, C2,
                     reinterpret_cast<const float4*>(dwg), reinterpret_cast<const float4*>(dwl), v,
                     reinterpret_cast<const float4*>(wgram), reinterpret_cast<const float4*>(xterm), reinterpret_cast<float4*>(dw3));
  return check_launch("embed_dw3_assemble");
}
