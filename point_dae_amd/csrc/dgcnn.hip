// dgcnn.hip -- the DGCNN encoder's EdgeConv layers and its global pool for gfx950
// (models/dgcnn_util.py:7-34 knn + get_graph_feature, :87-136 encoder; models/PointCAE_DGCNN.py:146-231 the
// auto-encoder the published non-Transformer checkpoints were trained with).
//
// The reference materialises cat(x_j - x_i, x_i) as a (B, 2C, N, 20) tensor per layer, convolves it, and runs
// BatchNorm2d, LeakyReLU and the max over the 20 neighbours as separate passes over (B, C', N, 20).  Here, with
// activations as rows (points) x channels:
//   * conv([x_j - x_i, x_i]) = W1 x_j + (W2 - W1) x_i = p[j] + q[i]: ONE row GEMM per layer on the stacked weight
//     gives pq = [p | q] per POINT (20x fewer FLOPs, no edge tensor);
//   * the graph: Gram rows from the batched row GEMM -> `gram_topk` (one wave per row, the exact 64-bit-key selection
//     of knn.hip on the reference's own expression -xx_i + 2 g_ij - xx_j);
//   * BatchNorm(train) + LeakyReLU + max over the neighbours needs the edge values only for (a) the channel sums and
//     (b) the winner -- and y = lrelu(e * scale + shift) is monotone in e, increasing or decreasing with the SIGN OF
//     GAMMA, which is known before the statistics are: `edge_gather_stats` makes ONE pass over the 20 gathered rows of
//     every point and keeps the channel sums (fp64), the winning edge value (max of e if gamma > 0, min if < 0, the
//     first edge if = 0: torch.max's first-occurrence rule), the winner's id and sum_j p[j]; `bn_lrelu_rows` then is
//     a per-POINT pass.  No (rows x 20 x C') tensor exists in either direction;
//   * backward: d e[r,j,c] = scale_c (dy[r,j,c] - c1 - xhat[r,j,c] c2) is dense over the edges through the xhat
//     term, but affine in p[j] + q[r]: dq[r] needs sum_j p[j] (kept by the forward), dp[s] needs the sums of q[r] and
//     of the winners' g[r] over the edges ARRIVING at s -- a gather over the reverse graph (`knn_reverse`: an adjacency
//     bitmap per cloud in LDS, walked per target in ascending source order: deterministic, no atomics);
//   * conv5 -> BatchNorm1d -> LeakyReLU -> max over the cloud's points: `cloud_pool_stats` (same monotone trick, one
//     pass over the GEMM output), `cloud_pool_backward`.
#include "common.h"
#include "wave_select.h"

namespace pdae {

constexpr float kSlope = 0.2f;   // nn.LeakyReLU(negative_slope=0.2), dgcnn_util.py:99-115

// ---- |x_r|^2 per row ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_sqnorm_kernel(int R, int C, const float* __restrict__ x,
                                                          float* __restrict__ xx) {
  const int sub = threadIdx.x & 15;
  const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  float s = 0.f;
  if (r < R)
    for (int c = sub * 4; c < C; c += 64) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (r < R && sub == 0) xx[r] = s;
}

// ---- top-k of the reference's pairwise_distance rows ---------------------------------------------------------------
__device__ __forceinline__ unsigned ordered_bits(float f) {   // unsigned order == float order (finite values)
  const unsigned u = __float_as_uint(f);
  return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
// -pairwise_distance[i][j] of dgcnn_util.knn (:8-10): xx is (B, 1, N), so `-xx - inner` broadcasts xx over the COLUMNS:
// pd[i][j] = ((-xx_j) - inner_ij) - xx_i with inner = -2 g (an exact scaling), each subtraction rounded once
__device__ __forceinline__ float neg_pd(float g, float xi, float xj) { return xi - (2.0f * g - xj); }

// One wave per row: two passes over the (L2-resident) row.  Measured alternatives, both slower (76-81 against 66-69 us
// at 32 x 1024^2): the row held in registers (one read, 16 more live registers), and four rows per wave (the
// selection is a dependent chain per row; more resident waves hide it better).
// XYZ: the first EdgeConv's graph is built on the points themselves (C = 3, stored as 4 columns): its Gram entries are
// three products each, computed here (g = fma(z z', fma(y y', x x'))) from the cloud's rows -- staged ONCE per block into
// LDS (16 B per point; a block then serves 16 rows, four per wave: read per candidate from global the 16-byte rows cost
// 1 KB of L1 traffic per wave load, 103 us per launch) -- instead of being written as a (b, n, n) matrix by a K = 4 GEMM
// and read back (42 us + 134 MB each way per step).
template <bool XYZ>
__global__ __launch_bounds__(256) void gram_topk_kernel(int n, int k, const float* __restrict__ src_all,
                                                        const float* __restrict__ xx_all, int* __restrict__ idx_all,
                                                        float* __restrict__ pd_out) {
  __shared__ unsigned long long stage_all[4][128];
  extern __shared__ __attribute__((aligned(16))) char lds_xyz[];      // XYZ: the cloud's n rows | their n squared norms
  constexpr int RPW = XYZ ? 4 : 1;                                     // rows per wave
  const int lane = lane_id(), wave = threadIdx.x / kWave;
  unsigned long long* stage = stage_all[wave];
  const int bi = blockIdx.y;
  const float* xx = xx_all + (size_t)bi * n;
  const float4* pts = reinterpret_cast<const float4*>(lds_xyz);
  if (XYZ) {
    float4* pw = reinterpret_cast<float4*>(lds_xyz);
    float* xw = reinterpret_cast<float*>(lds_xyz + (size_t)n * 16);
    const float4* gsrc = reinterpret_cast<const float4*>(src_all) + (size_t)bi * n;
    for (int p = threadIdx.x; p < n; p += 256) pw[p] = gsrc[p], xw[p] = xx[p];
    __syncthreads();
    xx = xw;
  }
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = (blockIdx.x * RPW + rr) * 4 + wave;
    if (i >= n) return;
    const float* g = XYZ ? nullptr : src_all + ((size_t)bi * n + i) * n;
    const float xi = xx[i];
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (XYZ) pi = pts[i];
    auto dist = [&](int p) {
      if (!XYZ) return neg_pd(g[p], xi, xx[p]);
      const float4 q = pts[p];
      return neg_pd(fmaf(pi.z, q.z, fmaf(pi.y, q.y, pi.x * q.x)), xi, xx[p]);
    };
    float lane_min = __builtin_huge_valf();
    for (int p = lane; p < n; p += kWave) lane_min = fminf(lane_min, dist(p));
    const float t = wave_kth_smallest(lane_min, k);
    KnnSelect st;
    st.best = kKeyMax;
    st.bound = ((unsigned long long)ordered_bits(t) << 32) | 0xffffffffull;
    st.staged = 0;
    st.have_best = false;
    for (int p0 = 0; p0 < n; p0 += kWave) {
      const int p = p0 + lane;
      const bool in = p < n;
      const float d = in ? dist(p) : 0.f;
      if (XYZ && pd_out && in) pd_out[((size_t)bi * n + i) * n + p] = d;     // (tests: the values the selection saw)
      knn_offer(st, stage, ((unsigned long long)ordered_bits(d) << 32) | (unsigned)p, in, k);
    }
    if (st.staged > 0 || !st.have_best) knn_flush(st, stage, st.staged, k);
    if (lane < k) idx_all[((size_t)bi * n + i) * k + lane] = (int)(st.best & 0xffffffffull);
  }
}

// ---- reverse graph: for every point the points that list it as a neighbour, ascending ----------------------------
// One block per cloud.  LDS: cnt[n] | cursor[n] | bitmap[n][W + 1] (W words = `chunk` source rows per pass).
__global__ __launch_bounds__(1024) void knn_reverse_kernel(int n, int k, int chunk, const int* __restrict__ idx_all,
                                                           int* __restrict__ start_all, int* __restrict__ src_all) {
  extern __shared__ int lds_rev[];
  int* cnt = lds_rev;
  int* cursor = cnt + n;
  unsigned* bitmap = reinterpret_cast<unsigned*>(cursor + n);
  __shared__ int wave_tot[16];
  const int W = chunk / 32, WS = W + 1 + (W & 1);
  const int bi = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int* idx = idx_all + (size_t)bi * n * k;
  int* start = start_all + (size_t)bi * (n + 1);
  int* src = src_all + (size_t)bi * n * k;
  for (int s = tid; s < n; s += nt) cnt[s] = 0;
  __syncthreads();
  for (int e = tid; e < n * k; e += nt) atomicAdd(&cnt[idx[e]], 1);
  __syncthreads();
  // exclusive scan of cnt: a thread owns `per` consecutive targets
  const int per = (n + nt - 1) / nt;
  int mine = 0;
  for (int q = 0; q < per; ++q) {
    const int s = tid * per + q;
    if (s < n) mine += cnt[s];
  }
  int incl = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int v = __shfl_up(incl, o, kWave);
    if (lane_id() >= o) incl += v;
  }
  if (lane_id() == kWave - 1) wave_tot[tid / kWave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < tid / kWave; ++w) base += wave_tot[w];
  int run = base + incl - mine;
  for (int q = 0; q < per; ++q) {
    const int s = tid * per + q;
    if (s < n) {
      const int c = cnt[s];
      cursor[s] = run;
      start[s] = run;
      run += c;
    }
  }
  if (tid == 0) start[n] = n * k;
  for (int r0 = 0; r0 < n; r0 += chunk) {
    __syncthreads();
    for (int q = tid; q < n * WS; q += nt) bitmap[q] = 0u;
    __syncthreads();
    const int r1 = min(n, r0 + chunk);
    for (int e = r0 * k + tid; e < r1 * k; e += nt) {
      const int rl = e / k - r0;
      atomicOr(&bitmap[idx[e] * WS + (rl >> 5)], 1u << (rl & 31));
    }
    __syncthreads();
    for (int s = tid; s < n; s += nt) {
      int cur = cursor[s];
      for (int w = 0; w < W; ++w) {
        unsigned word = bitmap[s * WS + w];
        while (word) {
          const int bit = __builtin_ctz(word);
          word &= word - 1;
          src[cur++] = r0 + w * 32 + bit;
        }
      }
      cursor[s] = cur;
    }
  }
}

// ---- EdgeConv forward: one pass over the gathered neighbour rows -------------------------------------------------
// pq [R][2 co]: p = W1 x (columns 0..co), q = (W2 - W1) x (columns co..2co).  A thread owns 4 channels of a row.
// esel = winning p[j] + q, sel = the winner's point id within the cloud, psum = sum_j p[j];
// part [gridDim.x][2][co] doubles: sum e, sum e^2 over this block's rows x neighbours.
__global__ __launch_bounds__(256) void edge_gather_stats_kernel(int R, int n, int k, int co,
                                                                const float* __restrict__ pq,
                                                                const int* __restrict__ idx,
                                                                const float* __restrict__ gamma,
                                                                float* __restrict__ esel,
                                                                unsigned short* __restrict__ sel,
                                                                float* __restrict__ psum, double* __restrict__ part) {
  extern __shared__ double lds_red[];                       // [256][8]
  const int tpr = co / 4, rpb = 256 / tpr;
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr, c4 = tc * 4;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + c4);
  const int mode[4] = {gm.x > 0.f ? 1 : (gm.x < 0.f ? -1 : 0), gm.y > 0.f ? 1 : (gm.y < 0.f ? -1 : 0),
                       gm.z > 0.f ? 1 : (gm.z < 0.f ? -1 : 0), gm.w > 0.f ? 1 : (gm.w < 0.f ? -1 : 0)};
  double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
  const int ld = 2 * co;
  // XCD-aware row order: blocks b = x (mod 8) run on XCD x and take the clouds x, x + 8, ... -- a cloud's p rows (at most
  // 1 MB) then stay in that XCD's 4 MB L2 for all 20 gathers of each of them instead of being fetched through the fabric
  // by all eight (grid a multiple of 8)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nclouds = R / n, mine = (nclouds - xcd + 7) / 8;
  for (long long i = (long long)slot * rpb + rl; i < (long long)mine * n; i += (long long)nslots * rpb) {
    const long long base = (long long)(xcd + 8 * (int)(i / n)) * n;
    const long long r = base + i % n;
    const float4 qv = *reinterpret_cast<const float4*>(pq + r * ld + co + c4);
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    const int* nb = idx + r * k;
    float best[4], ps[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, rss[4] = {0.f, 0.f, 0.f, 0.f};
    int who[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) best[u] = mode[u] > 0 ? -__builtin_huge_valf() : __builtin_huge_valf();
#pragma unroll 4
    for (int j = 0; j < k; ++j) {
      const int id = nb[j];
      const float4 pv = *reinterpret_cast<const float4*>(pq + (base + id) * ld + c4);
      const float p[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float e = p[u] + q[u];
        ps[u] += p[u];
        rs[u] += e;
        rss[u] += e * e;
        const bool take = mode[u] > 0 ? e > best[u] : (mode[u] < 0 ? e < best[u] : j == 0);
        best[u] = take ? e : best[u];
        who[u] = take ? id : who[u];
      }
    }
    *reinterpret_cast<float4*>(esel + r * co + c4) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<float4*>(psum + r * co + c4) = make_float4(ps[0], ps[1], ps[2], ps[3]);
    *reinterpret_cast<ushort4*>(sel + r * co + c4) =
        make_ushort4((unsigned short)who[0], (unsigned short)who[1], (unsigned short)who[2], (unsigned short)who[3]);
#pragma unroll
    for (int u = 0; u < 4; ++u) s1[u] += (double)rs[u], s2[u] += (double)rss[u];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) lds_red[threadIdx.x * 8 + u] = s1[u], lds_red[threadIdx.x * 8 + 4 + u] = s2[u];
  __syncthreads();
  if (rl == 0) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = 0.;
    for (int w = 0; w < rpb; ++w)
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += lds_red[(w * tpr + tc) * 8 + u];
    double* o = part + (size_t)blockIdx.x * 2 * co;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[c4 + u] = t[u], o[co + c4 + u] = t[4 + u];
  }
}

// out[c] = sum_p part[p][c]: 16 interleaved chains per column (p = i, i + 16, ...), added in chain order -- a fixed
// order whatever the launch; optionally the float copies fa = out[0..half), fb = out[half..)
__global__ __launch_bounds__(1024) void part_reduce_f64_kernel(int P, int width, const double* __restrict__ part,
                                                               double* __restrict__ out, float* __restrict__ fa,
                                                               float* __restrict__ fb) {
  __shared__ double red[16][kWave];
  const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
  const int c = blockIdx.x * kWave + cx;
  double t = 0.;
  if (c < width) {
#pragma unroll 8
    for (int p = py; p < P; p += 16) t += part[(size_t)p * width + c];
  }
  red[py][cx] = t;
  __syncthreads();
  if (py == 0 && c < width) {
    double a = 0.;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += red[i][cx];
    out[c] = a;
    const int half = width / 2;
    if (fa && c < half) fa[c] = (float)a;
    if (fb && c >= half) fb[c - half] = (float)a;
  }
}

// y = lrelu(e * scale + shift) on rows; e [R][C]; out (row stride C) and, optionally, a second copy into a wider
// row-major tensor (the concatenated features conv5 reads).
__global__ __launch_bounds__(256) void bn_lrelu_rows_kernel(long long R, int C, const float* __restrict__ e,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            float* __restrict__ out2, int ld2) {
  const int c4n = C / 4;
  const long long total = R * c4n;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / c4n;
    const int c4 = (int)(t - r * c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(e + r * C + c4);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4), sh = *reinterpret_cast<const float4*>(shift + c4);
    float y[4] = {v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) y[u] = y[u] > 0.f ? y[u] : y[u] * kSlope;
    const float4 o = make_float4(y[0], y[1], y[2], y[3]);
    *reinterpret_cast<float4*>(out + r * C + c4) = o;
    if (out2) *reinterpret_cast<float4*>(out2 + r * ld2 + c4) = o;
  }
}

// Backward through LeakyReLU at the winners + the BatchNorm sums: g = (d1 + d2) * lrelu'(y), y = e scale + shift;
// part [gridDim.x][2][C]: sum g, sum g xhat with xhat = (e - mean) invstd.
__global__ __launch_bounds__(256) void bn_lrelu_backward_reduce_kernel(long long R, int C, const float* __restrict__ d1,
                                                                       const float* __restrict__ d2, int ld2,
                                                                       const float* __restrict__ e,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ shift,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd,
                                                                       float* __restrict__ g, double* __restrict__ part) {
  extern __shared__ double lds_red[];
  const int tpr = C / 4 < 256 ? C / 4 : 256, rpb = 256 / tpr;
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr;
  double* o = part + (size_t)blockIdx.x * 2 * C;
  for (int c4 = tc * 4; c4 < C; c4 += tpr * 4) {              // (C > 1024: several channel passes)
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4), sh = *reinterpret_cast<const float4*>(shift + c4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4), is = *reinterpret_cast<const float4*>(invstd + c4);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
    double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
    for (long long r = (long long)blockIdx.x * rpb + rl; r < R; r += (long long)gridDim.x * rpb) {
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d1) d = *reinterpret_cast<const float4*>(d1 + r * C + c4);
      if (d2) {
        const float4 w = *reinterpret_cast<const float4*>(d2 + r * ld2 + c4);
        d.x += w.x, d.y += w.y, d.z += w.z, d.w += w.w;
      }
      const float4 ev = *reinterpret_cast<const float4*>(e + r * C + c4);
      const float dv[4] = {d.x, d.y, d.z, d.w}, evv[4] = {ev.x, ev.y, ev.z, ev.w};
      float gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float y = evv[u] * scv[u] + shv[u];
        gv[u] = y > 0.f ? dv[u] : dv[u] * kSlope;
        const float xh = (evv[u] - muv[u]) * isv[u];
        s1[u] += (double)gv[u];
        s2[u] += (double)(gv[u] * xh);
      }
      *reinterpret_cast<float4*>(g + r * C + c4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) lds_red[threadIdx.x * 8 + u] = s1[u], lds_red[threadIdx.x * 8 + 4 + u] = s2[u];
    __syncthreads();
    if (rl == 0) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = 0.;
      for (int w = 0; w < rpb; ++w)
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] += lds_red[(w * tpr + tc) * 8 + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[c4 + u] = t[u], o[C + c4 + u] = t[4 + u];
    }
  }
}

// dpq [R][2 co] = [dp | dq] from g, the forward's records and the reverse graph (header note).
__global__ __launch_bounds__(256) void edge_backward_kernel(int R, int n, int k, int co, const float* __restrict__ g,
                                                            const float* __restrict__ pq,
                                                            const unsigned short* __restrict__ sel,
                                                            const float* __restrict__ psum,
                                                            const int* __restrict__ rev_start,
                                                            const int* __restrict__ rev_src,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const double* __restrict__ sums, float* __restrict__ dpq) {
  const int tpr = co / 4, rpb = 256 / tpr;
  const int tc = threadIdx.x % tpr, rl = threadIdx.x / tpr, c4 = tc * 4, ld = 2 * co;
  const double edges = (double)R * k;
  float sc[4], mu[4], c1[4], c2i[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    sc[u] = scale[c4 + u], mu[u] = mean[c4 + u];
    c1[u] = (float)(sums[c4 + u] / edges);
    c2i[u] = (float)(sums[co + c4 + u] / edges) * invstd[c4 + u];
  }
  const float kf = (float)k;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;      // (XCD-aware, as the forward pass)
  const int nclouds = R / n, mine = (nclouds - xcd + 7) / 8;
  for (long long i = (long long)slot * rpb + rl; i < (long long)mine * n; i += (long long)nslots * rpb) {
    const long long b = xcd + 8 * (int)(i / n), base = b * n;
    const int sl = (int)(i % n);
    const long long s = base + sl;
    const int* st = rev_start + b * (n + 1);
    const int e0 = st[sl], e1 = st[sl + 1];
    const int* src = rev_src + b * n * k;
    float G[4] = {0.f, 0.f, 0.f, 0.f}, Q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int t = e0; t < e1; ++t) {
      const long long r = base + src[t];
      const float4 gv = *reinterpret_cast<const float4*>(g + r * co + c4);
      const float4 qv = *reinterpret_cast<const float4*>(pq + r * ld + co + c4);
      const ushort4 w = *reinterpret_cast<const ushort4*>(sel + r * co + c4);
      G[0] += w.x == sl ? gv.x : 0.f, G[1] += w.y == sl ? gv.y : 0.f;
      G[2] += w.z == sl ? gv.z : 0.f, G[3] += w.w == sl ? gv.w : 0.f;
      Q[0] += qv.x, Q[1] += qv.y, Q[2] += qv.z, Q[3] += qv.w;
    }
    const float cntf = (float)(e1 - e0);
    const float4 pv = *reinterpret_cast<const float4*>(pq + s * ld + c4);
    const float4 qs = *reinterpret_cast<const float4*>(pq + s * ld + co + c4);
    const float4 gs = *reinterpret_cast<const float4*>(g + s * co + c4);
    const float4 ps = *reinterpret_cast<const float4*>(psum + s * co + c4);
    const float p[4] = {pv.x, pv.y, pv.z, pv.w}, q[4] = {qs.x, qs.y, qs.z, qs.w};
    const float gg[4] = {gs.x, gs.y, gs.z, gs.w}, pss[4] = {ps.x, ps.y, ps.z, ps.w};
    float dp[4], dq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dp[u] = sc[u] * (G[u] - cntf * c1[u] - c2i[u] * (cntf * (p[u] - mu[u]) + Q[u]));
      dq[u] = sc[u] * (gg[u] - kf * c1[u] - c2i[u] * (pss[u] + kf * (q[u] - mu[u])));
    }
    *reinterpret_cast<float4*>(dpq + s * ld + c4) = make_float4(dp[0], dp[1], dp[2], dp[3]);
    *reinterpret_cast<float4*>(dpq + s * ld + co + c4) = make_float4(dq[0], dq[1], dq[2], dq[3]);
  }
}

// ---- conv5's BatchNorm1d + LeakyReLU + max over a cloud's points -------------------------------------------------
// y [b n][C]; a block = one cloud x 256 channels x one of `rs` row ranges, wave w takes rows w, w+4, ... of its range.
// pv / pr [b][rs][C]: the range's winning value and its row within the cloud; part [b rs][2][C] doubles.
__global__ __launch_bounds__(256) void cloud_pool_stats_kernel(int n, int C, int rs, const float* __restrict__ y,
                                                               const float* __restrict__ gamma,
                                                               float* __restrict__ pv, int* __restrict__ pr,
                                                               double* __restrict__ part) {
  __shared__ double red[4][kWave][2];
  __shared__ float bestv[4][kWave];
  __shared__ int bestr[4][kWave];
  const int lane = lane_id(), wave = threadIdx.x / kWave;
  const int c4 = blockIdx.x * 256 + lane * 4, bi = blockIdx.y, ri = blockIdx.z;
  const int per = (n + rs - 1) / rs, r0 = ri * per, r1 = min(n, r0 + per);
  const bool live = c4 < C;
  float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) gm = *reinterpret_cast<const float4*>(gamma + c4);
  const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
  int mode[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) mode[u] = gmv[u] > 0.f ? 1 : (gmv[u] < 0.f ? -1 : 0);
  double s1[4] = {0., 0., 0., 0.}, s2[4] = {0., 0., 0., 0.};
  float best[4];
  int who[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
  for (int u = 0; u < 4; ++u) best[u] = mode[u] > 0 ? -__builtin_huge_valf() : __builtin_huge_valf();
  const float* yc = y + (size_t)bi * n * C;
  if (live)
    for (int r = r0 + wave; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(yc + (size_t)r * C + c4);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1[u] += (double)vv[u];
        s2[u] += (double)vv[u] * (double)vv[u];
        const bool take = mode[u] > 0 ? vv[u] > best[u] : (mode[u] < 0 ? vv[u] < best[u] : r == r0 + wave);
        best[u] = take ? vv[u] : best[u];
        who[u] = take ? r : who[u];
      }
    }
  double* o = part + ((size_t)bi * rs + ri) * 2 * C;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    __syncthreads();
    red[wave][lane][0] = s1[u], red[wave][lane][1] = s2[u];
    bestv[wave][lane] = best[u], bestr[wave][lane] = who[u];
    __syncthreads();
    if (wave == 0 && live) {
      double a = 0., b2 = 0.;
      float bv = bestv[0][lane];
      int br = bestr[0][lane];
      for (int w = 0; w < 4; ++w) {
        a += red[w][lane][0], b2 += red[w][lane][1];
        if (w > 0 && bestr[w][lane] != 0x7fffffff) {
          const float cv = bestv[w][lane];
          const int cr = bestr[w][lane];
          // the earliest row wins ties (torch.max's first occurrence); gamma = 0: the first row
          const bool better = mode[u] > 0 ? (cv > bv || (cv == bv && cr < br))
                                          : (mode[u] < 0 ? (cv < bv || (cv == bv && cr < br)) : cr < br);
          if (better || br == 0x7fffffff) bv = cv, br = cr;
        }
      }
      o[c4 + u] = a, o[C + c4 + u] = b2;
      pv[((size_t)bi * rs + ri) * C + c4 + u] = bv;
      pr[((size_t)bi * rs + ri) * C + c4 + u] = br;
    }
  }
}

// the winner of a cloud among its row ranges' winners (ranges ascend in rows: the first best keeps torch.max's rule)
__global__ void cloud_pool_merge_kernel(int total, int C, int rs, const float* __restrict__ pv, const int* __restrict__ pr,
                                        const float* __restrict__ gamma, float* __restrict__ ysel,
                                        int* __restrict__ arow) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int bi = t / C, c = t - bi * C;
  const float gmv = gamma[c];
  const int mode = gmv > 0.f ? 1 : (gmv < 0.f ? -1 : 0);
  float bv = 0.f;
  int br = 0x7fffffff;
  for (int ri = 0; ri < rs; ++ri) {
    const float cv = pv[((size_t)bi * rs + ri) * C + c];
    const int cr = pr[((size_t)bi * rs + ri) * C + c];
    if (cr == 0x7fffffff) continue;
    const bool better = br == 0x7fffffff || (mode > 0 ? cv > bv : (mode < 0 ? cv < bv : false));
    if (better) bv = cv, br = cr;
  }
  ysel[t] = bv;
  arow[t] = br;
}

// dy[r][c] = scale_c ((r == arow[b][c] ? g[b][c] : 0) - c1_c - xhat[r][c] c2_c), xhat = (y - mean) invstd
__global__ __launch_bounds__(256) void cloud_pool_backward_kernel(long long R, int n, int C, const float* __restrict__ y,
                                                                  const float* __restrict__ g,
                                                                  const int* __restrict__ arow,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const double* __restrict__ sums,
                                                                  float* __restrict__ dy) {
  const int c4n = C / 4;
  const long long total = R * c4n;
  // the launch makes the grid's thread count a multiple of C / 4: a thread keeps its four channels over the whole sweep,
  // so the per-channel terms (two fp64 divisions each) are formed once, not per element
  const int c4 = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % c4n) * 4;
  float c1[4], c2[4], mu[4], is[4], sc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    c1[u] = (float)(sums[c4 + u] / (double)R), c2[u] = (float)(sums[C + c4 + u] / (double)R);
    mu[u] = mean[c4 + u], is[u] = invstd[c4 + u], sc[u] = scale[c4 + u];
  }
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / c4n;
    const long long b = r / n;
    const int rl = (int)(r - b * n);
    const float4 v = *reinterpret_cast<const float4*>(y + r * C + c4);
    const float4 gv = *reinterpret_cast<const float4*>(g + b * C + c4);
    const int4 ar = *reinterpret_cast<const int4*>(arow + b * C + c4);
    const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    const int aa[4] = {ar.x, ar.y, ar.z, ar.w};
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (vv[u] - mu[u]) * is[u];
      d[u] = sc[u] * ((aa[u] == rl ? gg[u] : 0.f) - c1[u] - xh * c2[u]);
    }
    *reinterpret_cast<float4*>(dy + r * C + c4) = make_float4(d[0], d[1], d[2], d[3]);
  }
}

// ws [2 co][kp] = [W1; W2 - W1] zero-padded to kp columns, from the conv weight w [co][2 cin] = [W1 | W2]
__global__ void edge_weight_stack_kernel(int co, int cin, int kp, const float* __restrict__ w, float* __restrict__ ws) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * co * kp) return;
  const int o = t / kp, c = t - o * kp;
  float v = 0.f;
  if (c < cin) v = o < co ? w[o * 2 * cin + c] : w[(o - co) * 2 * cin + cin + c] - w[(o - co) * 2 * cin + c];
  ws[t] = v;
}
// its transpose: dw [co][2 cin] = [dWs_top - dWs_bottom | dWs_bottom]
__global__ void edge_weight_unstack_kernel(int co, int cin, int kp, const float* __restrict__ dws, float* __restrict__ dw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= co * 2 * cin) return;
  const int o = t / (2 * cin), c = t - o * 2 * cin;
  dw[t] = c < cin ? dws[o * kp + c] - dws[(co + o) * kp + c] : dws[(co + o) * kp + c - cin];
}
// the same for up to 8 layers in one launch (blockIdx.y = layer): the encoder's four EdgeConv weights per step
struct EdgeWeightJobs {
  int n;
  int co[8], cin[8], kp[8];
  const float* src[8];
  float* dst[8];
};
__global__ void edge_weight_stack_multi_kernel(const EdgeWeightJobs j) {
  const int q = blockIdx.y, co = j.co[q], cin = j.cin[q], kp = j.kp[q];
  const float* w = j.src[q];
  float* ws = j.dst[q];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * co * kp) return;
  const int o = t / kp, c = t - o * kp;
  float v = 0.f;
  if (c < cin) v = o < co ? w[o * 2 * cin + c] : w[(o - co) * 2 * cin + cin + c] - w[(o - co) * 2 * cin + c];
  ws[t] = v;
}
__global__ void edge_weight_unstack_multi_kernel(const EdgeWeightJobs j) {
  const int q = blockIdx.y, co = j.co[q], cin = j.cin[q], kp = j.kp[q];
  const float* dws = j.src[q];
  float* dw = j.dst[q];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= co * 2 * cin) return;
  const int o = t / (2 * cin), c = t - o * 2 * cin;
  dw[t] = c < cin ? dws[o * kp + c] - dws[(co + o) * kp + c] : dws[(co + o) * kp + c - cin];
}
// out [R][cp] = x [R][c] with zero columns appended
__global__ void rows_pad_kernel(long long total, int c, int cp, const float* __restrict__ x, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long long r = t / cp;
  const int j = (int)(t - r * cp);
  out[t] = j < c ? x[r * c + j] : 0.f;
}

static long long gcd_ll(long long a, long long b) { return b ? gcd_ll(b, a % b) : a; }

static int part_reduce(hipStream_t s, int P, int width, const double* part, double* out, float* fa, float* fb) {
  hipLaunchKernelGGL(part_reduce_f64_kernel, dim3((width + kWave - 1) / kWave), dim3(1024), 0, s, P, width, part, out, fa, fb);
  return check_launch("part_reduce");
}

constexpr int kEdgeBlocks = 512;      // partial rows of the streaming reductions
constexpr int kGatherBlocks = 1024;   // the gather pass wants every wave slot (latency-bound on L2 gathers)

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_rows_sqnorm(int R, int C, const float* x, float* xx, pdae_stream_t stream) {
  if (R < 0 || C <= 0 || C % 4 != 0) return bad_arg("rows_sqnorm: C must be a positive multiple of 4");
  if (R == 0) return PDAE_OK;
  if (!x || !xx) return bad_arg("rows_sqnorm: null pointer");
  hipLaunchKernelGGL(rows_sqnorm_kernel, dim3((R + 15) / 16), dim3(256), 0, as_stream(stream), R, C, x, xx);
  return check_launch("rows_sqnorm");
}

extern "C" int pdae_gram_topk(int b, int n, int k, const float* gram, const float* xx, int* idx, pdae_stream_t stream) {
  if (b < 0 || n <= 0 || k <= 0) return bad_arg("gram_topk: b>=0, n>0, k>0 required");
  if (k > n) return bad_arg("gram_topk: k > n");
  if (k > 64) return unsupported("gram_topk: k > 64 not implemented");
  if (b > 65535) return unsupported("gram_topk: b > 65535");
  if (b == 0) return PDAE_OK;
  if (!gram || !xx || !idx) return bad_arg("gram_topk: null pointer");
  hipLaunchKernelGGL(gram_topk_kernel<false>, dim3((n + 3) / 4, b), dim3(256), 0, as_stream(stream), n, k, gram, xx, idx,
                     static_cast<float*>(nullptr));
  return check_launch("gram_topk");
}

extern "C" int pdae_xyz_topk(int b, int n, int k, const float* x4, const float* xx, int* idx, float* pd_out,
                             pdae_stream_t stream) {
  if (b < 0 || n <= 0 || k <= 0) return bad_arg("xyz_topk: b>=0, n>0, k>0 required");
  if (k > n) return bad_arg("xyz_topk: k > n");
  if (k > 64) return unsupported("xyz_topk: k > 64 not implemented");
  if (b > 65535) return unsupported("xyz_topk: b > 65535");
  if (b == 0) return PDAE_OK;
  if (!x4 || !xx || !idx) return bad_arg("xyz_topk: null pointer");
  if (n > 6144) return unsupported("xyz_topk: clouds of more than 6144 points (the cloud is staged in LDS)");
  const size_t lds = (size_t)n * 20;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gram_topk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            6144 * 20) != hipSuccess)
      return unsupported("xyz_topk: dynamic LDS limit");
    attr = true;
  }
  hipLaunchKernelGGL(gram_topk_kernel<true>, dim3((n + 15) / 16, b), dim3(256), lds, as_stream(stream), n, k, x4, xx, idx, pd_out);
  return check_launch("xyz_topk");
}

extern "C" int pdae_knn_reverse(int b, int n, int k, const int* idx, int* rev_start, int* rev_src, pdae_stream_t stream) {
  if (b < 0 || n <= 0 || k <= 0 || k > n) return bad_arg("knn_reverse: b>=0, 0<k<=n required");
  if (n > 4096) return unsupported("knn_reverse: clouds of more than 4096 points");
  if (b == 0) return PDAE_OK;
  if (!idx || !rev_start || !rev_src) return bad_arg("knn_reverse: null pointer");
  // the largest chunk of source rows (W words of 32) whose bitmap [n][WS], WS odd >= W + 1 (bank spread of the per-target
  // walk), fits 156 KB beside cnt / cursor
  const int kLds = 156 * 1024;
  int W = (kLds - 8 * n) / 4 / n - 2;
  if (W < 1) W = 1;
  if (W > (n + 31) / 32) W = (n + 31) / 32;
  const int chunk = W * 32, WS = W + 1 + (W & 1);
  const size_t lds = (size_t)(2 * n + (size_t)n * WS) * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn_reverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLds) != hipSuccess)
      return unsupported("knn_reverse: dynamic LDS limit");
    attr = true;
  }
  hipLaunchKernelGGL(knn_reverse_kernel, dim3(b), dim3(1024), lds, as_stream(stream), n, k, chunk, idx, rev_start, rev_src);
  return check_launch("knn_reverse");
}

extern "C" int pdae_edge_parts(void) { return kGatherBlocks; }

static bool edge_width_ok(int co) { return co == 64 || co == 128 || co == 256 || co == 32 || co == 16 || co == 512 || co == 1024; }

extern "C" int pdae_edge_gather_stats(int b, int n, int k, int co, const float* pq, const int* idx, const float* gamma,
                                      float* esel, unsigned short* sel, float* psum, double* part, double* sums,
                                      pdae_stream_t stream) {
  if (b <= 0 || n <= 0 || k <= 0 || k > n) return bad_arg("edge_gather_stats: b>0, 0<k<=n required");
  if (!edge_width_ok(co)) return unsupported("edge_gather_stats: channel counts 16..1024 in powers of two");
  if (n > 65535) return unsupported("edge_gather_stats: clouds of more than 65535 points (16-bit winner ids)");
  if ((long long)b * n * 2 * co >= (1LL << 31)) return unsupported("edge_gather_stats: more than 2^31 elements");
  if (!pq || !idx || !gamma || !esel || !sel || !psum || !part || !sums) return bad_arg("edge_gather_stats: null pointer");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(edge_gather_stats_kernel, dim3(kGatherBlocks), dim3(256), 256 * 8 * sizeof(double), s, b * n, n, k, co,
                     pq, idx, gamma, esel, sel, psum, part);
  int rc = check_launch("edge_gather_stats");
  if (rc != PDAE_OK) return rc;
  return part_reduce(s, kGatherBlocks, 2 * co, part, sums, nullptr, nullptr);
}

extern "C" int pdae_bn_lrelu_rows(long long R, int C, const float* e, const float* scale, const float* shift, float* out,
                                  float* out2, int ld2, pdae_stream_t stream) {
  if (R < 0 || C <= 0 || C % 4 != 0) return bad_arg("bn_lrelu_rows: C must be a positive multiple of 4");
  if (R == 0) return PDAE_OK;
  if (!e || !scale || !shift || !out || (out2 && (ld2 < C || ld2 % 4 != 0))) return bad_arg("bn_lrelu_rows: bad pointer / stride");
  long long blocks = (R * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_lrelu_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), R, C, e, scale, shift, out,
                     out2, ld2);
  return check_launch("bn_lrelu_rows");
}

extern "C" int pdae_bn_lrelu_backward_reduce(long long R, int C, const float* d1, const float* d2, int ld2, const float* e,
                                             const float* scale, const float* shift, const float* mean,
                                             const float* invstd, float* g, double* part, double* sums, float* dgamma,
                                             float* dbeta, pdae_stream_t stream) {
  if (R <= 0 || C <= 0 || C % 4 != 0 || (C / 4 < 256 && 256 % (C / 4) != 0) || (C / 4 >= 256 && C % 1024 != 0))
    return bad_arg("bn_lrelu_backward_reduce: C must be 4 x a divisor of 256, or a multiple of 1024");
  if ((!d1 && !d2) || (d2 && (ld2 < C || ld2 % 4 != 0))) return bad_arg("bn_lrelu_backward_reduce: no gradient operand / bad stride");
  if (!e || !scale || !shift || !mean || !invstd || !g || !part || !sums) return bad_arg("bn_lrelu_backward_reduce: null pointer");
  hipStream_t s = as_stream(stream);
  const int tpr = C / 4 < 256 ? C / 4 : 256, rpb = 256 / tpr;
  long long want = (R + rpb - 1) / rpb;
  const int blocks = (int)(want < kEdgeBlocks ? want : kEdgeBlocks);
  hipLaunchKernelGGL(bn_lrelu_backward_reduce_kernel, dim3(blocks), dim3(256), 256 * 8 * sizeof(double), s, R, C, d1, d2, ld2,
                     e, scale, shift, mean, invstd, g, part);
  int rc = check_launch("bn_lrelu_backward_reduce");
  if (rc != PDAE_OK) return rc;
  return part_reduce(s, blocks, 2 * C, part, sums, dbeta, dgamma);
}

extern "C" int pdae_edge_backward(int b, int n, int k, int co, const float* g, const float* pq, const unsigned short* sel,
                                  const float* psum, const int* rev_start, const int* rev_src, const float* scale,
                                  const float* mean, const float* invstd, const double* sums, float* dpq,
                                  pdae_stream_t stream) {
  if (b <= 0 || n <= 0 || k <= 0 || k > n) return bad_arg("edge_backward: b>0, 0<k<=n required");
  if (!edge_width_ok(co)) return unsupported("edge_backward: channel counts 16..1024 in powers of two");
  if ((long long)b * n * 2 * co >= (1LL << 31)) return unsupported("edge_backward: more than 2^31 elements");
  if (!g || !pq || !sel || !psum || !rev_start || !rev_src || !scale || !mean || !invstd || !sums || !dpq)
    return bad_arg("edge_backward: null pointer");
  const int rpb = 256 / (co / 4);
  long long want = ((long long)b * n + rpb - 1) / rpb;
  const int blocks = (int)((want < 4096 ? want : 4096) + 7) / 8 * 8;
  hipLaunchKernelGGL(edge_backward_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), b * n, n, k, co, g, pq, sel, psum,
                     rev_start, rev_src, scale, mean, invstd, sums, dpq);
  return check_launch("edge_backward");
}

extern "C" int pdae_cloud_pool_splits(int b, int n) {
  int rs = 1;
  while (rs < 16 && (long long)b * rs < 128 && n / (rs * 2) >= 32) rs *= 2;
  return rs;
}

extern "C" int pdae_cloud_pool_stats(int b, int n, int C, const float* y, const float* gamma, float* ysel, int* arow,
                                     float* pv, int* pr, double* part, double* sums, pdae_stream_t stream) {
  if (b <= 0 || n <= 0 || C <= 0 || C % 4 != 0) return bad_arg("cloud_pool_stats: b, n > 0, C a positive multiple of 4");
  if (b > 65535) return unsupported("cloud_pool_stats: b > 65535");
  if (!y || !gamma || !ysel || !arow || !pv || !pr || !part || !sums) return bad_arg("cloud_pool_stats: null pointer");
  hipStream_t s = as_stream(stream);
  const int rs = pdae_cloud_pool_splits(b, n);
  hipLaunchKernelGGL(cloud_pool_stats_kernel, dim3((C + 255) / 256, b, rs), dim3(256), 0, s, n, C, rs, y, gamma, pv, pr, part);
  int rc = check_launch("cloud_pool_stats");
  if (rc != PDAE_OK) return rc;
  hipLaunchKernelGGL(cloud_pool_merge_kernel, dim3((b * C + 255) / 256), dim3(256), 0, s, b * C, C, rs, pv, pr, gamma, ysel, arow);
  rc = check_launch("cloud_pool_merge");
  if (rc != PDAE_OK) return rc;
  return part_reduce(s, b * rs, 2 * C, part, sums, nullptr, nullptr);
}

extern "C" int pdae_cloud_pool_backward(int b, int n, int C, const float* y, const float* g, const int* arow,
                                        const float* scale, const float* mean, const float* invstd, const double* sums,
                                        float* dy, pdae_stream_t stream) {
  if (b <= 0 || n <= 0 || C <= 0 || C % 4 != 0) return bad_arg("cloud_pool_backward: b, n > 0, C a positive multiple of 4");
  if (!y || !g || !arow || !scale || !mean || !invstd || !sums || !dy) return bad_arg("cloud_pool_backward: null pointer");
  const long long R = (long long)b * n;
  // a grid whose thread count is a multiple of C / 4 (a thread then owns four channels for the whole sweep)
  const long long per = C / 4, unit = per / gcd_ll(per, 256);       // blocks come in multiples of `unit`
  long long blocks = (R * per + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  blocks = (blocks + unit - 1) / unit * unit;
  hipLaunchKernelGGL(cloud_pool_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), R, n, C, y, g, arow,
                     scale, mean, invstd, sums, dy);
  return check_launch("cloud_pool_backward");
}

extern "C" int pdae_edge_weight_stack(int co, int cin, int kp, const float* w, float* ws, pdae_stream_t stream) {
  if (co <= 0 || cin <= 0 || kp < cin) return bad_arg("edge_weight_stack: co, cin > 0, kp >= cin required");
  if (!w || !ws) return bad_arg("edge_weight_stack: null pointer");
  hipLaunchKernelGGL(edge_weight_stack_kernel, dim3((2 * co * kp + 255) / 256), dim3(256), 0, as_stream(stream), co, cin, kp, w, ws);
  return check_launch("edge_weight_stack");
}

extern "C" int pdae_edge_weight_unstack(int co, int cin, int kp, const float* dws, float* dw, pdae_stream_t stream) {
  if (co <= 0 || cin <= 0 || kp < cin) return bad_arg("edge_weight_unstack: co, cin > 0, kp >= cin required");
  if (!dws || !dw) return bad_arg("edge_weight_unstack: null pointer");
  hipLaunchKernelGGL(edge_weight_unstack_kernel, dim3((2 * co * cin + 255) / 256), dim3(256), 0, as_stream(stream), co, cin, kp, dws, dw);
  return check_launch("edge_weight_unstack");
}

static int edge_weight_multi(const char* what, bool unstack, int n, const int* co, const int* cin, const int* kp,
                             const float* const* src, float* const* dst, pdae_stream_t stream) {
  if (n < 0 || n > 8 || (n > 0 && (!co || !cin || !kp || !src || !dst))) return bad_arg(what);
  if (n == 0) return PDAE_OK;
  EdgeWeightJobs j = {};
  j.n = n;
  int most = 0;
  for (int q = 0; q < n; ++q) {
    if (co[q] <= 0 || cin[q] <= 0 || kp[q] < cin[q] || !src[q] || !dst[q]) return bad_arg(what);
    j.co[q] = co[q], j.cin[q] = cin[q], j.kp[q] = kp[q], j.src[q] = src[q], j.dst[q] = dst[q];
    const int e = unstack ? 2 * co[q] * cin[q] : 2 * co[q] * kp[q];
    most = e > most ? e : most;
  }
  if (unstack) hipLaunchKernelGGL(edge_weight_unstack_multi_kernel, dim3((most + 255) / 256, n), dim3(256), 0, as_stream(stream), j);
  else hipLaunchKernelGGL(edge_weight_stack_multi_kernel, dim3((most + 255) / 256, n), dim3(256), 0, as_stream(stream), j);
  return check_launch(what);
}
extern "C" int pdae_edge_weight_stack_multi(int n, const int* co, const int* cin, const int* kp, const float* const* w,
                                            float* const* ws, pdae_stream_t stream) {
  return edge_weight_multi("edge_weight_stack_multi: 0..8 layers, co, cin > 0, kp >= cin, no null pointer", false, n, co, cin, kp, w, ws, stream);
}
extern "C" int pdae_edge_weight_unstack_multi(int n, const int* co, const int* cin, const int* kp, const float* const* dws,
                                              float* const* dw, pdae_stream_t stream) {
  return edge_weight_multi("edge_weight_unstack_multi: 0..8 layers, co, cin > 0, kp >= cin, no null pointer", true, n, co, cin, kp, dws, dw, stream);
}

extern "C" int pdae_rows_pad(long long R, int c, int cp, const float* x, float* out, pdae_stream_t stream) {
  if (R < 0 || c <= 0 || cp < c) return bad_arg("rows_pad: R >= 0, 0 < c <= cp required");
  if (R == 0) return PDAE_OK;
  if (!x || !out) return bad_arg("rows_pad: null pointer");
  hipLaunchKernelGGL(rows_pad_kernel, dim3((unsigned)((R * cp + 255) / 256)), dim3(256), 0, as_stream(stream), R * cp, c, cp, x, out);
  return check_launch("rows_pad");
}
