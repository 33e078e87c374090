// chamfer.hip -- Chamfer distance forward / backward for gfx950.
//
// Semantics: extensions/chamfer_dist/chamfer.cu:15-145 (forward) and :173-201
// (backward) of the reference, restated in oracle/pdae_oracle.c.  For every
// point of one cloud: squared distance to, and index of, the nearest point of
// the other cloud, lowest index on ties.
//
// The reference launches <<<dim3(32,16),512>>> whatever the shape; with the
// Transformer path's ~5000 clouds of 32 points only 32 lanes of each block do
// work and the backward serialises the batch over 16 blocks.
// Here the grid follows the data:
//   * packed kernel (small clouds): queries of consecutive clouds are packed
//     densely into 256-thread workgroups; the candidate clouds those queries
//     need are staged once in LDS as float4 and read back as broadcasts.
//   * tiled kernel (large clouds): Q query points per lane in registers,
//     candidates streamed through a 1024-point LDS tile.
// Backward has a deterministic gather form for small clouds (bit-identical to
// the oracle's ordering) and an atomic scatter form for large ones (the
// reference itself accumulates with atomicAdd in unspecified order).
#include "common.h"

namespace pdae {

constexpr int kPackT = 256;

// ---- forward, packed small clouds ------------------------------------------
// thread -> query row r = blockIdx.x*256 + tid of the flattened (b*n) queries.
__global__ __launch_bounds__(kPackT) void chamfer_fwd_packed(
    int b, int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
    float* __restrict__ dist, int32_t* __restrict__ idx) {
  extern __shared__ float4 cand[];  // [(c1-c0+1) * m]
  const long long total = (long long)b * n;
  const long long r0 = (long long)blockIdx.x * kPackT;
  const long long r1 = min(total, r0 + kPackT) - 1;
  const int c0 = (int)(r0 / n), c1 = (int)(r1 / n);
  const int ncand = (c1 - c0 + 1) * m;
  const float* src = xyz2 + (size_t)c0 * m * 3;
  for (int i = threadIdx.x; i < ncand; i += kPackT)
    cand[i] = make_float4(src[i * 3 + 0], src[i * 3 + 1], src[i * 3 + 2], 0.f);
  __syncthreads();
  const long long r = r0 + threadIdx.x;
  if (r >= total) return;
  const int c = (int)(r / n);
  const float x1 = xyz1[r * 3 + 0], y1 = xyz1[r * 3 + 1], z1 = xyz1[r * 3 + 2];
  const float4* my = cand + (c - c0) * m;
  float best = __builtin_huge_valf();
  int besti = 0;
#pragma unroll 4
  for (int k = 0; k < m; ++k) {
    const float4 q = my[k];
    const float d = sqdist(q.x, q.y, q.z, x1, y1, z1);
    if (d < best) {
      best = d;
      besti = k;
    }
  }
  dist[r] = best;
  idx[r] = besti;
}

// ---- forward, tiled large clouds -------------------------------------------
template <int Q>
__global__ __launch_bounds__(256) void chamfer_fwd_tiled(int n, int m,
                                                         const float* __restrict__ xyz1,
                                                         const float* __restrict__ xyz2,
                                                         float* __restrict__ dist,
                                                         int32_t* __restrict__ idx) {
  constexpr int TILE = 1024;
  __shared__ float4 cand[TILE];
  const int bi = blockIdx.y;
  const float* p1 = xyz1 + (size_t)bi * n * 3;
  const float* p2 = xyz2 + (size_t)bi * m * 3;
  float x1[Q], y1[Q], z1[Q], best[Q];
  int besti[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int j = (blockIdx.x * Q + i) * 256 + threadIdx.x;
    const bool in = j < n;
    x1[i] = in ? p1[j * 3 + 0] : 0.f;
    y1[i] = in ? p1[j * 3 + 1] : 0.f;
    z1[i] = in ? p1[j * 3 + 2] : 0.f;
    best[i] = __builtin_huge_valf();
    besti[i] = 0;
  }
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 256)
      cand[i] = make_float4(p2[(k0 + i) * 3 + 0], p2[(k0 + i) * 3 + 1], p2[(k0 + i) * 3 + 2], 0.f);
    __syncthreads();
    if constexpr (Q % 2 == 0) {
      // two queries per packed instruction: v_pk_add_f32 / v_pk_mul_f32 evaluate ((dx*dx + dy*dy) + dz*dz) for a PAIR
      // of this lane's queries at once -- the same eight roundings per pair as the scalar form (no FMA), half the
      // issue slots; the strict `<` update (lowest index on ties, chamfer.cu:15-145) stays per query
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 px[Q / 2], py[Q / 2], pz[Q / 2];
#pragma unroll
      for (int i = 0; i < Q / 2; ++i) {
        px[i] = f32x2{x1[2 * i], x1[2 * i + 1]}, py[i] = f32x2{y1[2 * i], y1[2 * i + 1]};
        pz[i] = f32x2{z1[2 * i], z1[2 * i + 1]};
      }
#pragma unroll 4
      for (int k = 0; k < cnt; ++k) {
        const float4 q = cand[k];
        const f32x2 qx = f32x2{q.x, q.x}, qy = f32x2{q.y, q.y}, qz = f32x2{q.z, q.z};
#pragma unroll
        for (int i = 0; i < Q / 2; ++i) {
          const f32x2 dx = qx - px[i], dy = qy - py[i], dz = qz - pz[i];
          const f32x2 d = (dx * dx + dy * dy) + dz * dz;
          if (d.x < best[2 * i]) best[2 * i] = d.x, besti[2 * i] = k0 + k;
          if (d.y < best[2 * i + 1]) best[2 * i + 1] = d.y, besti[2 * i + 1] = k0 + k;
        }
      }
    } else {
#pragma unroll 2
      for (int k = 0; k < cnt; ++k) {
        const float4 q = cand[k];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          const float d = sqdist(q.x, q.y, q.z, x1[i], y1[i], z1[i]);
          if (d < best[i]) {
            best[i] = d;
            besti[i] = k0 + k;
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int j = (blockIdx.x * Q + i) * 256 + threadIdx.x;
    if (j < n) {
      dist[(size_t)bi * n + j] = best[i];
      idx[(size_t)bi * n + j] = besti[i];
    }
  }
}

// ---- forward, FEW queries against MANY candidates (the 1024-point ground truth against the 16384-point fine cloud of
// Point_CAE_PointNetv2): one query per lane cannot hide the compare / select chain of a 16384-step scan (876 us at
// B = 128), so every lane runs FOUR running minima over the candidates k = 0, 1, 2, 3 (mod 4) -- two packed pairs -- and
// merges them at the end: the smallest distance, the LOWEST index among equal ones = what the strict `<` scan in index
// order returns (chamfer.cu:15-145).  Candidates are staged as x / y / z rows so that two consecutive ones load as one
// 8-byte LDS read into a register pair for v_pk_add_f32 / v_pk_mul_f32.
__global__ __launch_bounds__(256) void chamfer_fwd_many(int n, int m, const float* __restrict__ xyz1,
                                                        const float* __restrict__ xyz2, float* __restrict__ dist,
                                                        int32_t* __restrict__ idx) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int TILE = 2048;
  __shared__ float sx[TILE], sy[TILE], sz[TILE];
  const int bi = blockIdx.y;
  const float* p1 = xyz1 + (size_t)bi * n * 3;
  const float* p2 = xyz2 + (size_t)bi * m * 3;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool in = j < n;
  const float x1 = in ? p1[j * 3 + 0] : 0.f, y1 = in ? p1[j * 3 + 1] : 0.f, z1 = in ? p1[j * 3 + 2] : 0.f;
  const f32x2 qx = f32x2{x1, x1}, qy = f32x2{y1, y1}, qz = f32x2{z1, z1};
  float best[4];
  int besti[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) best[a] = __builtin_huge_valf(), besti[a] = 0;
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < TILE; i += 256) {
      const bool ok = i < cnt;                       // the tail of the last tile: +inf never wins a strict `<`
      sx[i] = ok ? p2[(size_t)(k0 + i) * 3 + 0] : __builtin_huge_valf();
      sy[i] = ok ? p2[(size_t)(k0 + i) * 3 + 1] : 0.f;
      sz[i] = ok ? p2[(size_t)(k0 + i) * 3 + 2] : 0.f;
    }
    __syncthreads();
    const int lim = (cnt + 3) & ~3;
#pragma unroll 2
    for (int k = 0; k < lim; k += 4) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 cx = *reinterpret_cast<const f32x2*>(sx + k + 2 * h);
        const f32x2 cy = *reinterpret_cast<const f32x2*>(sy + k + 2 * h);
        const f32x2 cz = *reinterpret_cast<const f32x2*>(sz + k + 2 * h);
        const f32x2 dx = cx - qx, dy = cy - qy, dz = cz - qz;
        const f32x2 d = (dx * dx + dy * dy) + dz * dz;
        if (d.x < best[2 * h]) best[2 * h] = d.x, besti[2 * h] = k0 + k + 2 * h;
        if (d.y < best[2 * h + 1]) best[2 * h + 1] = d.y, besti[2 * h + 1] = k0 + k + 2 * h + 1;
      }
    }
  }
  float b = best[0];
  int bidx = besti[0];
#pragma unroll
  for (int a = 1; a < 4; ++a)
    if (best[a] < b || (best[a] == b && besti[a] < bidx)) b = best[a], bidx = besti[a];
  if (in) {
    dist[(size_t)bi * n + j] = b;
    idx[(size_t)bi * n + j] = bidx;
  }
}

static bool use_packed(int n, int m) {
  // LDS for the clouds a 256-query workgroup can touch
  const long long clouds = (kPackT + n - 1) / n + 1;
  return n <= 256 && clouds * m * (long long)sizeof(float4) <= 64 * 1024;
}

static void chamfer_fwd_dir(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist,
                            int32_t* idx, hipStream_t s) {
  if (n == 0) return;
  if (use_packed(n, m)) {
    const long long total = (long long)b * n;
    const int blocks = (int)((total + kPackT - 1) / kPackT);
    const size_t lds = (size_t)((kPackT + n - 1) / n + 1) * m * sizeof(float4);
    hipLaunchKernelGGL(chamfer_fwd_packed, dim3(blocks), dim3(kPackT), lds, s, b, n, m, xyz1, xyz2,
                       dist, idx);
  } else if ((long long)b * ((n + 1023) / 1024) >= 1024 || n >= 4096) {
    dim3 grid((n + 1023) / 1024, b);
    hipLaunchKernelGGL((chamfer_fwd_tiled<4>), grid, dim3(256), 0, s, n, m, xyz1, xyz2, dist, idx);
  } else if (m >= 2048) {
    dim3 grid((n + 255) / 256, b);
    hipLaunchKernelGGL(chamfer_fwd_many, grid, dim3(256), 0, s, n, m, xyz1, xyz2, dist, idx);
  } else {
    dim3 grid((n + 255) / 256, b);
    hipLaunchKernelGGL((chamfer_fwd_tiled<1>), grid, dim3(256), 0, s, n, m, xyz1, xyz2, dist, idx);
  }
}

// ---- backward, gather form (small clouds) ----------------------------------
// grad_a[j] = own(j) + sum over l with idx_b[l]==j of -(2 g_b[l] (b_l - a_j)),
// scatter terms in ascending l.  own_first selects whether the own term is
// added before (direction 1 of the oracle) or after (direction 2) them.
__global__ __launch_bounds__(kPackT) void chamfer_bwd_packed(
    int b, int n, int m, const float* __restrict__ xa, const float* __restrict__ xb,
    const int32_t* __restrict__ idx_a, const int32_t* __restrict__ idx_b,
    const float* __restrict__ g_a, const float* __restrict__ g_b, float* __restrict__ grad_a,
    int own_first, int gs, float gdiv_a, float gdiv_b) {
  // gs / gdiv: the distance gradients are g_a[r * gs] / gdiv_a -- gs 1, gdiv 1: the arrays as given; gs 0: ONE device
  // scalar divided by the element count, i.e. the backward of a mean without materialising its expand / div
  extern __shared__ float4 other[];  // per candidate: (x, y, z, 2*g) of cloud b
  const long long total = (long long)b * n;
  const long long r0 = (long long)blockIdx.x * kPackT;
  const long long r1 = min(total, r0 + kPackT) - 1;
  const int c0 = (int)(r0 / n), c1 = (int)(r1 / n);
  const int ncand = (c1 - c0 + 1) * m;
  int* other_idx = reinterpret_cast<int*>(other + ncand);
  const size_t base = (size_t)c0 * m;
  for (int i = threadIdx.x; i < ncand; i += kPackT) {
    other[i] = make_float4(xb[(base + i) * 3 + 0], xb[(base + i) * 3 + 1],
                           xb[(base + i) * 3 + 2], (g_b[(base + i) * gs] / gdiv_b) * 2);
    other_idx[i] = idx_b[base + i];
  }
  __syncthreads();
  const long long r = r0 + threadIdx.x;
  if (r >= total) return;
  const int c = (int)(r / n);
  const int j = (int)(r - (long long)c * n);
  const float x1 = xa[r * 3 + 0], y1 = xa[r * 3 + 1], z1 = xa[r * 3 + 2];
  const float4* ob = other + (c - c0) * m;
  const int* oi = other_idx + (c - c0) * m;
  // own term: g (a_j - b_{idx_a[j]}), chamfer.cu:192-195
  const int j2 = idx_a[r];
  const float g = (g_a[r * gs] / gdiv_a) * 2;
  const float ox = g * (x1 - ob[j2].x), oy = g * (y1 - ob[j2].y), oz = g * (z1 - ob[j2].z);
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (own_first) {
    gx += ox;
    gy += oy;
    gz += oz;
  }
  for (int l = 0; l < m; ++l) {
    if (oi[l] == j) {  // chamfer.cu:196-198 with the roles of the clouds swapped
      const float4 q = ob[l];
      gx += -(q.w * (q.x - x1));
      gy += -(q.w * (q.y - y1));
      gz += -(q.w * (q.z - z1));
    }
  }
  if (!own_first) {
    gx += ox;
    gy += oy;
    gz += oz;
  }
  grad_a[r * 3 + 0] = gx;
  grad_a[r * 3 + 1] = gy;
  grad_a[r * 3 + 2] = gz;
}

// ---- backward, scatter form (large clouds) ----------------------------------
// pass 1 (plain stores): grad_a[j] = own(j).  pass 2: atomics for the scatter.
__global__ void chamfer_bwd_own(long long total, int n, int m, const float* __restrict__ xa,
                                const float* __restrict__ xb, const int32_t* __restrict__ idx_a,
                                const float* __restrict__ g_a, float* __restrict__ grad_a, int gs,
                                float gdiv_a) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= total) return;
  const long long c = r / n;
  const size_t o = ((size_t)c * m + idx_a[r]) * 3;
  const float g = (g_a[r * gs] / gdiv_a) * 2;
  grad_a[r * 3 + 0] = g * (xa[r * 3 + 0] - xb[o + 0]);
  grad_a[r * 3 + 1] = g * (xa[r * 3 + 1] - xb[o + 1]);
  grad_a[r * 3 + 2] = g * (xa[r * 3 + 2] - xb[o + 2]);
}

__global__ void chamfer_bwd_scatter(long long total, int n, int m, const float* __restrict__ xa,
                                    const float* __restrict__ xb,
                                    const int32_t* __restrict__ idx_a,
                                    const float* __restrict__ g_a, float* __restrict__ grad_b, int gs,
                                    float gdiv_a) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= total) return;
  const long long c = r / n;
  const size_t o = ((size_t)c * m + idx_a[r]) * 3;
  const float g = (g_a[r * gs] / gdiv_a) * 2;
  atomicAdd(grad_b + o + 0, -(g * (xa[r * 3 + 0] - xb[o + 0])));
  atomicAdd(grad_b + o + 1, -(g * (xa[r * 3 + 1] - xb[o + 1])));
  atomicAdd(grad_b + o + 2, -(g * (xa[r * 3 + 2] - xb[o + 2])));
}

// The same scatter for MANY queries onto FEW targets (cfg2's fine loss: 16384 predictions onto 1024 ground-truth points
// per cloud, 16+ colliding global atomics per address: 600 us of the 687 us this step spent in the kernel above).  A
// cloud's queries are ordered by target with a counting sort in LDS, then summed run by run (below).
template <int SPLIT>
__global__ __launch_bounds__(256) void chamfer_bwd_scatter_sorted(int n, int m, const float* __restrict__ xa,
                                                                  const float* __restrict__ xb,
                                                                  const int32_t* __restrict__ idx_a,
                                                                  const float* __restrict__ g_a, float* __restrict__ grad_b,
                                                                  int gs, float gdiv_a) {
  extern __shared__ int cs_lds[];                // cnt[m] | off[m + 1] | list[n]
  int* cnt = cs_lds;
  int* off = cs_lds + m;
  int* list = off + m + 1;
  const int c = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT, tid = threadIdx.x;
  const int32_t* ia = idx_a + (size_t)c * n;
  for (int i = tid; i < m; i += 256) cnt[i] = 0;
  __syncthreads();
  // Neighbouring predictions share their nearest target (the folded surface is spatially coherent): lanes of a wave
  // that hold the same target in a row form a RUN, its first lane does ONE LDS atomic for the run (colliding LDS atomics
  // serialise: one per lane took 710 us for the fine loss, more than the global float atomics they replaced).
  auto run_of = [&](int t, bool valid, int& start, int& len) {
    const int lane = tid & 63;
    const int prev = __shfl_up(t, 1, 64);
    const bool head = valid && (lane == 0 || t != prev || !__shfl_up((int)valid, 1, 64));
    const unsigned long long heads = __ballot(head), live = __ballot(valid);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);           // heads at or below this lane
    start = 63 - __clzll(below | 1ull);
    unsigned long long above = heads & ~((2ull << lane) - 1ull);                // next head above
    const int end_head = above ? __ffsll((long long)above) - 1 : 64;
    const int last_live = live ? 64 - __clzll(live) : 0;
    len = (end_head < last_live ? end_head : last_live) - start;
    return head;
  };
  for (int r0 = 0; r0 < n; r0 += 256) {
    const int r = r0 + tid;
    const bool valid = r < n;
    const int t = valid ? ia[r] : -1;
    int start, len;
    if (run_of(t, valid, start, len)) atomicAdd(&cnt[t], len);
  }
  __syncthreads();
  if (tid < 64) {                                // exclusive prefix sum: m / 64 entries per lane + a shuffle scan
    const int per = (m + 63) / 64, j0 = tid * per;
    int sum = 0;
    for (int k = 0; k < per; ++k)
      if (j0 + k < m) sum += cnt[j0 + k];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (tid >= d) incl += t;
    }
    int run = incl - sum;
    for (int k = 0; k < per; ++k)
      if (j0 + k < m) {
        off[j0 + k] = run;
        run += cnt[j0 + k];
      }
    if (tid == 63) off[m] = incl;
  }
  __syncthreads();
  for (int i = tid; i < m; i += 256) cnt[i] = off[i];
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 256) {
    const int r = r0 + tid;
    const bool valid = r < n;
    const int t = valid ? ia[r] : -1;
    int start, len;
    int base = 0;
    if (run_of(t, valid, start, len)) base = atomicAdd(&cnt[t], len);
    base = __shfl(base, start, 64);                                  // the run's first lane holds its slot base
    if (valid) list[base + ((tid & 63) - start)] = r;
  }
  __syncthreads();
  // ---- sums.  The counts per target are heavy-tailed (early in training most predictions sit on a few targets): one
  // thread per target was a serial loop over thousands of terms (741 us).  Instead the SORTED list is walked by position,
  // 64 entries per wave; equal targets are adjacent, so a segmented shuffle scan leaves every run's total in its last
  // lane, which adds it to the target's LDS accumulator (a few float atomics per wave, not one per term).
  float* acc = reinterpret_cast<float*>(list + n);                 // [m][3]
  for (int i = tid; i < 3 * m; i += 256) acc[i] = 0.f;
  __syncthreads();
  // this block's share: the targets whose bins START in its quarter of the list positions -- whole bins (the order
  // INSIDE a bin is this block's own order of arrival and differs between the blocks of a cloud; the bin boundaries,
  // off[], do not), so every target has exactly one owner
  auto first_bin_at_or_after = [&](int pos) {          // lower bound over off[0..m]
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] < pos) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int t_lo = first_bin_at_or_after((int)((long long)n * part / SPLIT));
  const int t_hi = part == SPLIT - 1 ? m : first_bin_at_or_after((int)((long long)n * (part + 1) / SPLIT));
  const int k0 = off[t_lo], k1 = off[t_hi];
  const float* xac = xa + (size_t)c * n * 3;
  const float* gac = g_a + (size_t)c * n * gs;
  const float* xbc = xb + (size_t)c * m * 3;
  for (int kb = k0; kb < k1; kb += 256) {
    const int k = kb + tid, lane = tid & 63;
    const bool valid = k < k1;
    const int r = valid ? list[k] : 0;
    const int t = valid ? ia[r] : -1;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (valid) {
      const float g = (gac[(size_t)r * gs] / gdiv_a) * 2;
      vx = g * (xac[r * 3] - xbc[t * 3]), vy = g * (xac[r * 3 + 1] - xbc[t * 3 + 1]), vz = g * (xac[r * 3 + 2] - xbc[t * 3 + 2]);
    }
    int start, len;
    run_of(t, valid, start, len);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float ux = __shfl_up(vx, d, 64), uy = __shfl_up(vy, d, 64), uz = __shfl_up(vz, d, 64);
      if (lane - d >= start) vx += ux, vy += uy, vz += uz;
    }
    if (valid && lane == start + len - 1) {
      atomicAdd(&acc[t * 3], vx), atomicAdd(&acc[t * 3 + 1], vy), atomicAdd(&acc[t * 3 + 2], vz);
    }
  }
  __syncthreads();
  float* gb = grad_b + (size_t)c * m * 3;
  for (int i = 3 * t_lo + tid; i < 3 * t_hi; i += 256) gb[i] -= acc[i];      // one owner per target: plain update
}

static bool use_packed_bwd(int n, int m) {
  const long long clouds = (kPackT + n - 1) / n + 1;
  return n <= 256 && m <= 256 && clouds * m * 20ll <= 64 * 1024;
}

}  // namespace pdae

extern "C" int pdae_chamfer_forward(int b, int n, const float* xyz1, int m, const float* xyz2,
                                    float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                                    pdae_stream_t stream) {
  using namespace pdae;
  if (b < 0 || n < 0 || m < 0) return bad_arg("chamfer_forward: negative size");
  if (b == 0 || (n == 0 && m == 0)) return PDAE_OK;
  if (n == 0 || m == 0) return bad_arg("chamfer_forward: one cloud is empty");
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2)
    return bad_arg("chamfer_forward: null pointer");
  if (b > 65535 && !(use_packed(n, m) && use_packed(m, n)))
    return unsupported("chamfer_forward: b > 65535 with large clouds");
  hipStream_t s = as_stream(stream);
  chamfer_fwd_dir(b, n, m, xyz1, xyz2, dist1, idx1, s);
  chamfer_fwd_dir(b, m, n, xyz2, xyz1, dist2, idx2, s);
  return check_launch("chamfer_forward");
}

static int chamfer_backward_impl(int b, int n, const float* xyz1, int m, const float* xyz2, const int32_t* idx1,
                                 const int32_t* idx2, const float* grad_dist1, const float* grad_dist2, int gs,
                                 float div1, float div2, float* grad_xyz1, float* grad_xyz2, pdae_stream_t stream,
                                 const char* what) {
  using namespace pdae;
  if (b < 0 || n < 0 || m < 0) return bad_arg("chamfer_backward: negative size");
  if (b == 0 || (n == 0 && m == 0)) return PDAE_OK;
  if (n == 0 || m == 0) return bad_arg("chamfer_backward: one cloud is empty");
  if (!xyz1 || !xyz2 || !idx1 || !idx2 || !grad_dist1 || !grad_dist2 || !grad_xyz1 || !grad_xyz2)
    return bad_arg("chamfer_backward: null pointer");
  hipStream_t s = as_stream(stream);
  const long long t1 = (long long)b * n, t2 = (long long)b * m;
  if (use_packed_bwd(n, m) && use_packed_bwd(m, n)) {
    const size_t lds1 = (size_t)((kPackT + n - 1) / n + 1) * m * 20;
    const size_t lds2 = (size_t)((kPackT + m - 1) / m + 1) * n * 20;
    hipLaunchKernelGGL(chamfer_bwd_packed, dim3((unsigned)((t1 + kPackT - 1) / kPackT)),
                       dim3(kPackT), lds1, s, b, n, m, xyz1, xyz2, idx1, idx2, grad_dist1,
                       grad_dist2, grad_xyz1, 1, gs, div1, div2);
    hipLaunchKernelGGL(chamfer_bwd_packed, dim3((unsigned)((t2 + kPackT - 1) / kPackT)),
                       dim3(kPackT), lds2, s, b, m, n, xyz2, xyz1, idx2, idx1, grad_dist2,
                       grad_dist1, grad_xyz2, 0, gs, div2, div1);
  } else {
    const int T = 256;
    hipLaunchKernelGGL(chamfer_bwd_own, dim3((unsigned)((t1 + T - 1) / T)), dim3(T), 0, s, t1, n, m,
                       xyz1, xyz2, idx1, grad_dist1, grad_xyz1, gs, div1);
    hipLaunchKernelGGL(chamfer_bwd_own, dim3((unsigned)((t2 + T - 1) / T)), dim3(T), 0, s, t2, m, n,
                       xyz2, xyz1, idx2, grad_dist2, grad_xyz2, gs, div2);
    // many queries onto few targets: the sorted form (no colliding float atomics); else atomics (few collisions)
    auto scatter = [&](long long tq, int nq, int mt, const float* xq, const float* xt, const int32_t* iq, const float* gq,
                       float* grad_t, float div) {
      constexpr int SPLIT = 4;
      const size_t lds = sizeof(int) * ((size_t)5 * mt + 1 + nq);
      if (nq >= 4 * mt && lds <= 150 * 1024 && (long long)b * SPLIT <= 0x7fffffffLL) {
        static bool once = false;
        if (!once) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chamfer_bwd_scatter_sorted<SPLIT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
          once = true;
        }
        hipLaunchKernelGGL(chamfer_bwd_scatter_sorted<SPLIT>, dim3((unsigned)(b * SPLIT)), dim3(256), lds, s, nq, mt, xq, xt,
                           iq, gq, grad_t, gs, div);
      } else {
        hipLaunchKernelGGL(chamfer_bwd_scatter, dim3((unsigned)((tq + T - 1) / T)), dim3(T), 0, s, tq, nq, mt, xq, xt, iq,
                           gq, grad_t, gs, div);
      }
    };
    scatter(t1, n, m, xyz1, xyz2, idx1, grad_dist1, grad_xyz2, div1);
    scatter(t2, m, n, xyz2, xyz1, idx2, grad_dist2, grad_xyz1, div2);
  }
  return check_launch(what);
}

extern "C" int pdae_chamfer_backward(int b, int n, const float* xyz1, int m, const float* xyz2,
                                     const int32_t* idx1, const int32_t* idx2,
                                     const float* grad_dist1, const float* grad_dist2,
                                     float* grad_xyz1, float* grad_xyz2, pdae_stream_t stream) {
  return chamfer_backward_impl(b, n, xyz1, m, xyz2, idx1, idx2, grad_dist1, grad_dist2, 1, 1.0f, 1.0f, grad_xyz1,
                               grad_xyz2, stream, "chamfer_backward");
}

extern "C" int pdae_chamfer_backward_mean(int b, int n, const float* xyz1, int m, const float* xyz2,
                                          const int32_t* idx1, const int32_t* idx2, const float* grad_loss,
                                          float* grad_xyz1, float* grad_xyz2, pdae_stream_t stream) {
  // the gradient of mean(dist1) + mean(dist2): grad_dist1 = grad_loss / (b n), grad_dist2 = grad_loss / (b m)
  return chamfer_backward_impl(b, n, xyz1, m, xyz2, idx1, idx2, grad_loss, grad_loss, 0, (float)((long long)b * n),
                               (float)((long long)b * m), grad_xyz1, grad_xyz2, stream, "chamfer_backward_mean");
}

namespace pdae {
// out[0] = mean(a) + mean(b) in two small launches with a fixed summation order: MS_BLOCKS blocks leave one partial
// sum per input (thread-strided float4 reads, wave shuffles, LDS slots in order), one wave adds the partials in order.
constexpr int MS_BLOCKS = 128;
__global__ __launch_bounds__(256) void mean_sum2_partial_kernel(long long na, const float* __restrict__ a, long long nb,
                                                                const float* __restrict__ b, float* __restrict__ part) {
  __shared__ float red[2][4];
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)MS_BLOCKS * 256;
  float sa = 0.f, sb = 0.f;
  for (long long i = t; i < na; i += stride) sa += a[i];
  for (long long i = t; i < nb; i += stride) sb += b[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sa += __shfl_xor(sa, o, kWave), sb += __shfl_xor(sb, o, kWave);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = sa, red[1][threadIdx.x >> 6] = sb;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    part[MS_BLOCKS + blockIdx.x] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  }
}
__global__ __launch_bounds__(64) void mean_sum2_final_kernel(long long na, long long nb, const float* __restrict__ part,
                                                             float* __restrict__ out) {
  const int l = threadIdx.x;
  float sa = part[l] + part[l + 64], sb = part[MS_BLOCKS + l] + part[MS_BLOCKS + l + 64];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sa += __shfl_xor(sa, o, kWave), sb += __shfl_xor(sb, o, kWave);
  if (l == 0) out[0] = sa / (float)na + sb / (float)nb;
}
}  // namespace pdae

extern "C" int pdae_mean_sum2(long long na, const float* a, long long nb, const float* b, float* workspace, float* out,
                              pdae_stream_t stream) {
  using namespace pdae;
  static_assert(MS_BLOCKS == 128, "the final kernel adds two partials per lane");
  if (na <= 0 || nb <= 0 || !a || !b || !out || !workspace) return bad_arg("mean_sum2: empty input or null pointer");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(mean_sum2_partial_kernel, dim3(MS_BLOCKS), dim3(256), 0, s, na, a, nb, b, workspace);
  hipLaunchKernelGGL(mean_sum2_final_kernel, dim3(1), dim3(64), 0, s, na, nb, workspace, out);
  return check_launch("mean_sum2");
}
