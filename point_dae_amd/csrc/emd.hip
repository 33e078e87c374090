// emd.hip -- approximate earth mover's distance (auction-style soft matching).
//
// Semantics: extensions/emd/cuda/emd_kernel.cu:25-158 (approxmatch), :200-243
// (matchcost), :286-355 (matchcostgrad2 / matchcostgrad1) of the reference,
// restated in oracle/pdae_oracle.c.  No model of the reference calls EMD
// (SURVEY F3); it is on the path because the north star names it.
//
// approxmatch, three forms of the same 10 levels x 3 phases (emd_kernel.cu:43-157):
//   small clouds (n, m <= 64; the 32 x 32 patches of the pretraining step): ONE WAVE runs a whole cloud pair -- two
//     pairs per wave when both clouds have <= 32 points -- with the points and the remain / ratio vectors in LDS as
//     float4 (x, y, z, weight) broadcast reads, every lane's row of `match` in registers for all 10 levels (written
//     once), the reference's sequential summation order per thread;
//   large clouds: each phase is its own launch over the whole chip (a phase's sums range over ALL points of the other
//     cloud, so a cloud split over work-groups meets at a grid-wide boundary 30 times: kernel boundaries, ~2 us each
//     inside one C call / one captured graph): phase 1 / 2: 16 points x 16 slices of the other cloud per block,
//     slice partials added by a fixed shuffle tree; phase 3: 64 points x 4 quarter ranges per block with `match`
//     coalesced on the lanes.  remain / ratio vectors live in `temp` ((n + m) * 2 floats per cloud, the reference's
//     own scratch: emd.cpp:12).  Summation order differs from the reference's by the slicing: within the 2e-3 that the
//     hardware exp (v_exp_f32 vs libm) already costs `match`;
//   the first version (round 1-3, one 1024-thread work-group per cloud pair, a serial loop per thread) is kept for
//     clouds that fit neither (m > 8192 on the large path).
// matchcost / matchcostgrad keep the reference's 512- and 256-way strided
// partial sums and pairwise trees, so given the same `match` they are
// bit-identical to the oracle.
#include "common.h"

namespace pdae {

constexpr int kEmdT = 1024;

__global__ __launch_bounds__(kEmdT) void approxmatch_kernel(int n, int m,
                                                            const float* __restrict__ xyz1_all,
                                                            const float* __restrict__ xyz2_all,
                                                            float* __restrict__ match_all) {
  extern __shared__ float4 lds4[];
  float4* p1 = lds4;      // n: (x,y,z,-)
  float4* p2 = lds4 + n;  // m
  float* remainL = reinterpret_cast<float*>(lds4 + n + m);
  float* remainR = remainL + n;
  float* ratioL = remainR + m;
  float* ratioR = ratioL + n;
  const int i = blockIdx.x;
  const float* xyz1 = xyz1_all + (size_t)i * n * 3;
  const float* xyz2 = xyz2_all + (size_t)i * m * 3;
  float* match = match_all + (size_t)i * n * m;
  float multiL, multiR;
  if (n >= m) {
    multiL = 1;
    multiR = (float)(n / m);
  } else {
    multiL = (float)(m / n);
    multiR = 1;
  }
  // every match element is zeroed and later accumulated by the SAME thread
  // (k on the lanes), so the read-modify-write below needs no cross-thread
  // visibility through global memory
  for (int k = threadIdx.x; k < n; k += kEmdT)
    for (int l = 0; l < m; ++l) match[(size_t)l * n + k] = 0;
  for (int j = threadIdx.x; j < n; j += kEmdT) {
    p1[j] = make_float4(xyz1[j * 3 + 0], xyz1[j * 3 + 1], xyz1[j * 3 + 2], 0.f);
    remainL[j] = multiL;
  }
  for (int j = threadIdx.x; j < m; j += kEmdT) {
    p2[j] = make_float4(xyz2[j * 3 + 0], xyz2[j * 3 + 1], xyz2[j * 3 + 2], 0.f);
    remainR[j] = multiR;
  }
  __syncthreads();
  for (int j = 7; j >= -2; j--) {
    float level = -powf(4.0f, (float)j);
    if (j == -2) level = 0;
    // phase 1 (emd_kernel.cu:51-83): ratioL[k] = remainL[k] / sum_l exp(level d) remainR[l]
    for (int k = threadIdx.x; k < n; k += kEmdT) {
      const float4 a = p1[k];
      float suml = 1e-9f;
      for (int l = 0; l < m; ++l) {
        const float4 q = p2[l];
        const float d = level * sqdist(q.x, q.y, q.z, a.x, a.y, a.z);
        suml += __expf(d) * remainR[l];
      }
      ratioL[k] = remainL[k] / suml;
    }
    __syncthreads();
    // phase 2 (:85-118)
    for (int l = threadIdx.x; l < m; l += kEmdT) {
      const float4 q = p2[l];
      float sumr = 0;
      for (int k = 0; k < n; ++k) {
        const float4 a = p1[k];
        sumr += __expf(level * sqdist(q.x, q.y, q.z, a.x, a.y, a.z)) * ratioL[k];
      }
      const float rr = remainR[l];
      sumr *= rr;
      const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
      ratioR[l] = consumption * rr;
      remainR[l] = fmaxf(0.0f, rr - sumr);
    }
    __syncthreads();
    // phase 3 (:120-155)
    for (int k = threadIdx.x; k < n; k += kEmdT) {
      const float4 a = p1[k];
      const float rl = ratioL[k];
      float suml = 0;
      for (int l = 0; l < m; ++l) {
        const float4 q = p2[l];
        const float w = __expf(level * sqdist(q.x, q.y, q.z, a.x, a.y, a.z)) * rl * ratioR[l];
        match[(size_t)l * n + k] += w;
        suml += w;
      }
      remainL[k] = fmaxf(0.0f, remainL[k] - suml);
    }
    __syncthreads();
  }
}

// ---- small clouds: one wave per cloud pair (two pairs per wave for MP = 32)
template <int MP>
__global__ __launch_bounds__(256) void approxmatch_small_kernel(int b, int n, int m, const float* __restrict__ xyz1_all,
                                                                const float* __restrict__ xyz2_all,
                                                                float* __restrict__ match_all) {
  constexpr int PPW = 64 / MP;
  __shared__ float4 P1s[4][64], P2s[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / MP, idx = lane % MP;
  const int pair = (blockIdx.x * 4 + wave) * PPW + sub;
  const int pc = min(pair, b - 1);                     // pairs past the end: a duplicate, nothing stored
  float4* P1 = &P1s[wave][sub * MP];
  float4* P2 = &P2s[wave][sub * MP];
  const float* x1 = xyz1_all + ((size_t)pc * n + min(idx, n - 1)) * 3;
  const float* x2 = xyz2_all + ((size_t)pc * m + min(idx, m - 1)) * 3;
  const float ax = x1[0], ay = x1[1], az = x1[2], qx = x2[0], qy = x2[1], qz = x2[2];
  float multiL, multiR;
  if (n >= m) multiL = 1, multiR = (float)(n / m);
  else multiL = (float)(m / n), multiR = 1;
  // lanes past a cloud's last point carry weight 0 through every phase (their terms add exact zeros to the sums of the
  // others): the loops below run over all MP slots without a per-slot test
  float remainL = idx < n ? multiL : 0.f, remainR = idx < m ? multiR : 0.f, ratioL = 0.f, ratioR = 0.f;
  float macc[MP];
#pragma unroll
  for (int l = 0; l < MP; ++l) macc[l] = 0.f;
  P1[idx] = make_float4(ax, ay, az, 0.f);
  P2[idx] = make_float4(qx, qy, qz, remainR);
  __syncthreads();
  if constexpr (MP == 32) {
    // 32 x 32: the squared distances of this lane's two points against the other cloud stay in registers for all ten
    // levels, and phase 3 reuses phase 1's exponentials (the reference recomputes the same expf of the same argument):
    // 2 exponentials and ~13 vector instructions per pair and level instead of 3 and ~45.  Same operations, same order.
    float d2[MP], d2t[MP], e[MP];
#pragma unroll
    for (int l = 0; l < MP; ++l) {
      const float4 q = P2[l], a = P1[l];
      d2[l] = sqdist(q.x, q.y, q.z, ax, ay, az);
      d2t[l] = sqdist(qx, qy, qz, a.x, a.y, a.z);
    }
    for (int j = 7; j >= -2; j--) {
      float level = -powf(4.0f, (float)j);
      if (j == -2) level = 0;
      {   // phase 1 (emd_kernel.cu:51-83)
        float suml = 1e-9f;
#pragma unroll
        for (int l = 0; l < MP; ++l) {
          e[l] = __expf(level * d2[l]);
          suml += e[l] * P2[l].w;
        }
        ratioL = remainL / suml;
        P1[idx].w = ratioL;
      }
      __syncthreads();
      {   // phase 2 (:85-118)
        float sumr = 0;
#pragma unroll
        for (int k = 0; k < MP; ++k) sumr += __expf(level * d2t[k]) * P1[k].w;
        sumr *= remainR;
        const float consumption = fminf(remainR / (sumr + 1e-9f), 1.0f);
        ratioR = consumption * remainR;
        remainR = fmaxf(0.0f, remainR - sumr);
        P2[idx].w = ratioR;
      }
      __syncthreads();
      {   // phase 3 (:120-155)
        float suml = 0;
#pragma unroll
        for (int l = 0; l < MP; ++l) {
          const float w = e[l] * ratioL * P2[l].w;
          macc[l] += w;
          suml += w;
        }
        remainL = fmaxf(0.0f, remainL - suml);
      }
      __syncthreads();
      P2[idx].w = remainR;
      __syncthreads();
    }
  } else {
    for (int j = 7; j >= -2; j--) {
      float level = -powf(4.0f, (float)j);
      if (j == -2) level = 0;
      {   // phase 1 (emd_kernel.cu:51-83): this lane's point of cloud 1 against all of cloud 2
        float suml = 1e-9f;
  #pragma unroll
        for (int l = 0; l < MP; ++l) {
          const float4 q = P2[l];
          suml += __expf(level * sqdist(q.x, q.y, q.z, ax, ay, az)) * q.w;
        }
        ratioL = remainL / suml;
        P1[idx].w = ratioL;
      }
      __syncthreads();
      {   // phase 2 (:85-118): this lane's point of cloud 2 against all of cloud 1
        float sumr = 0;
  #pragma unroll
        for (int k = 0; k < MP; ++k) {
          const float4 a = P1[k];
          sumr += __expf(level * sqdist(qx, qy, qz, a.x, a.y, a.z)) * a.w;
        }
        sumr *= remainR;
        const float consumption = fminf(remainR / (sumr + 1e-9f), 1.0f);
        ratioR = consumption * remainR;
        remainR = fmaxf(0.0f, remainR - sumr);
        P2[idx].w = ratioR;
      }
      __syncthreads();
      {   // phase 3 (:120-155)
        float suml = 0;
  #pragma unroll
        for (int l = 0; l < MP; ++l) {
          const float4 q = P2[l];
          const float w = __expf(level * sqdist(q.x, q.y, q.z, ax, ay, az)) * ratioL * q.w;
          macc[l] += w;
          suml += w;
        }
        remainL = fmaxf(0.0f, remainL - suml);
      }
      __syncthreads();
      P2[idx].w = remainR;
      __syncthreads();
    }
  }
  if (pair < b && idx < n) {
    float* match = match_all + (size_t)pair * n * m;
#pragma unroll
    for (int l = 0; l < MP; ++l)
      if (l < m) match[(size_t)l * n + idx] = macc[l];
  }
}

// ---- large clouds: one launch per phase.  temp per cloud: [remainL n][remainR m][ratioL n][ratioR m]
// phase 1 / 2 (SWAP): 16 points of cloud A x 16 slices of cloud B per block; cloud B and its weights staged in LDS
__device__ __forceinline__ float tree16(float v) {    // fixed order over the 16 lanes that share a point
  v += __shfl_xor(v, 8, kWave);
  v += __shfl_xor(v, 4, kWave);
  v += __shfl_xor(v, 2, kWave);
  v += __shfl_xor(v, 1, kWave);
  return v;
}
template <bool PHASE2>
__global__ __launch_bounds__(256) void approxmatch_phase12_kernel(int n, int m, float level, int first, float multiL, float multiR,
                                                                  const float* __restrict__ xyz1_all,
                                                                  const float* __restrict__ xyz2_all,
                                                                  float* __restrict__ temp_all) {
  extern __shared__ float4 emd_p[];                    // the other cloud: (x, y, z, weight)
  const int i = blockIdx.y;
  float* temp = temp_all + (size_t)i * 2 * (n + m);
  float* remainL = temp;
  float* remainR = temp + n;
  float* ratioL = remainR + m;
  float* ratioR = ratioL + n;
  const int na = PHASE2 ? m : n, nb = PHASE2 ? n : m;  // A: the cloud whose points own the sums
  const float* xa = (PHASE2 ? xyz2_all + (size_t)i * m * 3 : xyz1_all + (size_t)i * n * 3);
  const float* xb = (PHASE2 ? xyz1_all + (size_t)i * n * 3 : xyz2_all + (size_t)i * m * 3);
  for (int j = threadIdx.x; j < nb; j += 256) {
    float w;
    if (PHASE2) w = ratioL[j];                         // phase 2 weighs cloud 1 by ratioL
    else w = first ? multiR : remainR[j];              // phase 1 weighs cloud 2 by remainR
    emd_p[j] = make_float4(xb[j * 3 + 0], xb[j * 3 + 1], xb[j * 3 + 2], w);
  }
  __syncthreads();
  const int kq = threadIdx.x >> 4, sl = threadIdx.x & 15;
  const int k = blockIdx.x * 16 + kq, kc = min(k, na - 1);
  const float ax = xa[kc * 3 + 0], ay = xa[kc * 3 + 1], az = xa[kc * 3 + 2];
  float part = 0.f;
  for (int l = sl; l < nb; l += 16) {
    const float4 q = emd_p[l];
    const float d = PHASE2 ? sqdist(ax, ay, az, q.x, q.y, q.z) : sqdist(q.x, q.y, q.z, ax, ay, az);
    part += __expf(level * d) * q.w;
  }
  const float sum = tree16(part);
  if (sl == 0 && k < na) {
    if (!PHASE2) {
      // the remainL update of the previous level's phase 3 (its row sums were parked in ratioL) happens here
      const float rl = first ? multiL : fmaxf(0.0f, remainL[k] - ratioL[k]);
      remainL[k] = rl;
      ratioL[k] = rl / (sum + 1e-9f);
    } else {
      const float rr = first ? multiR : remainR[k];
      const float sumr = sum * rr;
      const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
      ratioR[k] = consumption * rr;
      remainR[k] = fmaxf(0.0f, rr - sumr);
    }
  }
}

// phase 3: 64 points of cloud 1 on the lanes x 16 ranges of cloud 2 on the waves; match (b, m, n) coalesced on the
// lanes.  `match` is read-modify-written once per level: sixteen loads are issued before the first dependent store (a
// load behind every store cost a memory round trip per point of cloud 2: 44 us per launch instead of ~6).
constexpr int kP3W = 16;                               // waves per block
__global__ __launch_bounds__(64 * kP3W) void approxmatch_phase3_kernel(int n, int m, float level, int first,
                                                                       const float* __restrict__ xyz1_all,
                                                                       const float* __restrict__ xyz2_all,
                                                                       float* __restrict__ temp_all, float* __restrict__ match_all) {
  extern __shared__ float4 emd_p[];                    // cloud 2: (x, y, z, ratioR); then kP3W x 64 partial row sums
  const int i = blockIdx.y;
  float* temp = temp_all + (size_t)i * 2 * (n + m);
  float* ratioL = temp + n + m;
  const float* ratioR = ratioL + n;
  const float* xyz1 = xyz1_all + (size_t)i * n * 3;
  const float* xyz2 = xyz2_all + (size_t)i * m * 3;
  float* match = match_all + (size_t)i * n * m;
  for (int j = threadIdx.x; j < m; j += 64 * kP3W)
    emd_p[j] = make_float4(xyz2[j * 3 + 0], xyz2[j * 3 + 1], xyz2[j * 3 + 2], ratioR[j]);
  float* red = reinterpret_cast<float*>(emd_p + m);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane, kc = min(k, n - 1);
  const float ax = xyz1[kc * 3 + 0], ay = xyz1[kc * 3 + 1], az = xyz1[kc * 3 + 2];
  const float rl = ratioL[kc];
  const int mq = (m + kP3W - 1) / kP3W, l0 = wave * mq, l1 = min(m, l0 + mq);
  float suml = 0.f;
  if (k < n) {
    for (int lb = l0; lb < l1; lb += 16) {
      float old[16], w[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int l = min(lb + t, l1 - 1);
        old[t] = first ? 0.f : match[(size_t)l * n + k];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float4 q = emd_p[min(lb + t, l1 - 1)];
        w[t] = __expf(level * sqdist(q.x, q.y, q.z, ax, ay, az)) * rl * q.w;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (lb + t < l1) {
          match[(size_t)(lb + t) * n + k] = old[t] + w[t];
          suml += w[t];
        }
      }
    }
  }
  red[wave * 64 + lane] = suml;
  __syncthreads();
  // the row sum, parked in ratioL[k] for the next level's phase 1 (every reader of ratioL[k] is in this block)
  if (wave == 0 && k < n) {
    float t = red[lane];
#pragma unroll
    for (int w = 1; w < kP3W; ++w) t += red[w * 64 + lane];
    ratioL[k] = t;
  }
}

__global__ __launch_bounds__(512) void matchcost_kernel(int n, int m,
                                                        const float* __restrict__ xyz1_all,
                                                        const float* __restrict__ xyz2_all,
                                                        const float* __restrict__ match_all,
                                                        float* __restrict__ out) {
  __shared__ float allsum[512];
  const int i = blockIdx.x;
  const float* xyz1 = xyz1_all + (size_t)i * n * 3;
  const float* xyz2 = xyz2_all + (size_t)i * m * 3;
  const float* match = match_all + (size_t)i * n * m;
  float subsum = 0;
  for (int k = threadIdx.x; k < n; k += 512) {
    const float x1 = xyz1[k * 3 + 0], y1 = xyz1[k * 3 + 1], z1 = xyz1[k * 3 + 2];
    // sixteen rows of `match` in flight per thread (the chain of additions stays in l order: one load behind every
    // addition kept a single CU at 14 GB/s on a cloud's 4 MB)
    int l = 0;
    for (; l + 16 <= m; l += 16) {
      float mv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) mv[u] = match[(size_t)(l + u) * n + k];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float d = sqdist(xyz2[(l + u) * 3 + 0], xyz2[(l + u) * 3 + 1], xyz2[(l + u) * 3 + 2], x1, y1, z1);
        subsum += d * mv[u];
      }
    }
    for (; l < m; ++l) {
      const float d = sqdist(xyz2[l * 3 + 0], xyz2[l * 3 + 1], xyz2[l * 3 + 2], x1, y1, z1);
      subsum += d * match[(size_t)l * n + k];
    }
  }
  allsum[threadIdx.x] = subsum;
  for (int j = 1; j < 512; j <<= 1) {  // emd_kernel.cu:230-235
    __syncthreads();
    if ((threadIdx.x & (2 * j - 1)) == 0) allsum[threadIdx.x] += allsum[threadIdx.x + j];
  }
  if (threadIdx.x == 0) out[i] = allsum[0];
}

__global__ __launch_bounds__(256) void matchcostgrad1_kernel(int n, int m,
                                                             const float* __restrict__ grad_cost,
                                                             const float* __restrict__ xyz1_all,
                                                             const float* __restrict__ xyz2_all,
                                                             const float* __restrict__ match_all,
                                                             float* __restrict__ grad1_all) {
  const int i = blockIdx.y;
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n) return;
  const float* xyz1 = xyz1_all + (size_t)i * n * 3;
  const float* xyz2 = xyz2_all + (size_t)i * m * 3;
  const float* match = match_all + (size_t)i * n * m;
  const float x1 = xyz1[l * 3 + 0], y1 = xyz1[l * 3 + 1], z1 = xyz1[l * 3 + 2];
  float dx = 0, dy = 0, dz = 0;
  int k = 0;
  for (; k + 16 <= m; k += 16) {                       // sixteen loads in flight; the sums stay in k order
    float mv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) mv[u] = match[(size_t)(k + u) * n + l];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float d = mv[u] * 2;
      dx += (x1 - xyz2[(k + u) * 3 + 0]) * d;
      dy += (y1 - xyz2[(k + u) * 3 + 1]) * d;
      dz += (z1 - xyz2[(k + u) * 3 + 2]) * d;
    }
  }
  for (; k < m; ++k) {
    const float d = match[(size_t)k * n + l] * 2;
    dx += (x1 - xyz2[k * 3 + 0]) * d;
    dy += (y1 - xyz2[k * 3 + 1]) * d;
    dz += (z1 - xyz2[k * 3 + 2]) * d;
  }
  const float gc = grad_cost[i];
  float* g = grad1_all + ((size_t)i * n + l) * 3;
  g[0] = dx * gc;
  g[1] = dy * gc;
  g[2] = dz * gc;
}

// one workgroup per (xyz2 point k, cloud): 256 strided partials + the tree
__global__ __launch_bounds__(256) void matchcostgrad2_kernel(int n, int m,
                                                             const float* __restrict__ grad_cost,
                                                             const float* __restrict__ xyz1_all,
                                                             const float* __restrict__ xyz2_all,
                                                             const float* __restrict__ match_all,
                                                             float* __restrict__ grad2_all) {
  __shared__ float sum_grad[256 * 3];
  const int k = blockIdx.x, i = blockIdx.y;
  const float* xyz1 = xyz1_all + (size_t)i * n * 3;
  const float* xyz2 = xyz2_all + (size_t)i * m * 3;
  const float* match = match_all + (size_t)i * n * m + (size_t)k * n;
  const float x2 = xyz2[k * 3 + 0], y2 = xyz2[k * 3 + 1], z2 = xyz2[k * 3 + 2];
  float sx = 0, sy = 0, sz = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float d = match[j] * 2;
    sx += (x2 - xyz1[j * 3 + 0]) * d;
    sy += (y2 - xyz1[j * 3 + 1]) * d;
    sz += (z2 - xyz1[j * 3 + 2]) * d;
  }
  sum_grad[threadIdx.x * 3 + 0] = sx;
  sum_grad[threadIdx.x * 3 + 1] = sy;
  sum_grad[threadIdx.x * 3 + 2] = sz;
  for (int j = 1; j < 256; j <<= 1) {
    __syncthreads();
    if ((threadIdx.x & (2 * j - 1)) == 0) {
      sum_grad[threadIdx.x * 3 + 0] += sum_grad[(threadIdx.x + j) * 3 + 0];
      sum_grad[threadIdx.x * 3 + 1] += sum_grad[(threadIdx.x + j) * 3 + 1];
      sum_grad[threadIdx.x * 3 + 2] += sum_grad[(threadIdx.x + j) * 3 + 2];
    }
  }
  if (threadIdx.x == 0) {
    const float gc = grad_cost[i];
    float* g = grad2_all + ((size_t)i * m + k) * 3;
    g[0] = sum_grad[0] * gc;
    g[1] = sum_grad[1] * gc;
    g[2] = sum_grad[2] * gc;
  }
}

// the same for clouds of at most 64 points: ONE WAVE per (point of cloud 2, cloud).  The 256-way strided partials are
// then single terms in lanes 0 .. n - 1 (zeros elsewhere) and the pairwise tree is six xor-shuffles -- the same operands
// in the same order, bit-identical to the kernel above, without its 8 barriers and 168 k work-groups of 32 busy threads
__global__ __launch_bounds__(256) void matchcostgrad2_small_kernel(int n, int m, const float* __restrict__ grad_cost,
                                                                   const float* __restrict__ xyz1_all,
                                                                   const float* __restrict__ xyz2_all,
                                                                   const float* __restrict__ match_all,
                                                                   float* __restrict__ grad2_all) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6), i = blockIdx.y;
  if (k >= m) return;
  const float* xyz1 = xyz1_all + (size_t)i * n * 3;
  const float* xyz2 = xyz2_all + (size_t)i * m * 3;
  const float* match = match_all + (size_t)i * n * m + (size_t)k * n;
  const float x2 = xyz2[k * 3 + 0], y2 = xyz2[k * 3 + 1], z2 = xyz2[k * 3 + 2];
  float sx = 0, sy = 0, sz = 0;
  if (lane < n) {
    const float d = match[lane] * 2;
    sx += (x2 - xyz1[lane * 3 + 0]) * d;
    sy += (y2 - xyz1[lane * 3 + 1]) * d;
    sz += (z2 - xyz1[lane * 3 + 2]) * d;
  }
#pragma unroll
  for (int j = 1; j < 64; j <<= 1) {
    sx += __shfl_xor(sx, j, kWave);
    sy += __shfl_xor(sy, j, kWave);
    sz += __shfl_xor(sz, j, kWave);
  }
  if (lane == 0) {
    const float gc = grad_cost[i];
    float* g = grad2_all + ((size_t)i * m + k) * 3;
    g[0] = sx * gc;
    g[1] = sy * gc;
    g[2] = sz * gc;
  }
}

static int emd_check(int b, int n, int m) {
  if (b < 0 || n <= 0 || m <= 0) return bad_arg("emd: b>=0, n>0, m>0 required");
  if (b > 65535 || m > 65535) return unsupported("emd: b or m > 65535");
  return PDAE_OK;
}

}  // namespace pdae

extern "C" int pdae_emd_approxmatch(int b, int n, int m, const float* xyz1, const float* xyz2,
                                    float* match, float* temp, pdae_stream_t stream) {
  using namespace pdae;
  int rc = emd_check(b, n, m);
  if (rc) return rc;
  if (b == 0) return PDAE_OK;
  if (!xyz1 || !xyz2 || !match) return bad_arg("emd_approxmatch: null pointer");
  hipStream_t s = as_stream(stream);
  if (n <= 64 && m <= 64) {                              // one wave per cloud pair
    if (n <= 32 && m <= 32) hipLaunchKernelGGL(approxmatch_small_kernel<32>, dim3((b + 7) / 8), dim3(256), 0, s, b, n, m, xyz1, xyz2, match);
    else hipLaunchKernelGGL(approxmatch_small_kernel<64>, dim3((b + 3) / 4), dim3(256), 0, s, b, n, m, xyz1, xyz2, match);
    return check_launch("emd_approxmatch");
  }
  if (temp && n <= 8192 && m <= 8192) {                  // one launch per phase over the whole chip
    float multiL, multiR;
    if (n >= m) multiL = 1, multiR = (float)(n / m);
    else multiL = (float)(m / n), multiR = 1;
    static bool once = false;
    if (!once) {
      once = true;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(approxmatch_phase12_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(approxmatch_phase12_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(approxmatch_phase3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16 + kP3W * 256);
    }
    for (int j = 7; j >= -2; j--) {
      float level = -powf(4.0f, (float)j);
      if (j == -2) level = 0;
      const int first = j == 7;
      hipLaunchKernelGGL(approxmatch_phase12_kernel<false>, dim3((n + 15) / 16, b), dim3(256), (size_t)m * 16, s, n, m, level, first,
                         multiL, multiR, xyz1, xyz2, temp);
      hipLaunchKernelGGL(approxmatch_phase12_kernel<true>, dim3((m + 15) / 16, b), dim3(256), (size_t)n * 16, s, n, m, level, first,
                         multiL, multiR, xyz1, xyz2, temp);
      hipLaunchKernelGGL(approxmatch_phase3_kernel, dim3((n + 63) / 64, b), dim3(64 * kP3W), (size_t)m * 16 + kP3W * 256, s, n, m, level,
                         first, xyz1, xyz2, temp, match);
    }
    return check_launch("emd_approxmatch");
  }
  // no scratch (or a cloud past the staged size): one work-group per cloud pair, everything in its LDS
  const size_t lds = (size_t)(n + m) * (sizeof(float4) + 2 * sizeof(float));
  if (lds > 160 * 1024) return unsupported("emd_approxmatch: n + m > 6826 without `temp`, or a cloud of more than 8192 points");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(approxmatch_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(approxmatch_kernel, dim3(b), dim3(kEmdT), lds, s, n, m, xyz1,
                     xyz2, match);
  return check_launch("emd_approxmatch");
}

extern "C" int pdae_emd_matchcost(int b, int n, int m, const float* xyz1, const float* xyz2,
                                  const float* match, float* cost, pdae_stream_t stream) {
  using namespace pdae;
  int rc = emd_check(b, n, m);
  if (rc) return rc;
  if (b == 0) return PDAE_OK;
  if (!xyz1 || !xyz2 || !match || !cost) return bad_arg("emd_matchcost: null pointer");
  hipLaunchKernelGGL(matchcost_kernel, dim3(b), dim3(512), 0, as_stream(stream), n, m, xyz1, xyz2,
                     match, cost);
  return check_launch("emd_matchcost");
}

extern "C" int pdae_emd_matchcost_grad(int b, int n, int m, const float* grad_cost,
                                       const float* xyz1, const float* xyz2, const float* match,
                                       float* grad1, float* grad2, pdae_stream_t stream) {
  using namespace pdae;
  int rc = emd_check(b, n, m);
  if (rc) return rc;
  if (b == 0) return PDAE_OK;
  if (!grad_cost || !xyz1 || !xyz2 || !match || !grad1 || !grad2)
    return bad_arg("emd_matchcost_grad: null pointer");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(matchcostgrad1_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, n, m,
                     grad_cost, xyz1, xyz2, match, grad1);
  if (n <= 64) hipLaunchKernelGGL(matchcostgrad2_small_kernel, dim3((m + 3) / 4, b), dim3(256), 0, s, n, m, grad_cost, xyz1, xyz2, match, grad2);
  else hipLaunchKernelGGL(matchcostgrad2_kernel, dim3(m, b), dim3(256), 0, s, n, m, grad_cost, xyz1,
                          xyz2, match, grad2);
  return check_launch("emd_matchcost_grad");
}
