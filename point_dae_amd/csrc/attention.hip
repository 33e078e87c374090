// attention.hip -- fused multi-head attention core for short sequences (T <= 128,
// head dim 64), forward and backward, on the fp32 matrix cores.
//
// Semantics: Attention.forward of the reference
// (models/PointCAE_transformer.py:125-137) between the qkv and proj Linears:
//     q,k,v = qkv.reshape(B,T,3,H,64);  o = softmax(q k^T * scale) v
// The reference materialises the (B,H,T,T) scores, runs two batched GEMMs, a
// softmax kernel and several permute copies (and their backward twins).  Here one
// workgroup per (sample, head) keeps Q, K, V in LDS and the T x T score tile in
// MFMA accumulators; nothing but o (and one log-sum-exp per row, for the
// backward) goes back to HBM.  Sequences are the 13..64 (cfg5: 128) patch tokens
// of a cloud, so the whole score tile fits in registers: no online softmax.
//
// Layout trick (guide: "an accumulator tile as the next MFMA's operand"): with
// v_mfma_f32_32x32x2_f32 an accumulator tile X has its column on the lane and
// its rows in the 16 registers, so a product that sums over X's ROW index,
// X^T . B, takes the registers directly as the A operand (register e of lane
// half h is row (e&3) + 8(e>>2) + 4h; B is read from LDS at that row).
//   forward : X = S^T = K Q^T (rows = keys, lane = query): softmax over keys is
//             in-lane (+ one shuffle across the halves), then O = X^T . V.
//   backward: dQ = dS . K needs X = dS^T (rows = keys);  dV = P^T dO and
//             dK = dS^T Q need X = P, dS (rows = queries).  Both orientations
//             are recomputed from Q, K, V, dO and the saved log-sum-exp -- four
//             64-deep products -- instead of transposing through LDS.
// One wave per 32-row tile of queries (keys in the backward's second half).
#include <cstdlib>

#include "common.h"

namespace pdae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AD = 64;        // head dim
constexpr int ALD = AD + 4;   // padded LDS row (floats): conflict-free ds_read_b128
constexpr int AMAXT = 128;

__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

__device__ __forceinline__ void zero16(f32x16& v) {
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = 0.f;
}

// acc += A_tile(32 x 64) . B_tile(32 x 64)^T with both tiles K-contiguous in LDS
// (row stride ALD): acc[col = B row on lane][row = A row in regs].
__device__ __forceinline__ void mma_nt_64(f32x16& acc, const float* As, const float* Bs, int r, int h) {
#pragma unroll
  for (int s = 0; s < AD / 8; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(As + r * ALD + s * 8 + 4 * h);
    const float4 b = *reinterpret_cast<const float4*>(Bs + r * ALD + s * 8 + 4 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}

// acc(32 x 32 block of columns d0..d0+31) += X^T . Bt where X is an accumulator
// tile (rows = the summed index) and Bt rows are the same summed index in LDS.
__device__ __forceinline__ void mma_xt_b(f32x16& acc, const f32x16& X, const float* Bt, int d0,
                                         int r, int h) {
#pragma unroll
  for (int e = 0; e < 16; ++e)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[e], Bt[acc_row(e, h) * ALD + d0 + r], acc, 0, 0, 0);
}

// stage rows [0,T) of one head's q/k/v (row stride 3*H*64 floats in global) into
// a padded LDS tile; rows T..Tpad-1 are zero
__device__ __forceinline__ void stage_head(float* dst, const float* src, int T, int Tpad,
                                           size_t row_stride, int tid, int nthreads) {
  for (int i = tid; i < Tpad * (AD / 4); i += nthreads) {
    const int row = i / (AD / 4), c4 = (i % (AD / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < T) v = *reinterpret_cast<const float4*>(src + (size_t)row * row_stride + c4);
    *reinterpret_cast<float4*>(dst + row * ALD + c4) = v;
  }
}

// The same staging with every global load of the thread issued BEFORE the first LDS store
// (NJ float4 per tensor and thread, Tpad * 16 == NJ * nthreads): the loop above waits out a
// full memory latency per float4 -- 24 of them in a row made up most of the forward's time.
template <int NJ>
__device__ __forceinline__ void load_head(float4 (&buf)[NJ], const float* src, int T, size_t row_stride,
                                          int tid, int nthreads) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + j * nthreads;
    const int row = i >> 4, c4 = (i & 15) * 4;
    buf[j] = row < T ? *reinterpret_cast<const float4*>(src + (size_t)row * row_stride + c4)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NJ>
__device__ __forceinline__ void store_head(float* dst, const float4 (&buf)[NJ], int tid, int nthreads) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + j * nthreads;
    *reinterpret_cast<float4*>(dst + (i >> 4) * ALD + (i & 15) * 4) = buf[j];
  }
}

// ---------------------------------------------------------------- forward
// grid (H, B), block = 64 * ceil(T/32) threads.  qkv (B*T, 3*H*64); o (B*T, H*64);
// lse (B, H, T).
__global__ __launch_bounds__(256) void attention_fwd_kernel(int T, int H, float scale,
                                                            const float* __restrict__ qkv,
                                                            float* __restrict__ o,
                                                            float* __restrict__ lse) {
  extern __shared__ float lds[];
  const int NW = blockDim.x >> 6;  // query tiles == key tiles
  const int Tpad = NW * 32;
  float* Qs = lds;
  float* Ks = Qs + Tpad * ALD;
  float* Vs = Ks + Tpad * ALD;
  const int hd = blockIdx.x, b = blockIdx.y;
  const size_t rs = (size_t)3 * H * AD;
  const float* base = qkv + (size_t)b * T * rs + hd * AD;
  {   // blockDim = 2 * Tpad threads: 8 float4 per tensor and thread, 24 loads in flight
    float4 bq[8], bk[8], bv[8];
    load_head(bq, base, T, rs, threadIdx.x, blockDim.x);
    load_head(bk, base + (size_t)H * AD, T, rs, threadIdx.x, blockDim.x);
    load_head(bv, base + (size_t)2 * H * AD, T, rs, threadIdx.x, blockDim.x);
    store_head(Qs, bq, threadIdx.x, blockDim.x);
    store_head(Ks, bk, threadIdx.x, blockDim.x);
    store_head(Vs, bv, threadIdx.x, blockDim.x);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int qi = w * 32 + r;  // this lane's query (columns of S^T)

  // S^T tiles: rows = keys of tile jt, lane = query
  f32x16 st[AMAXT / 32];
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int jt = 0; jt < AMAXT / 32; ++jt) {
    if (jt < NW) {
      zero16(st[jt]);
      mma_nt_64(st[jt], Ks + jt * 32 * ALD, Qs + w * 32 * ALD, r, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = jt * 32 + acc_row(e, h);
        st[jt][e] = key < T ? st[jt][e] * scale : -__builtin_huge_valf();
        m = fmaxf(m, st[jt][e]);
      }
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, kWave));
  float l = 0.f;
#pragma unroll
  for (int jt = 0; jt < AMAXT / 32; ++jt) {
    if (jt < NW) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = expf(st[jt][e] - m);
        st[jt][e] = p;
        l += p;
      }
    }
  }
  l += __shfl_xor(l, 32, kWave);
  const float inv = 1.0f / l;
  if (h == 0 && qi < T) lse[((size_t)b * H + hd) * T + qi] = m + logf(l);

  // O = P V: two 32-column halves of the head dim
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16 oa;
    zero16(oa);
#pragma unroll
    for (int jt = 0; jt < AMAXT / 32; ++jt)
      if (jt < NW) mma_xt_b(oa, st[jt], Vs + jt * 32 * ALD, dt * 32, r, h);
    // oa: lane = d column, regs = query rows of tile w.  NOTE: the softmax
    // denominator belongs to the QUERY, which sits on the rows here, so it is
    // fetched from the lane that owns that query.
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qrow = acc_row(e, h);
      const float iv = __shfl(inv, qrow, kWave);  // lane qrow (h = 0 half) holds query qrow's 1/l
      const int q = w * 32 + qrow;
      if (q < T) store_wt(&o[((size_t)b * T + q) * H * AD + hd * AD + dt * 32 + r], oa[e] * iv);
    }
  }
}

// ---------------------------------------------------------------- backward
// dqkv (B*T, 3*H*64) fully written.  grid (H, B), block = 2 * 64 * ceil(T/32): the
// first half of the waves computes dQ (one wave per query tile), the second half
// dK and dV (one wave per key tile), concurrently.
__global__ __launch_bounds__(512) void attention_bwd_kernel(int T, int H, float scale,
                                                            const float* __restrict__ qkv,
                                                            const float* __restrict__ o,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ d_o,
                                                            float* __restrict__ dqkv) {
  extern __shared__ float lds[];
  const int NW = blockDim.x >> 7;
  const int Tpad = NW * 32;
  float* Qs = lds;
  float* Ks = Qs + Tpad * ALD;
  float* Vs = Ks + Tpad * ALD;
  float* dOs = Vs + Tpad * ALD;
  float* Ls = dOs + Tpad * ALD;   // [Tpad] log-sum-exp
  float* Ds = Ls + Tpad;          // [Tpad] delta = rowsum(dO * O)
  const int hd = blockIdx.x, b = blockIdx.y;
  const size_t rs = (size_t)3 * H * AD, os = (size_t)H * AD;
  const float* base = qkv + (size_t)b * T * rs + hd * AD;
  {   // blockDim = 4 * Tpad threads: 4 float4 per tensor and thread, 20 loads in flight
    float4 bq[4], bk[4], bv[4], bd[4], bo[4];
    load_head(bq, base, T, rs, threadIdx.x, blockDim.x);
    load_head(bk, base + (size_t)H * AD, T, rs, threadIdx.x, blockDim.x);
    load_head(bv, base + (size_t)2 * H * AD, T, rs, threadIdx.x, blockDim.x);
    load_head(bd, d_o + (size_t)b * T * os + hd * AD, T, os, threadIdx.x, blockDim.x);
    load_head(bo, o + (size_t)b * T * os + hd * AD, T, os, threadIdx.x, blockDim.x);
    const float ls = (int)threadIdx.x < T ? lse[((size_t)b * H + hd) * T + threadIdx.x] : 0.f;
    store_head(Qs, bq, threadIdx.x, blockDim.x);
    store_head(Ks, bk, threadIdx.x, blockDim.x);
    store_head(Vs, bv, threadIdx.x, blockDim.x);
    store_head(dOs, bd, threadIdx.x, blockDim.x);
    if ((int)threadIdx.x < Tpad) Ls[threadIdx.x] = ls;
    // delta[q] = sum_d dO[q][d] * O[q][d]: the 16 consecutive lanes that staged row q hold it
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float dl = (bd[j].x * bo[j].x + bd[j].y * bo[j].y) + (bd[j].z * bo[j].z + bd[j].w * bo[j].w);
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) dl += __shfl_xor(dl, m, kWave);
      const int i = threadIdx.x + j * blockDim.x;
      if ((i & 15) == 0) Ds[i >> 4] = dl;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int part = (threadIdx.x >> 6) / NW, w = (threadIdx.x >> 6) % NW;
  const int r = lane & 31, h = lane >> 5;

  // ---- part 1: dQ for query tile w.  X = dS^T (rows = keys, lane = query w*32+r)
  if (part == 0) {
    const int q = w * 32 + r;
    const float lq = Ls[q], dq_delta = Ds[q];
    f32x16 dqa[2];
    zero16(dqa[0]);
    zero16(dqa[1]);
#pragma unroll
    for (int jt = 0; jt < AMAXT / 32; ++jt) {
      if (jt < NW) {
        f32x16 s, dp;
        zero16(s);
        zero16(dp);
        mma_nt_64(s, Ks + jt * 32 * ALD, Qs + w * 32 * ALD, r, h);     // S^T[key][query]
        mma_nt_64(dp, Vs + jt * 32 * ALD, dOs + w * 32 * ALD, r, h);   // dP^T[key][query]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = jt * 32 + acc_row(e, h);
          const float p = (key < T && q < T) ? expf(s[e] * scale - lq) : 0.f;
          s[e] = p * (dp[e] - dq_delta) * scale;   // dS^T, scale folded in
        }
        mma_xt_b(dqa[0], s, Ks + jt * 32 * ALD, 0, r, h);
        mma_xt_b(dqa[1], s, Ks + jt * 32 * ALD, 32, r, h);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qq = w * 32 + acc_row(e, h);
        if (qq < T) store_wt(&dqkv[((size_t)b * T + qq) * rs + hd * AD + dt * 32 + r], dqa[dt][e]);
      }
  }
  // ---- part 2: dK, dV for key tile w.  X = P, dS (rows = queries, lane = key w*32+r)
  if (part == 1) {
    const int key = w * 32 + r;
    f32x16 dka[2], dva[2];
    zero16(dka[0]);
    zero16(dka[1]);
    zero16(dva[0]);
    zero16(dva[1]);
#pragma unroll
    for (int it = 0; it < AMAXT / 32; ++it) {
      if (it < NW) {
        f32x16 s, dp;
        zero16(s);
        zero16(dp);
        mma_nt_64(s, Qs + it * 32 * ALD, Ks + w * 32 * ALD, r, h);     // S[query][key]
        mma_nt_64(dp, dOs + it * 32 * ALD, Vs + w * 32 * ALD, r, h);   // dP[query][key]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int q = it * 32 + acc_row(e, h);
          const float p = (key < T && q < T) ? expf(s[e] * scale - Ls[q]) : 0.f;
          dp[e] = p * (dp[e] - Ds[q]) * scale;     // dS
          s[e] = p;                                 // P
        }
        mma_xt_b(dva[0], s, dOs + it * 32 * ALD, 0, r, h);
        mma_xt_b(dva[1], s, dOs + it * 32 * ALD, 32, r, h);
        mma_xt_b(dka[0], dp, Qs + it * 32 * ALD, 0, r, h);
        mma_xt_b(dka[1], dp, Qs + it * 32 * ALD, 32, r, h);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = w * 32 + acc_row(e, h);
        if (kk < T) {
          float* row = dqkv + ((size_t)b * T + kk) * rs + hd * AD + dt * 32 + r;
          store_wt(row + (size_t)H * AD, dka[dt][e]);
          store_wt(row + (size_t)2 * H * AD, dva[dt][e]);
        }
      }
  }
}

}  // namespace pdae

namespace pdae {
int gemm_arith_rows();     // rows_gemm.hip: the arithmetic in force for the dense layers
// attention3.hip: the same contract on the exact-split bf16 pipe, for 32 < T <= 64
int attention3_forward(int B, int T, int H, float scale, const float* qkv, float* o, float* lse, hipStream_t s);
int attention3_backward(int B, int T, int H, float scale, const float* qkv, const float* o, const float* lse, const float* d_o,
                        float* dqkv, hipStream_t s);
}  // namespace pdae

using namespace pdae;

// The exact-split kernels take the two-tile sequences when the library's GEMM arithmetic is the exact-split one
// (pdae_set_gemm_arith / PDAE_GEMM).  Measured at the decoder's size (B = 128, T = 64, H = 6; tools/lab/attn_time.py): the
// forward 16.0 us against the fp32-input kernel's 19.9: it ships.  The backward 49.7 (two tensors of planes per block,
// parts as separate blocks) / 46.8 (all four, one block) against 46.2: its blocks are one serial chain -- loads, ~1000
// split instructions, 190 MFMAs with their fragment reads, two exponential sweeps, stores: ~10 us per block of which the
// matrix pipe is 2.6 -- at 424 registers, one wave per SIMD: nothing overlaps it.  It stays in the library behind
// PDAE_ATTN=b3 (lab); PDAE_ATTN=f32 keeps the fp32-input kernels everywhere (A/B runs).
static bool attn3_takes(int T, bool backward) {
  static const int mode = [] {
    const char* e = getenv("PDAE_ATTN");
    return !e ? 0 : (e[0] == 'f' ? 1 : (e[0] == 'b' ? 2 : 0));
  }();
  if (mode == 1 || (backward && mode != 2)) return false;
  return T > 32 && T <= 64 && gemm_arith_rows() == PDAE_GEMM_BF16X3;
}

static int attn_check(int B, int T, int H, int D) {
  if (B < 0 || T <= 0 || H <= 0) return bad_arg("attention: bad size");
  if (D != AD) return unsupported("attention: head dim must be 64");
  if (T > AMAXT) return unsupported("attention: T > 128 not implemented");
  if (B > 65535 || H > 65535) return unsupported("attention: B or H > 65535");
  return PDAE_OK;
}

extern "C" int pdae_attention_forward(int B, int T, int H, int D, float scale, const float* qkv,
                                      float* o, float* lse, pdae_stream_t stream) {
  int rc = attn_check(B, T, H, D);
  if (rc) return rc;
  if (B == 0) return PDAE_OK;
  if (!qkv || !o || !lse) return bad_arg("attention_forward: null pointer");
  if (attn3_takes(T, false)) return attention3_forward(B, T, H, scale, qkv, o, lse, as_stream(stream));
  const int NW = (T + 31) / 32;
  const size_t lds = (size_t)3 * NW * 32 * ALD * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AMAXT * ALD * 4);
    once = true;
  }
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(H, B), dim3(64 * NW), lds, as_stream(stream), T, H,
                     scale, qkv, o, lse);
  return check_launch("attention_forward");
}

extern "C" int pdae_attention_backward(int B, int T, int H, int D, float scale, const float* qkv,
                                       const float* o, const float* lse, const float* d_o,
                                       float* dqkv, pdae_stream_t stream) {
  int rc = attn_check(B, T, H, D);
  if (rc) return rc;
  if (B == 0) return PDAE_OK;
  if (!qkv || !o || !lse || !d_o || !dqkv) return bad_arg("attention_backward: null pointer");
  if (attn3_takes(T, true)) return attention3_backward(B, T, H, scale, qkv, o, lse, d_o, dqkv, as_stream(stream));
  const int NW = (T + 31) / 32;
  const size_t lds = ((size_t)4 * NW * 32 * ALD + 2 * NW * 32) * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (4 * AMAXT * ALD + 2 * AMAXT) * 4);
    once = true;
  }
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(H, B), dim3(128 * NW), lds, as_stream(stream), T, H,
                     scale, qkv, o, lse, d_o, dqkv);
  return check_launch("attention_backward");
}
