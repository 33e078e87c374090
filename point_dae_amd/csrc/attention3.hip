// attention3.hip -- the fused attention core (attention.hip: same contract, same block / wave decomposition) with its five
// products per (sample, head) on the bf16 matrix pipe in the row GEMMs' exact-split arithmetic (round 6): every fp32
// operand -- q, k, v, dO as they are staged, the probabilities P and the score gradients dS as they sit in the
// accumulators -- is split exactly into three bf16 terms, a product is the six largest of the nine exact partial
// products (rows3_kernel.h), hh in one fp32 accumulator and the five small terms in a second one.  For the sequences
// where the fp32-input matrix pipe was the bound: 32 < T <= 64 (the decoder's 64 tokens: backward 49 us, fp32-MFMA-bound
// at 256 v_mfma_f32_32x32x2_f32 x 64 cycles per key-tile wave).  T <= 32 (one wave per head, launch- and latency-bound)
// and T > 64 (the planes of four tensors do not fit LDS) stay on attention.hip.
//
// LDS: per tensor three planes [Tpad rows][64 bf16] = 128-byte rows, ONE image for both kinds of read
// (cdna_hip_programming.md T10): byte(row, 16-byte chunk c) = 128 row + 16 (c ^ f(row)), f = (b1 << 2) | (b3 << 1) | b2 of
// the row's bits.  Row reads (the k-contiguous operands of S = K Q^T, dP = V dO^T: ds_read_b128, lane = row): the eight
// same-parity rows of a 16-lane group take eight different f => 16 different 16-byte slots.  Transposed reads (the
// operands summed over their ROW index: V in O = P V, K in dQ, dO in dV, Q in dK: ds_read_b64_tr_b16, four consecutive
// rows per 16-lane group): row bit 0 picks the 128-byte half, row bit 1 flips chunk bit 2 = the 64-byte window, the lanes'
// own 8 bytes fill the window: 32 lanes, 64 banks, once.
//
// An accumulator tile as the next product's operand (guide, "An accumulator tile as the next MFMA's operand"): registers
// 8 s .. 8 s + 7 of a 32 x 32 fp32 tile, split and packed, ARE the A fragment of 16-deep step s of X^T . B; its element j
// of lane half h stands for row 16 s + 8 (j >> 2) + 4 h + (j & 3) of X, which is the order two transposed reads (rows
// 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3) hand the B operand in.
#include "common.h"

namespace pdae {
namespace attn3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int AD = 64;

__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

__device__ __forceinline__ void zero16(f32x16& v) {
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = 0.f;
}

__device__ __forceinline__ int img(int row, int c) {
  const int f = (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
  return row * 128 + ((c ^ f) << 4);
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  f32x2 x = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}

// eight fp32 -> three planes of eight bf16, x = h + m + l exactly (rows3_kernel.h split_chunk, all chunks)
__device__ __forceinline__ void split8(float (&v)[8], bf16x8 (&out)[3]) {
  u32x4 pk[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned t = cvt_pk(v[2 * q], v[2 * q + 1]);
    v[2 * q] = v[2 * q] - __uint_as_float(t << 16);
    v[2 * q + 1] = v[2 * q + 1] - __uint_as_float(t & 0xffff0000u);
    pk[0][q] = t;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned t = cvt_pk(v[2 * q], v[2 * q + 1]);
    v[2 * q] = v[2 * q] - __uint_as_float(t << 16);
    v[2 * q + 1] = v[2 * q + 1] - __uint_as_float(t & 0xffff0000u);
    pk[1][q] = t;
    pk[2][q] = cvt_pk(v[2 * q], v[2 * q + 1]);
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) out[pl] = __builtin_bit_cast(bf16x8, pk[pl]);
}

// the six products of one 16-deep step: the five small terms into lo, hh into hi (planes 0 = h, 1 = m, 2 = l)
__device__ __forceinline__ void mma6(f32x16& hi, f32x16& lo, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
}

template <int PLANE>
struct Tens {
  const char* base;
  // k-contiguous fragment of rows 32 t .. 32 t + 31, 16-deep step s of the 64-wide head dim
  __device__ __forceinline__ void row(bf16x8 (&f)[3], int t, int s, int r, int h) const {
    const char* p = base + img(32 * t + r, 2 * s + h);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
  }
  // fragment summed over the ROW index: rows 32 t + 16 s .. + 15 (in the accumulator's register order), columns 32 dt ..
  __device__ __forceinline__ void tr(bf16x8 (&f)[3], int t, int s, int dt, int lane) const {
    const int h = lane >> 5, cb = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 32 * t + 16 * s + 4 * h + q, c = 4 * dt + 2 * cb + (pp >> 1), sub = 8 * (pp & 1);
    const char* p0 = base + img(r0, c) + sub;
    const char* p1 = base + img(r0 + 8, c) + sub;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0 + pl * PLANE));
      const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1 + pl * PLANE));
      f[pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  }
};

// registers 8 s .. 8 s + 7 of an accumulator tile as the A fragments (three planes) of step s
__device__ __forceinline__ void acc_frag(bf16x8 (&f)[3], const f32x16& x, int s) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x[8 * s + j];
  split8(v, f);
}

// stage rows [0, T) of one head (row stride `rs` floats in global) as three planes; rows T .. Tpad - 1 are zero.  NJ
// octets per thread, every global load issued before the first split.  With `other` (the backward's dO beside O): also
// returns, per octet, the dot product of the two tensors' octets (delta = rowsum(dO * O)).
template <int NJ, int PLANE>
__device__ __forceinline__ void stage3(char* dst, const float* src, int T, size_t rs, int tid, int nthreads,
                                       const float* other = nullptr, float* dots = nullptr) {
  float4 a[NJ][2], b[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + j * nthreads, row = i >> 3, c = i & 7;
    const bool in = row < T;
    const float* p = src + (size_t)row * rs + c * 8;
    a[j][0] = in ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    a[j][1] = in ? *reinterpret_cast<const float4*>(p + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (other) {
      const float* q = other + (size_t)row * rs + c * 8;
      b[j][0] = in ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[j][1] = in ? *reinterpret_cast<const float4*>(q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + j * nthreads, row = i >> 3, c = i & 7;
    float v[8] = {a[j][0].x, a[j][0].y, a[j][0].z, a[j][0].w, a[j][1].x, a[j][1].y, a[j][1].z, a[j][1].w};
    if (other) {
      dots[j] = ((v[0] * b[j][0].x + v[1] * b[j][0].y) + (v[2] * b[j][0].z + v[3] * b[j][0].w)) +
                ((v[4] * b[j][1].x + v[5] * b[j][1].y) + (v[6] * b[j][1].z + v[7] * b[j][1].w));
    }
    bf16x8 f[3];
    split8(v, f);
    char* d = dst + img(row, c);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(d + pl * PLANE) = f[pl];
  }
}

// A wave's OWN 32-row tile as k-contiguous fragments straight from global memory (no LDS: nobody else reads it): lane
// (r, h) holds row r, head-dim elements 16 s + 8 h .. + 7 of step s.  rows >= T read as zeros.
__device__ __forceinline__ void own_rows(float (&v)[4][8], const float* src, int row, bool in, size_t rs, int h) {
  const float* p = src + (size_t)row * rs + 8 * h;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 a = in ? *reinterpret_cast<const float4*>(p + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = in ? *reinterpret_cast<const float4*>(p + 16 * s + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[s][0] = a.x, v[s][1] = a.y, v[s][2] = a.z, v[s][3] = a.w, v[s][4] = b.x, v[s][5] = b.y, v[s][6] = b.z, v[s][7] = b.w;
  }
}

// ---------------------------------------------------------------- forward
// grid (H, B), block = 64 NW threads (NW = ceil(T / 32) query tiles == key tiles).  LDS: K and V (a wave's own Q tile
// goes global -> registers): 48 KB at NW = 2, three blocks per CU.
template <int NW>
__global__ __launch_bounds__(64 * NW) void attention3_fwd_kernel(int T, int H, float scale, const float* __restrict__ qkv,
                                                                 float* __restrict__ o, float* __restrict__ lse) {
  constexpr int Tpad = 32 * NW, PLANE = Tpad * 128, TENS = 3 * PLANE, NT = 64 * NW, NJ = Tpad * 8 / NT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const int qi = w * 32 + r;
  const size_t rs = (size_t)3 * H * AD;
  const float* base = qkv + (size_t)b * T * rs + hd * AD;
  float qv[4][8];
  own_rows(qv, base, qi, qi < T, rs, h);
  stage3<NJ, PLANE>(lds, base + (size_t)H * AD, T, rs, tid, NT);
  stage3<NJ, PLANE>(lds + TENS, base + (size_t)2 * H * AD, T, rs, tid, NT);
  bf16x8 qf[4][3];
#pragma unroll
  for (int s = 0; s < 4; ++s) split8(qv[s], qf[s]);
  __syncthreads();
  const Tens<PLANE> K{lds}, V{lds + TENS};
  // S^T tiles: rows = keys of tile jt, lane = query
  f32x16 st[NW];
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int jt = 0; jt < NW; ++jt) {
    f32x16 hi, lo;
    zero16(hi), zero16(lo);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 kf[3];
      K.row(kf, jt, s, r, h);
      mma6(hi, lo, kf, qf[s]);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = jt * 32 + acc_row(e, h);
      const float sv = (hi[e] + lo[e]) * scale;
      st[jt][e] = key < T ? sv : -__builtin_huge_valf();
      m = fmaxf(m, st[jt][e]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, kWave));
  float l = 0.f;
#pragma unroll
  for (int jt = 0; jt < NW; ++jt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = expf(st[jt][e] - m);
      st[jt][e] = p;
      l += p;
    }
  l += __shfl_xor(l, 32, kWave);
  const float inv = 1.0f / l;
  if (h == 0 && qi < T) lse[((size_t)b * H + hd) * T + qi] = m + logf(l);
  // P as A fragments, once for both halves of the head dim
  bf16x8 pf[NW][2][3];
#pragma unroll
  for (int jt = 0; jt < NW; ++jt)
#pragma unroll
    for (int s = 0; s < 2; ++s) acc_frag(pf[jt][s], st[jt], s);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16 hi, lo;
    zero16(hi), zero16(lo);
#pragma unroll
    for (int jt = 0; jt < NW; ++jt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 vf[3];
        V.tr(vf, jt, s, dt, lane);
        mma6(hi, lo, pf[jt][s], vf);
      }
    // lane = d column, registers = query rows of tile w; the softmax denominator belongs to the query: fetched from the lane
    // that owns it
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qrow = acc_row(e, h);
      const float iv = __shfl(inv, qrow, kWave);
      const int q = w * 32 + qrow;
      if (q < T) store_wt(&o[((size_t)b * T + q) * H * AD + hd * AD + dt * 32 + r], (hi[e] + lo[e]) * iv);
    }
  }
}

// ---------------------------------------------------------------- backward
// grid (H, B, 2), block = 64 NW threads.  blockIdx.z = 0: dQ (one wave per query tile; K and V in LDS, the wave's own Q, dO,
// O rows global -> registers); blockIdx.z = 1: dK and dV (one wave per key tile; Q and dO in LDS with the log-sum-exp and
// delta = rowsum(dO * O) of every query, the wave's own K, V rows global -> registers).  Two tensors of planes per block:
// 48.5 KB at NW = 2, three blocks per CU -- the one-block form (four tensors, 97 KB, one wave per SIMD) ran its staging,
// its products and its stores strictly one after the other: 46.8 us against the fp32 kernel's 46.3.
template <int NW>
__global__ __launch_bounds__(64 * NW) void attention3_bwd_kernel(int T, int H, float scale, const float* __restrict__ qkv,
                                                                 const float* __restrict__ o, const float* __restrict__ lse,
                                                                 const float* __restrict__ d_o, float* __restrict__ dqkv) {
  constexpr int Tpad = 32 * NW, PLANE = Tpad * 128, TENS = 3 * PLANE, NT = 64 * NW, NJ = Tpad * 8 / NT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, part = blockIdx.z;
  const int lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  const size_t rs = (size_t)3 * H * AD, os = (size_t)H * AD;
  const float* base = qkv + (size_t)b * T * rs + hd * AD;
  const float* gbase = d_o + (size_t)b * T * os + hd * AD;
  const float* obase = o + (size_t)b * T * os + hd * AD;

  if (part == 0) {
    // ---- dQ for query tile w.  X = dS^T (rows = keys, lane = query w 32 + r)
    const int q = w * 32 + r;
    const bool qin = q < T;
    float qv[4][8], gv[4][8], ov[4][8];
    own_rows(qv, base, q, qin, rs, h);
    own_rows(gv, gbase, q, qin, os, h);
    own_rows(ov, obase, q, qin, os, h);
    const float lq = qin ? lse[((size_t)b * H + hd) * T + q] : 0.f;
    stage3<NJ, PLANE>(lds, base + (size_t)H * AD, T, rs, tid, NT);
    stage3<NJ, PLANE>(lds + TENS, base + (size_t)2 * H * AD, T, rs, tid, NT);
    float dlt = 0.f;                                   // delta[q]: this lane holds half of the row (its h), the partner the rest
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) dlt += gv[s][j] * ov[s][j];
    dlt += __shfl_xor(dlt, 32, kWave);
    bf16x8 qf[4][3], gf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) split8(qv[s], qf[s]), split8(gv[s], gf[s]);
    __syncthreads();
    const Tens<PLANE> K{lds}, V{lds + TENS};
    f32x16 dqh[2], dql[2];
    zero16(dqh[0]), zero16(dqh[1]), zero16(dql[0]), zero16(dql[1]);
#pragma unroll
    for (int jt = 0; jt < NW; ++jt) {
      f32x16 sh_, sl_, ph_, pl_;
      zero16(sh_), zero16(sl_), zero16(ph_), zero16(pl_);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 kf[3], vf[3];
        K.row(kf, jt, s, r, h);
        V.row(vf, jt, s, r, h);
        mma6(sh_, sl_, kf, qf[s]);        // S^T[key][query]
        mma6(ph_, pl_, vf, gf[s]);        // dP^T[key][query]
      }
      f32x16 ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = jt * 32 + acc_row(e, h);
        const float p = (key < T && qin) ? expf((sh_[e] + sl_[e]) * scale - lq) : 0.f;
        ds[e] = p * ((ph_[e] + pl_[e]) - dlt) * scale;      // dS^T, scale folded in
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[3];
        acc_frag(af, ds, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8 kf[3];
          K.tr(kf, jt, s, dt, lane);
          mma6(dqh[dt], dql[dt], af, kf);
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qq = w * 32 + acc_row(e, h);
        if (qq < T) store_wt(&dqkv[((size_t)b * T + qq) * rs + hd * AD + dt * 32 + r], dqh[dt][e] + dql[dt][e]);
      }
  } else {
    // ---- dK, dV for key tile w.  X = P, dS (rows = queries, lane = key w 32 + r)
    float* Ls = reinterpret_cast<float*>(lds + 2 * TENS);   // [Tpad] log-sum-exp
    float* Ds = Ls + Tpad;                                   // [Tpad] delta = rowsum(dO * O)
    const int key = w * 32 + r;
    const bool kin = key < T;
    float kv[4][8], vv[4][8];
    own_rows(kv, base + (size_t)H * AD, key, kin, rs, h);
    own_rows(vv, base + (size_t)2 * H * AD, key, kin, rs, h);
    stage3<NJ, PLANE>(lds, base, T, rs, tid, NT);
    {
      float dots[NJ];
      stage3<NJ, PLANE>(lds + TENS, gbase, T, os, tid, NT, obase, dots);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {     // the eight consecutive lanes that staged a row hold its eight partial dot products
        float dl = dots[j];
        dl += __shfl_xor(dl, 4, kWave);
        dl += __shfl_xor(dl, 2, kWave);
        dl += __shfl_xor(dl, 1, kWave);
        const int i = tid + j * NT;
        if ((i & 7) == 0) Ds[i >> 3] = dl;
      }
      if (tid < Tpad) Ls[tid] = tid < T ? lse[((size_t)b * H + hd) * T + tid] : 0.f;
    }
    bf16x8 kf[4][3], vf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) split8(kv[s], kf[s]), split8(vv[s], vf[s]);
    __syncthreads();
    const Tens<PLANE> Q{lds}, dO{lds + TENS};
    f32x16 dkh[2], dkl[2], dvh[2], dvl[2];
    zero16(dkh[0]), zero16(dkh[1]), zero16(dkl[0]), zero16(dkl[1]);
    zero16(dvh[0]), zero16(dvh[1]), zero16(dvl[0]), zero16(dvl[1]);
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      f32x16 sh_, sl_, ph_, pl_;
      zero16(sh_), zero16(sl_), zero16(ph_), zero16(pl_);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 qf[3], gf[3];
        Q.row(qf, it, s, r, h);
        dO.row(gf, it, s, r, h);
        mma6(sh_, sl_, qf, kf[s]);        // S[query][key]
        mma6(ph_, pl_, gf, vf[s]);        // dP[query][key]
      }
      f32x16 pv, ds;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int q = it * 32 + acc_row(e, h);
        const float p = (kin && q < T) ? expf((sh_[e] + sl_[e]) * scale - Ls[q]) : 0.f;
        pv[e] = p;
        ds[e] = p * ((ph_[e] + pl_[e]) - Ds[q]) * scale;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ap[3], ad[3];
        acc_frag(ap, pv, s);
        acc_frag(ad, ds, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          bf16x8 gf[3], qf[3];
          dO.tr(gf, it, s, dt, lane);
          Q.tr(qf, it, s, dt, lane);
          mma6(dvh[dt], dvl[dt], ap, gf);   // dV += P^T dO
          mma6(dkh[dt], dkl[dt], ad, qf);   // dK += dS^T Q
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kk = w * 32 + acc_row(e, h);
        if (kk < T) {
          float* row = dqkv + ((size_t)b * T + kk) * rs + hd * AD + dt * 32 + r;
          store_wt(row + (size_t)H * AD, dkh[dt][e] + dkl[dt][e]);
          store_wt(row + (size_t)2 * H * AD, dvh[dt][e] + dvl[dt][e]);
        }
      }
  }
}

}  // namespace attn3

// 32 < T <= 64: two 32-row tiles (attention.hip dispatches here on the exact-split arithmetic)
int attention3_forward(int B, int T, int H, float scale, const float* qkv, float* o, float* lse, hipStream_t s) {
  constexpr int NW = 2;
  const size_t lds = (size_t)2 * 3 * 32 * NW * 128;
  auto k = attn3::attention3_fwd_kernel<NW>;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipLaunchKernelGGL(k, dim3(H, B), dim3(64 * NW), lds, s, T, H, scale, qkv, o, lse);
  return check_launch("attention_forward");
}

int attention3_backward(int B, int T, int H, float scale, const float* qkv, const float* o, const float* lse, const float* d_o,
                        float* dqkv, hipStream_t s) {
  constexpr int NW = 2;
  const size_t lds = (size_t)2 * 3 * 32 * NW * 128 + 2 * 32 * NW * sizeof(float);
  auto k = attn3::attention3_bwd_kernel<NW>;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipLaunchKernelGGL(k, dim3(H, B, 2), dim3(64 * NW), lds, s, T, H, scale, qkv, o, lse, d_o, dqkv);
  return check_launch("attention_backward");
}

}  // namespace pdae
