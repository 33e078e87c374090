// rows3_kernel.h -- fp32 GEMMs on the bf16 matrix pipe with EXACT-SPLIT operands ("bf16x3, six products").
//
// Every fp32 operand element x is split into three bf16 terms, x = h + m + l EXACTLY (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m); round-to-nearest at each stage, the residuals are exact in fp32).  A product a.b is then the
// sum of nine bf16 x bf16 products, each EXACT in fp32; the six largest -- hh, hm, mh, hl, lh, mm -- are accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16, the three dropped ones (ml, lm, ll) are below 2^-25 of the product.  hh goes to one
// accumulator, the five small terms to a second one (the sum of the small terms is 2^-8 of the result: its roundings
// do not count, and the hh chain sees K / 16 accumulator roundings where v_mfma_f32_32x32x2_f32 sees K / 2), added once
// in the epilogue: measured error against fp64 is at or below the fp32-input MFMA kernels' (tests/test_gpu_rows3.py).
// The bf16 pipe runs 16x the fp32-input MFMA rate: six products are 2.67x the fp32 kernels' ceiling.
//
// The split happens while a tile is staged: global (fp32) -> registers -> split -> LDS as three bf16 planes
// [plane][row][32 k], K CONTIGUOUS whatever the operand's layout in memory -- an operand stored along the reduction
// ([M,K] activations, [N,K] weights) is staged in octets of a row, one stored ACROSS it (the [K,N] weight of a data
// gradient, both operands of a weight gradient) in patches of 8 k x PW columns that a thread transposes in registers.
// No divergent branch in any main loop: threads without a piece of their own repeat another thread's (same loads, same
// stores), rows and columns past an edge are clamped addresses (and zeros by select where they enter a sum).
// Fragments are then one ds_read_b128 per (32-row tile, plane, 16-deep step) for every form.
#pragma once
#include <type_traits>

#include "nt_args.h"
#include "rows_common.h"

namespace pdae {
namespace rows3 {

using rows::Args;
using rows::f32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK3 = 32;          // reduction depth of one staged tile (two 16-deep MFMA steps)
// (lab, tools/lab/gemm3_anatomy.py) -DR3_STAMP: wave 0 of every block writes s_memrealtime (100 MHz) at kernel entry, before
// the first k-tile, behind the last one and behind the epilogue to Args::Z (4 x u64 per block, + s_memtime, the core clock, in 4 more); never
// defined in the library
#ifdef R3_STAMP
#define R3_STAMP_AT(i)                                                                                              \
  if (threadIdx.x == 0) {                                                                                           \
    unsigned long long* st_ = reinterpret_cast<unsigned long long*>(p.Z) + (size_t)(by * gx + bx) * 8; \
    st_[i] = __builtin_amdgcn_s_memrealtime(), st_[4 + (i)] = __builtin_amdgcn_s_memtime();                         \
  }
#else
#define R3_STAMP_AT(i)
#endif
// results leave through write-through stores (common.h store_wt)
#define R3_STORE(ptr, v) store_wt((ptr), (v))
// (lab, round 6) -DR3_PRIO: one s_setprio 1 for the younger half of an 8-wave block: measured, nothing (10.38 / 10.35 ms
// against 10.40 / 10.35)
#ifdef R3_PRIO
#define R3_SETPRIO() do { if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1); } while (0)
#else
#define R3_SETPRIO() do { } while (0)
#endif
#ifndef R3_FPS
#define R3_FPS 3                 // fragment reads per slot behind a tile's barrier (lab: tools/lab/fps_sweep.sh)
#endif

// The split of eight fp32 (an octet: 8 k of one row) into three planes of eight bf16, x = h + m + l exactly, cut in
// eight CHUNKS of 5-6 VALU instructions so that the main loop can place one chunk behind each MFMA: chunks 0-3 take
// the h term of the pairs (v[2q], v[2q+1]) and leave the residual in v, chunks 4-7 the m and l terms.
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  f32x2 x = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}
template <int C>
__device__ __forceinline__ void split_chunk(float (&v)[8], u32x4 (&pk)[3]) {
  constexpr int q = C & 3;
  const unsigned t = cvt_pk_bf16(v[2 * q], v[2 * q + 1]);
  v[2 * q] = v[2 * q] - __uint_as_float(t << 16);
  v[2 * q + 1] = v[2 * q + 1] - __uint_as_float(t & 0xffff0000u);
  if (C < 4) {
    pk[0][q] = t;
  } else {
    pk[1][q] = t;
    pk[2][q] = cvt_pk_bf16(v[2 * q], v[2 * q + 1]);
  }
}

// product q (0..5) of one (A tile, B tile) pair and 16-deep step: the five small terms first into `lo`, hh into `hi`
// (DUAL false: everything into `hi`).  Planes: 0 = h, 1 = m, 2 = l.
#ifdef R3_MFMA16
// (lab, round 6: WRONG RESULTS, timing only) the same FLOPs as two v_mfma_f32_16x16x32_bf16 per product: does the shape the
// guide reports to hold a higher clock under load (MI355X_MICROARCH.md, DVFS give-back item 7) pay in THIS loop?
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16_pair(const bf16x8& a, const bf16x8& b, f32x16& c) {
  f32x4m c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
  c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
  c[0] = c0[0], c[1] = c0[1], c[2] = c0[2], c[3] = c0[3], c[4] = c1[0], c[5] = c1[1], c[6] = c1[2], c[7] = c1[3];
}
#endif
template <bool DUAL, int Q>
__device__ __forceinline__ void mfma_one(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& hi, f32x16& lo) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#ifdef R3_MFMA16
  if (DUAL && Q < 5) mfma16_pair(a[PA[Q]], b[PB[Q]], lo);
  else mfma16_pair(a[PA[Q]], b[PB[Q]], hi);
#else
  if (DUAL && Q < 5) lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[Q]], b[PB[Q]], lo, 0, 0, 0);
  else hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[Q]], b[PB[Q]], hi, 0, 0, 0);
#endif
}

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N - 1
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// 4 x 4 transpose inside a lane quad: afterwards x_k of lane i is what x_i of lane k was (b0 / b1 = lane bits 0 / 1)
__device__ __forceinline__ float dpp_f32_quad_xor1(float v) { return __uint_as_float(dpp_u32<0xB1>(__float_as_uint(v))); }
__device__ __forceinline__ float dpp_f32_quad_xor2(float v) { return __uint_as_float(dpp_u32<0x4E>(__float_as_uint(v))); }
__device__ __forceinline__ void quad_transpose(float& x0, float& x1, float& x2, float& x3, bool b0, bool b1) {
  float t = dpp_f32_quad_xor1(b0 ? x0 : x1);
  if (b0) x0 = t; else x1 = t;
  t = dpp_f32_quad_xor1(b0 ? x2 : x3);
  if (b0) x2 = t; else x3 = t;
  t = dpp_f32_quad_xor2(b1 ? x0 : x2);
  if (b1) x0 = t; else x2 = t;
  t = dpp_f32_quad_xor2(b1 ? x1 : x3);
  if (b1) x1 = t; else x3 = t;
}

// A 32x32x16 operand fragment out of an image stored ACROSS the reduction ([k][column] bf16): two ds_read_b64_tr_b16, each
// handing the lane four consecutive k of its column (cdna_hip_programming.md T10); p0 / p1 address rows 8 h .. 8 h + 3 and
// 8 h + 4 .. 8 h + 7 of the 16-deep step.  EXEC must be all ones (every main loop here is branch-free).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// C[M,N] = epi(A[M,K] . op(B)), the contract of rows::rows_gemm_kernel (same Args, same epilogues, same split-K slab
// form; its stream-K form is not built) for K % 32 == 0.  Block tile (32 TI WM) x (32 TJ WN), WM WN waves of TI x TJ MFMA tiles; an LDS
// tile holds KS 16-deep steps of the reduction (KS = 2: 80-B rows; KS = 1: 48-B rows, half the LDS, so that TWO blocks
// fit a CU and one block's first loads and epilogue run under the other's MFMAs).
// Pipeline per LDS tile t (double-buffered, ONE barrier per tile), written as S = 6 KS TI TJ SLOTS of one MFMA each
// with a share of the side work behind it, the order pinned by a sched_barrier per slot (one wave issues in order: an
// MFMA holds the vector issue port 8 of its 32 cycles, ~5 other instructions ride in its shadow; left to the compiler
// the split lands behind the MFMAs in one block and the waves of a SIMD alternate between an all-MFMA and an all-VALU
// phase in lockstep):
//   slots of step 0 (KS = 2)           + the fragment reads of step 1 (same LDS tile)
//   slots [0, SB - 1)                  + the split chunks of tile t + 1 (A octets, then B), each plane stored to
//                                        LDS[buf ^ 1] when complete; an operand's registers are then re-loaded with tile
//                                        t + 3: TWO register sets, two tiles of global loads in flight
//   barrier behind slot SB - 1
//   slots [SB, S)                      + the fragment reads of tile t + 1, step 0
// Fragments live in two register sets (the stage of a 16-deep step alternates), register sets and fragment stages are
// indexed statically: the tile loop is unrolled by two.  The loop is branch-free: past the end the last tile is loaded
// / split / stored again (into the buffer nobody reads).
// The body of gemm3_kernel for block (bx, by, bz) of a grid gx blocks wide: the kernel below is this and nothing else; a
// persistent caller (tools/lab/chain3_lab.hip) runs it for a (tile, slab) of its own choice.
template <int TI, int TJ, int WM, int WN, int KS, bool BKN, int EPI, bool DUAL, int ABL = 0, bool PERS = false>
__device__ __forceinline__ void gemm3_body(const Args& p, const int bx, const int by, const int bz, const int gx) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * TI * WM, BN = 32 * TJ * WN;
  constexpr int ROWB = KS == 2 ? 80 : 48;                     // bytes per LDS row: 16 KS bf16 + 16 B
  constexpr int BKT = 16 * KS;                                // reduction depth of an LDS tile
  constexpr int OPR = 2 * KS;                                 // octets (8 k) per row and tile
  // BKN (B stored [K, N], across the reduction: the weight of a data gradient), round 6: the B band is staged THE WAY IT LIES
  // IN MEMORY -- octets of a k-ROW (eight consecutive columns, two 16-byte loads; was: 8 k x 1 column patches of 4-byte
  // loads) into a natural [k][BN columns] image of RSB-byte rows, and its fragments are read transposed (tr_frag).  Image
  // byte (k, c) = RSB k + ((2 c) ^ (SWZ(k & 3) << 6)): the four rows a 16-lane group reads land in different 64-byte
  // quarters of the bank space (RSB = 256: quarter k & 3; RSB = 128 / 384: rows alternate halves already, quarter by k >> 1).
  // Needs N % 4 == 0 (16-byte aligned rows; rows_gemm.hip gemm3_takes); same MFMA order: results bit-identical.
  // Used where it pays: the 128-wide tiles (TJ = 2; 128 x 128: -3 % per launch in the step).  The 192-wide tile has no
  // registers left for the three transposed-read bases (it spilled: +15 %) and the 64-wide one gained nothing: they keep
  // the column patches (8 k x 1 column, 4-byte loads, a [column][32 k] image read with ds_read_b128).
  constexpr bool TRB = BKN && TJ == 2 && KS == 2;
  constexpr int RSB = 2 * BN;
  constexpr int PLANE = TRB ? BM * ROWB + BKT * RSB : (BM + BN) * ROWB, BUF = 3 * PLANE;
  constexpr int OA = (BM * OPR + NT - 1) / NT;                // octets of A per thread and tile
  constexpr int OB = (BN * OPR + NT - 1) / NT;                // octets of B: of a row (k-contiguous) / of a k-row (BKN)
  constexpr int G = TI * TJ;
  static_assert(BM % 8 == 0 && BN % 8 == 0, "staging map");
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  R3_STAMP_AT(0);
  // Persistent blocks: block (xcd, slot) walks tiles slot, slot + nslots, ... of its XCD's chunk (n fastest inside a chunk:
  // the re-reads of an A row band hit that XCD's L2) and the LDS tiles of consecutive output tiles form ONE stream
  // through the pipeline -- loads run three positions ahead across the tile boundary, so a boundary costs the epilogue
  // only.  The host launches one residency of blocks when the reduction of a block is an even number >= 4 of LDS tiles
  // (the stream then keeps its buffer / register-set parity across tiles), else one block per tile.  blockIdx.y = the
  // split of the reduction (slabs the consumer adds).  (The fp32-input kernel's stream-K form is not built here.)
  const int chunk = (p.tiles + 7) >> 3;
  const int xcd = bx & 7, nslots = gx >> 3;
  int slot = bx >> 3;
  auto tile_of = [&](int sl) {
    const int t = xcd * chunk + sl;
    return (sl < chunk && t < p.tiles) ? t : -1;
  };
  int tile = tile_of(slot);
  if (tile < 0) return;
  // octet o of an operand tile -> (row, octet of the row): eight consecutive lanes hit eight different 16-B slots of
  // the 128-B window a ds_write_b128 group covers (80-B rows: two rows four apart; 48-B rows: four rows two apart)
  auto octet_row = [](int o) { return KS == 2 ? ((o >> 5) << 3) + (((o >> 2) & 1) << 2) + ((o >> 3) & 3)
                                              : ((o >> 4) << 3) + (((o >> 1) & 3) << 1) + ((o >> 3) & 1); };
  auto octet_col = [](int o) { return KS == 2 ? (o & 3) : (o & 1); };
  const int kbeg = by * p.kchunk, kend = min(p.K, kbeg + p.kchunk), piece = by;
  const int KT = (kend - kbeg) / BKT;
  // 32-bit byte offsets of this thread's staged pieces from the (uniform) operand bases, [0] for the tile being
  // multiplied, [1] for the next one of this block; rows / columns past the matrix edge are clamped (their products
  // are never stored)
  unsigned aoff[2][OA], boff[2][OB], boff2[2][TRB ? OB : 1];   // (boff2: the second 16 bytes of a row-staged B octet)
  int alds[OA], blds[OB];
  auto offsets_of = [&](int t, unsigned (&ao)[OA], unsigned (&bo)[OB], unsigned (&bo2)[TRB ? OB : 1]) __attribute__((always_inline)) {
    const int tm0 = (t / p.tiles_n) * BM, tn0 = (t % p.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < OA; ++i) {
      const int o = (tid + i * NT) % (BM * OPR), row = octet_row(o), oc = octet_col(o);   // (past the tile: an earlier octet again)
      ao[i] = ((unsigned)min(tm0 + row, M - 1) * (unsigned)p.lda + oc * 8) * 4u;
    }
#pragma unroll
    for (int i = 0; i < OB; ++i) {
      const int o = (tid + i * NT) % (BN * OPR);
      if (TRB) {
        const int krow = o / (BN / 8), ch = o % (BN / 8);     // octet: columns 8 ch .. 8 ch + 7 of k-row krow
        // (past the edge: the last whole octet again; N % 8 == 4: the last QUAD twice -- the columns past N are never stored)
        const int c = min(tn0 + 8 * ch, (N & 7) ? N - 4 : N - 8);
        bo[i] = ((unsigned)krow * (unsigned)p.ldb + (unsigned)c) * 4u;
        bo2[i] = bo[i] + (c + 8 <= N ? 16u : 0u);
      } else if (BKN) {
        const int kg = o / BN, col = o % BN;                  // patch: k = 8 kg .. 8 kg + 7 of column col
        bo[i] = ((unsigned)(8 * kg) * (unsigned)p.ldb + (unsigned)min(tn0 + col, N - 1)) * 4u;
      } else {
        const int row = octet_row(o), oc = octet_col(o);
        bo[i] = ((unsigned)min(tn0 + row, N - 1) * (unsigned)p.ldb + oc * 8) * 4u;
      }
    }
  };
#pragma unroll
  for (int i = 0; i < OA; ++i) {
    const int o = (tid + i * NT) % (BM * OPR);
    alds[i] = octet_row(o) * ROWB + octet_col(o) * 16;
  }
#pragma unroll
  for (int i = 0; i < OB; ++i) {
    const int o = (tid + i * NT) % (BN * OPR);
    const int krow = o / (BN / 8), ch = o % (BN / 8);
    blds[i] = TRB ? BM * ROWB + krow * RSB + ((ch * 16) ^ ((BN == 128 ? (krow & 3) : ((krow >> 1) & 1)) << 6))
              : BKN ? (BM + o % BN) * ROWB + (o / BN) * 16 : (BM + octet_row(o)) * ROWB + octet_col(o) * 16;
  }
  offsets_of(tile, aoff[0], boff[0], boff2[0]);
  // PERS false (a launch of one tile per block: most of the Transformer blocks' GEMMs): no next tile, and none of its
  // registers or address selects in the loop
  int next = PERS ? tile_of(slot + nslots) : -1;
#pragma unroll
  for (int i = 0; i < OA; ++i) aoff[1][i] = aoff[0][i];
#pragma unroll
  for (int i = 0; i < OB; ++i) boff[1][i] = boff[0][i];
#pragma unroll
  for (int i = 0; i < (TRB ? OB : 1); ++i) boff2[1][i] = boff2[0][i];
  if (PERS && next >= 0) offsets_of(next, aoff[1], boff[1], boff2[1]);
  {
    const char* Ab = reinterpret_cast<const char*>(p.A + (size_t)bz * p.strideA) + (size_t)kbeg * 4;
    const char* Bb = reinterpret_cast<const char*>(p.B + (size_t)bz * p.strideB) +
                     (BKN ? (size_t)kbeg * p.ldb * 4 : (size_t)kbeg * 4);
    const size_t bstep = BKN ? (size_t)p.ldb * 4 : 4;         // bytes per unit of k in B
    const unsigned ldb4 = (unsigned)p.ldb * 4u;
    (void)ldb4;

    float ra[2][OA][8], rb[2][OB][8];                         // two register sets of staged fp32
    u32x4 pka[OA][3], pkb[OB][3];
    // position `pos` of the stream: LDS tile pos of this output tile, or pos - KT of the block's next one (past the last
    // tile: the last LDS tile again, split into the buffer nobody reads)
    auto gload_a = [&](auto set_c, int pos) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      const bool nx = PERS && pos >= KT && next >= 0;
      const char* Ak = Ab + (size_t)(nx ? pos - KT : min(pos, KT - 1)) * BKT * 4;
#pragma unroll
      for (int i = 0; i < OA; ++i) {
        {
          const unsigned ao = nx ? aoff[1][i] : aoff[0][i];
          const float4 v0 = *reinterpret_cast<const float4*>(Ak + ao);
          const float4 v1 = *reinterpret_cast<const float4*>(Ak + ao + 16);
          ra[set][i][0] = v0.x, ra[set][i][1] = v0.y, ra[set][i][2] = v0.z, ra[set][i][3] = v0.w;
          ra[set][i][4] = v1.x, ra[set][i][5] = v1.y, ra[set][i][6] = v1.z, ra[set][i][7] = v1.w;
        }
      }
    };
    auto gload_b = [&](auto set_c, int pos) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      const bool nx = PERS && pos >= KT && next >= 0;
      const char* Bk = Bb + (size_t)(nx ? pos - KT : min(pos, KT - 1)) * BKT * bstep;
#pragma unroll
      for (int i = 0; i < OB; ++i) {
        {
          const unsigned bo = nx ? boff[1][i] : boff[0][i];
          if constexpr (BKN && !TRB) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rb[set][i][q] = *reinterpret_cast<const float*>(Bk + bo + q * ldb4);
          } else {
            // (TRB: the second half through its own offset -- an octet that ends at column N, N % 8 == 4, reads the first half
            // twice; a block-uniform branch here made the compiler hoist the load out of both arms as four dword loads)
            const float4 v0 = *reinterpret_cast<const float4*>(Bk + bo);
            const float4 v1 = *reinterpret_cast<const float4*>(Bk + (TRB ? (nx ? boff2[1][TRB ? i : 0] : boff2[0][TRB ? i : 0]) : bo + 16));
            rb[set][i][0] = v0.x, rb[set][i][1] = v0.y, rb[set][i][2] = v0.z, rb[set][i][3] = v0.w;
            rb[set][i][4] = v1.x, rb[set][i][5] = v1.y, rb[set][i][6] = v1.z, rb[set][i][7] = v1.w;
          }
        }
      }
    };
    // chunk c of the side work of one tile: octet c / 8 (A octets first), chunk c % 8 of its split; the h plane is
    // stored behind chunk 3, the m and l planes behind chunk 7
    auto side_chunk = [&](auto c_c, auto set_c, int buf, auto abl_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value, o = c / 8, ch = c % 8, abl = decltype(abl_c)::value;
      constexpr int set = decltype(set_c)::value;
      auto run = [&](float (&v)[8], u32x4 (&pk)[3], char* d) __attribute__((always_inline)) {
        if constexpr (abl & 2) {                              // (lab) no split arithmetic: the raw bits as "planes"
          if (ch < 4) pk[0][ch] = __float_as_uint(v[2 * ch]), pk[1][ch] = __float_as_uint(v[2 * ch + 1]), pk[2][ch] = pk[0][ch];
        } else {
          split_chunk<ch>(v, pk);
        }
        if constexpr (abl & 4) {
          if (ch == 7) asm volatile("" ::"v"(pk[0]), "v"(pk[1]), "v"(pk[2]));
          return;
        }
        if (ch == 3) *reinterpret_cast<u32x4*>(d) = pk[0];
        if (ch == 7) {
          *reinterpret_cast<u32x4*>(d + PLANE) = pk[1];
          *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pk[2];
        }
      };
      if constexpr (o < OA) {
        run(ra[set][o], pka[o], lds3 + buf * BUF + alds[o]);
      } else {
        constexpr int i = o - OA;
        run(rb[set][i], pkb[i], lds3 + buf * BUF + blds[i]);
      }
    };

    f32x16 hi[TI][TJ], lo[TI][TJ];
    bf16x8 fa[2][TI][3], fb[2][TJ][3];                        // fragments: [stage][tile][plane]
    const int fa_off = (wm * TI * 32 + r) * ROWB + 16 * h, fb_off = (BM + wn * TJ * 32 + r) * ROWB + 16 * h;
    // BKN: lane 4 q + p of a 16-lane group addresses k-row 8 h + q, columns 4 p .. 4 p + 3 of its 4 x 16 block of B tile T
    int fbt_off[TJ];
    {
      const int lq = (lane >> 2) & 3, lp = lane & 3, lcb = (lane >> 4) & 1;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int T = wn * TJ + j;
        fbt_off[j] = BM * ROWB + RSB * (8 * h + lq) + 64 * (T ^ (BN == 128 ? lq : (lq >> 1))) + 32 * lcb + 16 * (lp >> 1) + 8 * (lp & 1);
      }
    }
    constexpr int NF = 3 * (TI + TJ);                         // fragment reads of one 16-deep step
    // read f of a step, in the order the MFMAs want them: A0.l B0.h A0.h B0.l A0.m B0.m, then the other tiles
    auto frag_one = [&](auto f_c, auto st_c, int buf, int s16) __attribute__((always_inline)) {
      constexpr int f = decltype(f_c)::value, st = decltype(st_c)::value;
      constexpr int FPL[6] = {2, 0, 0, 2, 1, 1};
      constexpr bool isa = f < 6 ? (f % 2 == 0) : (f - 6 < 3 * (TI - 1));
      constexpr int tl = f < 6 ? 0 : (isa ? 1 + (f - 6) / 3 : 1 + (f - 6 - 3 * (TI - 1)) / 3);
      constexpr int pl = f < 6 ? FPL[f] : (isa ? (f - 6) % 3 : (f - 6 - 3 * (TI - 1)) % 3);
      const char* base = lds3 + buf * BUF + s16 * 32 + pl * PLANE;
      if constexpr (isa) {
        fa[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + fa_off + tl * 32 * ROWB);
      } else if constexpr (TRB) {
        const char* bt = lds3 + buf * BUF + pl * PLANE + s16 * (16 * RSB) + fbt_off[tl];
        fb[st][tl][pl] = tr_frag(bt, bt + 4 * RSB);
      } else {
        fb[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + fb_off + tl * 32 * ROWB);
      }
    };
    constexpr int S = 6 * G * KS;                             // slots (MFMAs) of a tile
    constexpr int NR = (NF + R3_FPS - 1) / R3_FPS;            // slots behind the barrier: R3_FPS fragment reads each
    constexpr int SB = S - NR;                                // the barrier sits behind slot SB - 1
    constexpr int NC = 8 * (OA + OB);                         // split chunks of a tile
    constexpr int SC = SB - 1 > 0 ? SB - 1 : 1;               // ... spread over slots [0, SC)
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    // one LDS tile kt of parity P: MFMAs on LDS[P] (fragment stage P for KS = 1; stages 0, 1 for the two steps of
    // KS = 2), split of register set P ^ 1 (tile kt + 1) into LDS[P ^ 1], loads of tile kt + 3 into that set
    auto ktile = [&](auto par_c, int kt) __attribute__((always_inline)) {
      constexpr int P = decltype(par_c)::value;
      using SetN = std::integral_constant<int, P ^ 1>;
      static_for<S>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        constexpr int step = s / (6 * G), g = (s % (6 * G)) / 6, q = s % 6;
        constexpr int stg = KS == 2 ? step : P;               // fragment stage of this slot's MFMA
        mfma_one<DUAL, q>(fa[stg][g / TJ], fb[stg][g % TJ], hi[g / TJ][g % TJ], lo[g / TJ][g % TJ]);
        if constexpr (KS == 2) {                              // fragment reads of step 1: NF reads over the step-0 slots
          static_for<NF>([&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            if constexpr (f * (6 * G) / NF == s && !(ABL & 16)) frag_one(f_c, C1{}, P, 1);
          });
        }
        static_for<NC>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          if constexpr (c * SC / NC == s) side_chunk(c_c, SetN{}, P ^ 1, std::integral_constant<int, ABL>{});
        });
        if constexpr (s == (8 * OA - 1) * SC / NC && !(ABL & 1)) gload_a(SetN{}, kt + 3);
        if constexpr (s == (NC - 1) * SC / NC && !(ABL & 1)) gload_b(SetN{}, kt + 3);
        if constexpr (s == SB - 1 && !(ABL & 8)) __syncthreads();
        if constexpr (s >= SB) {
          static_for<NF>([&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            if constexpr (f / R3_FPS == s - SB && !(ABL & 16))
              frag_one(f_c, std::integral_constant<int, (KS == 2 ? 0 : P ^ 1)>{}, P ^ 1, 0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    R3_SETPRIO();
    if constexpr (EPI == rows::EPI_BNRELU_STATS) {             // the block's running column sums (LDS, behind `red`)
      float* accb = reinterpret_cast<float*>(lds3 + 2 * BUF) + WM * 2 * BN;
      for (int c = tid; c < 2 * N; c += NT) accb[c] = 0.f;
    }
    if (KT > 0) {
      gload_a(C0{}, 0);
      gload_b(C0{}, 0);
      gload_a(C1{}, 1);
      gload_b(C1{}, 1);
      static_for<NC>([&](auto c_c) { side_chunk(c_c, C0{}, 0, C0{}); });
      gload_a(C0{}, 2);
      gload_b(C0{}, 2);
      __syncthreads();
      static_for<NF>([&](auto f_c) { frag_one(f_c, C0{}, 0, 0); });
    }
    for (;;) {
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) hi[i][j][e] = 0.f, lo[i][j][e] = 0.f;
    // EPI_BNRELU_STATS: the tile's BatchNorm inputs X are requested HERE, ahead of the k-loop: they travel behind the two
    // k-tiles of operand loads already in flight and are in registers long before the epilogue wants them (requested in
    // the epilogue they were a cold HBM round trip per tile: 186 against 145 us per launch of the plain data gradient)
    float zv[EPI == rows::EPI_BNRELU_STATS ? TJ : 1][EPI == rows::EPI_BNRELU_STATS ? TI : 1][16];
    if constexpr (EPI == rows::EPI_BNRELU_STATS) {
      const unsigned ldz = (unsigned)p.ldc;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + (wn * TJ + j) * 32 + r;
        const bool colok = col < N;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int rtile = m0 + (wm * TI + i) * 32;                  // (wave-uniform; M % 32 == 0 with a list)
          int zrow0 = rtile;
          if (p.z_groups) zrow0 = rtile < M ? __builtin_amdgcn_readfirstlane(p.z_groups[rtile >> 5]) * 32 : 0;
          const size_t zoff = (size_t)(zrow0 + 4 * h) * ldz + col;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int lr = (e & 3) + 8 * (e >> 2);
            zv[j][i][e] = (colok && rtile + 4 * h + lr < M) ? p.Z[zoff + (unsigned)lr * ldz] : 0.f;
          }
        }
      }
    }
    R3_STAMP_AT(1);
    for (int kt = 0; kt < KT; kt += 2) {
      ktile(C0{}, kt);
      if (kt + 1 < KT) ktile(C1{}, kt + 1);
    }
    R3_STAMP_AT(2);

    // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float* Cs = p.C + (size_t)bz * p.strideC + (size_t)piece * p.slab;
    auto epilogue = [&](auto full_c, auto zero_c) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_c)::value;
      constexpr bool ZERO = decltype(zero_c)::value;
      const unsigned ldc = (unsigned)p.ldc;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + (wn * TJ + j) * 32 + r;
        const bool colok = FULL || col < N;
        const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int rbase = m0 + (wm * TI + i) * 32 + 4 * h;
          const size_t off = (size_t)rbase * ldc + col;
          float zv[16];
          if ((EPI == rows::EPI_MUL_GELUGRAD || EPI == rows::EPI_MUL_POS) && !ZERO) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int lr = (e & 3) + 8 * (e >> 2);
              zv[e] = (FULL || (colok && rbase + lr < M)) ? p.Z[off + (unsigned)lr * ldc] : 0.f;
            }
          }
#ifdef R3_WIDE_STORES
          // (lab, round 6: measured SLOWER -- the step 10.26 / 10.27 / 10.28 ms against 10.11 / 10.12 / 10.13 with the 4-byte
          // stores, interleaved on one box; results bit-identical.  Not compiled in.)
          // Whole tiles, rows of 16-byte multiples: the 16 results of a lane (one column, 16 rows) are transposed 4 x 4 inside
          // each lane quad (two DPP rounds), after which lane 4 c + i of a quad holds row i of the four columns 4 c .. 4 c + 3:
          // FOUR 16-byte write-through stores per 32 x 32 tile and lane instead of sixteen 4-byte ones -- a wave instruction
          // writes eight whole 128-byte row segments (guide T21; narrow sc1 stores are one fabric write each).
          if constexpr (FULL && !ZERO) {
            if ((ldc & 3u) == 0) {
              float vv[16], gg[16];
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                float v = (DUAL ? hi[i][j][e] + lo[i][j][e] : hi[i][j][e]) + bv;
                if (EPI == rows::EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
                if (EPI == rows::EPI_BIAS_GELU2) {
                  float ge, gr;
                  gelu_pair_f(v, ge, gr);
                  gg[e] = gr;
                  v = ge;
                }
                if (EPI == rows::EPI_MUL_GELUGRAD) v *= zv[e];
                if (EPI == rows::EPI_MUL_POS) v = zv[e] > 0.f ? v : 0.f;
                vv[e] = v;
              }
              const bool b0 = lane & 1, b1 = lane & 2;
              const size_t qoff = (size_t)(rbase + (lane & 3)) * ldc + (col - (lane & 3));
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4) {
                quad_transpose(vv[4 * g4], vv[4 * g4 + 1], vv[4 * g4 + 2], vv[4 * g4 + 3], b0, b1);
                store_wt4(&Cs[qoff + (unsigned)(8 * g4) * ldc], make_float4(vv[4 * g4], vv[4 * g4 + 1], vv[4 * g4 + 2], vv[4 * g4 + 3]));
                if (EPI == rows::EPI_BIAS_GELU2) {
                  quad_transpose(gg[4 * g4], gg[4 * g4 + 1], gg[4 * g4 + 2], gg[4 * g4 + 3], b0, b1);
                  store_wt4(&p.Z[qoff + (unsigned)(8 * g4) * ldc], make_float4(gg[4 * g4], gg[4 * g4 + 1], gg[4 * g4 + 2], gg[4 * g4 + 3]));
                }
              }
              continue;
            }
          }
#endif
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int lr = (e & 3) + 8 * (e >> 2);
            if (!FULL && !(colok && rbase + lr < M)) continue;
            if (ZERO) {
              Cs[off + (unsigned)lr * ldc] = 0.f;
              continue;
            }
            float v = (DUAL ? hi[i][j][e] + lo[i][j][e] : hi[i][j][e]) + bv;
            if (EPI == rows::EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
            if (EPI == rows::EPI_BIAS_GELU2) {
              float ge, gr;
              gelu_pair_f(v, ge, gr);                             // (common.h: one exponential for GELU and GELU')
              R3_STORE(&p.Z[off + (unsigned)lr * ldc], gr);
              v = ge;
            }
            if (EPI == rows::EPI_MUL_GELUGRAD) v *= zv[e];
            if (EPI == rows::EPI_MUL_POS) v = zv[e] > 0.f ? v : 0.f;
            R3_STORE(&Cs[off + (unsigned)lr * ldc], v);
          }
        }
      }
    };
    // EPI_BNRELU_STATS: t = relu'(bn(x)) ? acc : 0 stored; per column sum t and sum t xhat over the tile's rows: a lane's 16
    // rows, its partner lane, then the WM waves of the column through LDS (`red`, behind the tile buffers: they already
    // hold the next tile of a persistent block), added to the BLOCK's running sums (`accb` [2][N], LDS): one partial row
    // per block leaves at its end, whatever the number of tiles it walked.  No atomics: the finishing pass adds the
    // blocks' rows in order, in fp64.  (The tile's X values were requested ahead of the k-loop.)
    auto epilogue_bn = [&](auto full_c) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_c)::value;
      const unsigned ldc = (unsigned)p.ldc;
      float* red = reinterpret_cast<float*>(lds3 + 2 * BUF);          // [WM][2][BN]
      float* accb = red + WM * 2 * BN;                                // [2][N]
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + (wn * TJ + j) * 32 + r;
        const bool colok = FULL || col < N;
        const float sc = colok ? p.bn_scale[col] : 0.f, sh = colok ? p.bn_shift[col] : 0.f;
        const float mu = colok ? p.bn_mean[col] : 0.f, is = colok ? p.bn_invstd[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const int rbase = m0 + (wm * TI + i) * 32 + 4 * h;
          const size_t off = (size_t)rbase * ldc + col;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int lr = (e & 3) + 8 * (e >> 2);
            const bool in = FULL || (colok && rbase + lr < M);
            const float v = hi[i][j][e] + lo[i][j][e];
            const float t = (in && zv[j][i][e] * sc + sh > 0.f) ? v : 0.f;
            if (in) R3_STORE(&Cs[off + (unsigned)lr * ldc], t);
            s1 += t;
            s2 += t * ((zv[j][i][e] - mu) * is);
          }
        }
        s1 += __shfl_xor(s1, 32, kWave);
        s2 += __shfl_xor(s2, 32, kWave);
        if (h == 0) {
          red[(wm * 2 + 0) * BN + (wn * TJ + j) * 32 + r] = s1;
          red[(wm * 2 + 1) * BN + (wn * TJ + j) * 32 + r] = s2;
        }
      }
      __syncthreads();
      if (tid < 2 * BN) {
        const int half = tid / BN, cc = tid - half * BN;
        if (n0 + cc < N) {
          float t = red[half * BN + cc];
#pragma unroll
          for (int k = 1; k < WM; ++k) t += red[(k * 2 + half) * BN + cc];
          accb[half * N + n0 + cc] += t;               // (this thread owns the entry for every tile of these columns)
        }
      }
      __syncthreads();                                  // red is written again by the next tile's epilogue
    };
    const bool full = m0 + BM <= M && n0 + BN <= N;
    if constexpr (EPI == rows::EPI_BNRELU_STATS) {
      if (full) epilogue_bn(std::true_type{});
      else epilogue_bn(std::false_type{});
    } else {
      if (full) epilogue(std::true_type{}, std::false_type{});
      else epilogue(std::false_type{}, std::false_type{});
    }
    R3_STAMP_AT(3);
    if (!PERS || next < 0) break;
    slot += nslots;
    tile = next;
#pragma unroll
    for (int i = 0; i < OA; ++i) aoff[0][i] = aoff[1][i];
#pragma unroll
    for (int i = 0; i < OB; ++i) boff[0][i] = boff[1][i];
#pragma unroll
    for (int i = 0; i < (TRB ? OB : 1); ++i) boff2[0][i] = boff2[1][i];
    next = tile_of(slot + nslots);
    if (next >= 0) offsets_of(next, aoff[1], boff[1], boff2[1]);
    }   // tiles of this block
    if constexpr (EPI == rows::EPI_BNRELU_STATS) {             // one partial row of column sums per block
      const float* accb = reinterpret_cast<const float*>(lds3 + 2 * BUF) + WM * 2 * BN;
      __syncthreads();
      for (int c = tid; c < 2 * N; c += NT) p.stats_part[(size_t)bx * 2 * N + c] = accb[c];
    }
  }
}

template <int TI, int TJ, int WM, int WN, int KS, bool BKN, int EPI, bool DUAL, int ABL = 0, bool PERS = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm3_kernel(const Args p) {
  gemm3_body<TI, TJ, WM, WN, KS, BKN, EPI, DUAL, ABL, PERS>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradients dW_p[N_p, K_p] = dY_p^T . X_p on the same arithmetic and on gemm3_kernel's pipeline: the work
// layout, the partial-tile slots and the ordered reduction of rows::wgrad_kernel (rows_common.h); output tile 128 (n)
// x 128 (k), eight waves of 32 x 64, the reduction over rows m in 32-row LDS tiles (two 16-deep steps per barrier),
// fragments in two register sets, TWO register sets of staged fp32 (two tiles of global loads in flight; a first
// version with one set and 64 x 64 wave tiles waited for every tile's loads: 100 vs 130 TFLOP/s).  Both operands are
// stored ACROSS the reduction ([m][column]): a thread's eight loads run down the rows, the transpose is free in
// registers, and LDS holds [plane][column][32 m] exactly as gemm3_kernel's tiles.  Both bands are staged
// in patches of 8 m x 1 column: thread -> (m-octet tid / 128, column tid % 128) of the dY band AND of the X band, row
// pointers scalar, one 32-bit column offset per band.  Rows past the end of a segment, columns past the matrix edge,
// the BatchNorm + ReLU producer and the column sums are applied to a register set right before it is split
// (`fixup`, skipped by a block-uniform branch on whole tiles without a producer).
template <bool FORMS>
__global__ __launch_bounds__(512) void wgrad3b_kernel(const rows::WgradArgs g) {
  using rows::WgradProb;
  constexpr int TM = rows::WTM, TN = 128, WCH = rows::WCH, BKT = 32, ROWB = 80;
  constexpr int PLANE = (TM + TN) * ROWB, BUF = 3 * PLANE;
  constexpr int TI = 1, TJ = 2, G = TI * TJ, WN = 2;
  constexpr int WSLOT = rows::wslot(TN);
  constexpr bool DUAL = true;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int col = tid & 127, kg = tid >> 7;
  const int kg_u = __builtin_amdgcn_readfirstlane(kg);
  const int alds = col * ROWB + kg * 16, blds = (TM + col) * ROWB + kg * 16;
  const int nb8 = g.blocks >> 3;
  const int sb = g.blocks % 8 == 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  long long u = rows::wg_start(g, sb);
  const long long uend = rows::wg_start(g, sb + 1);
  float* slot = g.partials + (size_t)sb * g.slots * WSLOT;
  const int fa_off = (wm * TI * 32 + r) * ROWB + 16 * h, fb_off = (TM + wn * TJ * 32 + r) * ROWB + 16 * h;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  R3_SETPRIO();
  for (; u < uend; slot += WSLOT) {
    const WgradProb& P = g.p[rows::wg_prob_of_unit(g, u)];
    const long long rel = u - P.unit0;
    const int lt = P.lt0 + (int)(rel / P.chunks), c0 = (int)(rel % P.chunks);
    const int c1 = (int)min((long long)P.chunks, c0 + (uend - u));
    u += c1 - c0;
    const int bx = lt % P.tk, by = lt / P.tk;
    const int n0 = by * TM, k0 = bx * TN;
    const int mbeg = c0 * WCH, mend = min(P.M, c1 * WCH), mlast = mend - 1;
    const int KT = (mend - mbeg + BKT - 1) / BKT;
    const bool aok = n0 + col < P.N, bok = k0 + col < P.K;
    const unsigned acol = (unsigned)min(n0 + col, P.N - 1), bcol = (unsigned)min(k0 + col, P.K - 1);
    const unsigned lda = P.N, ldb = P.K;
    const float* const Ab = P.dY;
    const float* const Bb = P.X;
    const bool edge = n0 + TM > P.N || k0 + TN > P.K;             // (block-uniform)
    const bool sums = P.db != nullptr && bx == 0;
    const int* const grp_a = FORMS ? g.a_groups : nullptr;
    const int* const grp_b = FORMS ? g.b_groups : nullptr;
    const bool bnrelu = FORMS && g.scale != nullptr;
    float bsc = 1.f, bsh = 0.f, asum = 0.f;
    if (bnrelu && bok) bsc = g.scale[k0 + col], bsh = g.shift[k0 + col];

    float ra[2][8], rb[2][8];
    u32x4 pka[3], pkb[3];
    // byte offsets of this thread's patch (rows 8 kg .. 8 kg + 7 of a 32-row tile, one column) from the tile's first row:
    // with them a whole tile's sixteen loads share ONE scalar base per operand (global_load_dword v, voff, s[base]) -- the
    // per-row 64-bit pointer arithmetic was 142 of this loop's 388 instructions per tile and wave, and a wave issues one
    // instruction every ~4 cycles: more than the 1536 cycles the matrix pipe needs for the tile (tools/lab/NOTES.md)
    unsigned voa[8], vob[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      voa[q] = ((unsigned)(8 * kg + q) * lda + acol) * 4u;
      vob[q] = ((unsigned)(8 * kg + q) * ldb + bcol) * 4u;
    }
    auto gload = [&](auto set_c, int kt) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      const int mt = mbeg + min(kt, KT - 1) * BKT;
      int offa = 0, offb = 0;                          // stored row - product row of this tile (listed operands)
      if (FORMS) {
        const int gq = mt >> 5;
        if (grp_a) offa = (grp_a[gq] - gq) * 32;
        if (grp_b) offb = (grp_b[gq] - gq) * 32;
      }
      if (mt + BKT <= mend) {                          // (block-uniform) a whole tile: no row clamps
        const char* pa = reinterpret_cast<const char*>(Ab + (size_t)(mt + offa) * lda);
        const char* pb = reinterpret_cast<const char*>(Bb + (size_t)(mt + offb) * ldb);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          ra[set][q] = *reinterpret_cast<const float*>(pa + voa[q]);
          rb[set][q] = *reinterpret_cast<const float*>(pb + vob[q]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int m = min(mt + 8 * kg_u + q, mlast);   // (scalar; rows past the end re-read the last row)
          ra[set][q] = *(Ab + (size_t)(m + offa) * lda + acol);
          rb[set][q] = *(Bb + (size_t)(m + offb) * ldb + bcol);
        }
      }
    };
    // what a register set still needs before it is split: producer, zeros past the edges, the column sums
    auto fixup = [&](auto set_c, int kt) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      const int mt = mbeg + kt * BKT;
      const bool live = kt < KT;                       // (tiles past the end are split into the buffer nobody reads)
      if (bnrelu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) rb[set][q] = fmaxf(rb[set][q] * bsc + bsh, 0.f);
      }
      if (edge || mt + BKT > mend) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool in = mt + 8 * kg_u + q < mend;
          ra[set][q] = (in && aok) ? ra[set][q] : 0.f;
          rb[set][q] = (in && bok) ? rb[set][q] : 0.f;
        }
      }
      if (sums && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += ra[set][q];
        asum += t;
      }
    };
    auto side_chunk = [&](auto c_c, auto set_c, int buf) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value, o = c / 8, ch = c % 8, set = decltype(set_c)::value;
      auto run = [&](float (&v)[8], u32x4 (&pk)[3], char* d) __attribute__((always_inline)) {
        split_chunk<ch>(v, pk);
        if (ch == 3) *reinterpret_cast<u32x4*>(d) = pk[0];
        if (ch == 7) {
          *reinterpret_cast<u32x4*>(d + PLANE) = pk[1];
          *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pk[2];
        }
      };
      if constexpr (o == 0) run(ra[set], pka, lds3 + buf * BUF + alds);
      else run(rb[set], pkb, lds3 + buf * BUF + blds);
    };
    f32x16 hi[TI][TJ], lo[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) hi[i][j][e] = 0.f, lo[i][j][e] = 0.f;
    bf16x8 fa[2][TI][3], fb[2][TJ][3];
    constexpr int NF = 3 * (TI + TJ);
    auto frag_one = [&](auto f_c, auto st_c, int buf, int s16) __attribute__((always_inline)) {
      constexpr int f = decltype(f_c)::value, st = decltype(st_c)::value;
      constexpr int FPL[6] = {2, 0, 0, 2, 1, 1};
      constexpr bool isa = f < 6 ? (f % 2 == 0) : (f - 6 < 3 * (TI - 1));
      constexpr int tl = f < 6 ? 0 : (isa ? 1 + (f - 6) / 3 : 1 + (f - 6 - 3 * (TI - 1)) / 3);
      constexpr int pl = f < 6 ? FPL[f] : (isa ? (f - 6) % 3 : (f - 6 - 3 * (TI - 1)) % 3);
      const char* base = lds3 + buf * BUF + s16 * 32 + pl * PLANE;
      if constexpr (isa) fa[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + fa_off + tl * 32 * ROWB);
      else fb[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + fb_off + tl * 32 * ROWB);
    };
    constexpr int S = 12 * G, NR = (NF + R3_FPS - 1) / R3_FPS, SB = S - NR, NC = 16, SC = SB - 1;
    auto ktile = [&](auto par_c, int kt) __attribute__((always_inline)) {
      constexpr int P2 = decltype(par_c)::value;
      using SetN = std::integral_constant<int, P2 ^ 1>;
      fixup(SetN{}, kt + 1);
      static_for<S>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        constexpr int step = s / (6 * G), gi = (s % (6 * G)) / 6, q = s % 6;
        mfma_one<DUAL, q>(fa[step][gi / TJ], fb[step][gi % TJ], hi[gi / TJ][gi % TJ], lo[gi / TJ][gi % TJ]);
        static_for<NF>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          if constexpr (f * (6 * G) / NF == s) frag_one(f_c, C1{}, P2, 1);
        });
        static_for<NC>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          if constexpr (c * SC / NC == s) side_chunk(c_c, SetN{}, P2 ^ 1);
        });
        if constexpr (s == (NC - 1) * SC / NC) gload(SetN{}, kt + 3);
        if constexpr (s == SB - 1) __syncthreads();
        if constexpr (s >= SB) {
          static_for<NF>([&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            if constexpr (f / R3_FPS == s - SB) frag_one(f_c, C0{}, P2 ^ 1, 0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    gload(C0{}, 0);
    gload(C1{}, 1);
    fixup(C0{}, 0);
    static_for<NC>([&](auto c_c) { side_chunk(c_c, C0{}, 0); });
    gload(C0{}, 2);
    __syncthreads();
    static_for<NF>([&](auto f_c) { frag_one(f_c, C0{}, 0, 0); });
    for (int kt = 0; kt < KT; kt += 2) {
      ktile(C0{}, kt);
      if (kt + 1 < KT) ktile(C1{}, kt + 1);
    }
    // ---- the partial tile (whole TM x TN, edges included: the reduction stores what is inside)
    __syncthreads();
    if (sums) {
      float* red = reinterpret_cast<float*>(lds3);     // [4 m-octets][TM]; the tile buffers are free now
      red[kg * TM + col] = asum;
      __syncthreads();
      if (tid < TM) {
        const float t = (red[tid] + red[TM + tid]) + (red[2 * TM + tid] + red[3 * TM + tid]);
        if (!g.direct) slot[TM * TN + tid] = t;
        else if (n0 + tid < P.N) P.db[n0 + tid] = t;
      }
      __syncthreads();
    }
    if (g.direct) {                                    // this block summed the tile over ALL rows: the result, not a partial
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = n0 + (wm * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int cc = k0 + (wn * TJ + j) * 32 + r;
            if (row < P.N && cc < P.K) R3_STORE(&P.dW[(size_t)row * P.K + cc], hi[i][j][e] + lo[i][j][e]);
          }
      continue;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (wm * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          R3_STORE(&slot[row * TN + (wn * TJ + j) * 32 + r], hi[i][j][e] + lo[i][j][e]);
        }
  }
}


// ---------------------------------------------------------------------------------------------
// wgrad3b_kernel's contract with the operands staged THE WAY THEY LIE IN MEMORY (round 6): both bands are stored across
// the reduction ([m][column]), so a thread loads one OCTET OF A ROW -- eight consecutive columns of one of the tile's 32
// rows, two 16-byte loads per operand instead of eight 4-byte loads down a column -- splits it with the same split_chunk
// and stores each plane with one ds_write_b128 into a natural [plane][m][128 columns] image (256-byte rows, no padding:
// 96 KB for the two buffers instead of 122).  The transpose moves to the fragment reads: ds_read_b64_tr_b16 hands a lane
// four consecutive m of ITS column (cdna_hip_programming.md T10), two of them make the eight k of a 32x32x16 operand.
// Image: byte (m, c) = 256 m + ((2 c) ^ ((m & 3) << 6)): the four rows a 16-lane group reads land in the four 64-byte
// quarters of the bank space, so a half-wave's 32 addresses cover all 64 banks once (conflict-free by construction; the
// stores are eight consecutive 16-byte chunks of one row).  Same MFMA order as wgrad3b_kernel: dW is bit-identical; the
// column sums (db) are added in a different order (a thread owns eight columns of one row, not one column of eight).
// Needs N % 8 == 0, K % 8 == 0 and 16-byte aligned operands (the host falls back to wgrad3b_kernel otherwise).

template <bool FORMS>
__global__ __launch_bounds__(512) void wgrad3t_kernel(const rows::WgradArgs g) {
  using rows::WgradProb;
  constexpr int TM = rows::WTM, TN = 128, WCH = rows::WCH, BKT = 32;
  constexpr int PROW = 256, BAND = BKT * PROW, PLANE = 2 * BAND, BUF = 3 * PLANE;
  constexpr int TI = 1, TJ = 2, G = TI * TJ, WN = 2;
  constexpr int WSLOT = rows::wslot(TN);
  constexpr bool DUAL = true;
  static_assert(TM == 128, "one 256-byte image row per band");
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int srow = tid >> 4, sch = tid & 15;                  // staged octet: row srow of the tile, columns 8 sch .. 8 sch + 7
  const int srow_u = __builtin_amdgcn_readfirstlane(srow) & ~3;   // (a wave stages rows 4 w .. 4 w + 3)
  const int wlds = srow * PROW + ((sch * 16) ^ ((srow & 3) << 6));
  // transposed fragment reads: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of its 4 x 16 block
  const int lq = (lane >> 2) & 3, lp = lane & 3, lcb = (lane >> 4) & 1;
  auto tr_off = [&](int T) { return PROW * (8 * h + lq) + 64 * (T ^ lq) + 32 * lcb + 16 * (lp >> 1) + 8 * (lp & 1); };
  const int fa_off = tr_off(wm), fb_off0 = BAND + tr_off(wn * TJ), fb_off1 = BAND + tr_off(wn * TJ + 1);
  const int nb8 = g.blocks >> 3;
  const int sb = g.blocks % 8 == 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  long long u = rows::wg_start(g, sb);
  const long long uend = rows::wg_start(g, sb + 1);
  float* slot = g.partials + (size_t)sb * g.slots * WSLOT;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  R3_SETPRIO();
  for (; u < uend; slot += WSLOT) {
    const WgradProb& P = g.p[rows::wg_prob_of_unit(g, u)];
    const long long rel = u - P.unit0;
    const int lt = P.lt0 + (int)(rel / P.chunks), c0 = (int)(rel % P.chunks);
    const int c1 = (int)min((long long)P.chunks, c0 + (uend - u));
    u += c1 - c0;
    const int bx = lt % P.tk, by = lt / P.tk;
    const int n0 = by * TM, k0 = bx * TN;
    const int mbeg = c0 * WCH, mend = min(P.M, c1 * WCH), mlast = mend - 1;
    const int KT = (mend - mbeg + BKT - 1) / BKT;
    const bool aok = n0 + 8 * sch < P.N, bok = k0 + 8 * sch < P.K;             // (N, K multiples of 8: an octet is in or out)
    const unsigned acol = (unsigned)min(n0 + 8 * sch, P.N - 8), bcol = (unsigned)min(k0 + 8 * sch, P.K - 8);
    const unsigned lda = P.N, ldb = P.K;
    const float* const Ab = P.dY;
    const float* const Bb = P.X;
    const bool edge = n0 + TM > P.N || k0 + TN > P.K;             // (block-uniform)
    const bool sums = P.db != nullptr && bx == 0;
    const int* const grp_a = FORMS ? g.a_groups : nullptr;
    const int* const grp_b = FORMS ? g.b_groups : nullptr;
    const bool bnrelu = FORMS && g.scale != nullptr;
    float bsc[8], bsh[8], asum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsc[e] = 1.f, bsh[e] = 0.f, asum[e] = 0.f;
    if (bnrelu && bok) {
      const float4 s0 = *reinterpret_cast<const float4*>(g.scale + k0 + 8 * sch), s1 = *reinterpret_cast<const float4*>(g.scale + k0 + 8 * sch + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(g.shift + k0 + 8 * sch), h1 = *reinterpret_cast<const float4*>(g.shift + k0 + 8 * sch + 4);
      bsc[0] = s0.x, bsc[1] = s0.y, bsc[2] = s0.z, bsc[3] = s0.w, bsc[4] = s1.x, bsc[5] = s1.y, bsc[6] = s1.z, bsc[7] = s1.w;
      bsh[0] = h0.x, bsh[1] = h0.y, bsh[2] = h0.z, bsh[3] = h0.w, bsh[4] = h1.x, bsh[5] = h1.y, bsh[6] = h1.z, bsh[7] = h1.w;
    }

    float ra[2][8], rb[2][8];
    u32x4 pka[3], pkb[3];
    // byte offsets of this thread's octet from the tile's first row: a whole tile's loads share one scalar base per operand
    const unsigned voa = ((unsigned)srow * lda + acol) * 4u, vob = ((unsigned)srow * ldb + bcol) * 4u;
    auto put8 = [](float (&d)[8], const float4& v0, const float4& v1) __attribute__((always_inline)) {
      d[0] = v0.x, d[1] = v0.y, d[2] = v0.z, d[3] = v0.w, d[4] = v1.x, d[5] = v1.y, d[6] = v1.z, d[7] = v1.w;
    };
    auto gload = [&](auto set_c, int kt) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      const int mt = mbeg + min(kt, KT - 1) * BKT;
      int offa = 0, offb = 0;                          // stored row - product row of this tile (listed operands)
      if (FORMS) {
        const int gq = mt >> 5;
        if (grp_a) offa = (grp_a[gq] - gq) * 32;
        if (grp_b) offb = (grp_b[gq] - gq) * 32;
      }
      if (mt + BKT <= mend) {                          // (block-uniform) a whole tile: no row clamps
        const char* pa = reinterpret_cast<const char*>(Ab + (size_t)(mt + offa) * lda);
        const char* pb = reinterpret_cast<const char*>(Bb + (size_t)(mt + offb) * ldb);
        put8(ra[set], *reinterpret_cast<const float4*>(pa + voa), *reinterpret_cast<const float4*>(pa + voa + 16));
        put8(rb[set], *reinterpret_cast<const float4*>(pb + vob), *reinterpret_cast<const float4*>(pb + vob + 16));
      } else {                                         // rows past the end re-read the last row (zeroed in fixup)
        const int m = min(mt + srow, mlast);
        const float* pa = Ab + (size_t)(m + offa) * lda + acol;
        const float* pb = Bb + (size_t)(m + offb) * ldb + bcol;
        put8(ra[set], *reinterpret_cast<const float4*>(pa), *reinterpret_cast<const float4*>(pa + 4));
        put8(rb[set], *reinterpret_cast<const float4*>(pb), *reinterpret_cast<const float4*>(pb + 4));
      }
    };
    // what a register set still needs before it is split: producer, zeros past the edges, the column sums
    auto fixup = [&](auto set_c, int kt) __attribute__((always_inline)) {
      constexpr int set = decltype(set_c)::value;
      const int mt = mbeg + kt * BKT;
      const bool live = kt < KT;                       // (tiles past the end are split into the buffer nobody reads)
      if (bnrelu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) rb[set][q] = fmaxf(rb[set][q] * bsc[q] + bsh[q], 0.f);
      }
      if (edge || mt + BKT > mend) {
        const bool in = mt + srow < mend;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          ra[set][q] = (in && aok) ? ra[set][q] : 0.f;
          rb[set][q] = (in && bok) ? rb[set][q] : 0.f;
        }
      }
      if (sums && live) {
#pragma unroll
        for (int q = 0; q < 8; ++q) asum[q] += ra[set][q];
      }
    };
    auto side_chunk = [&](auto c_c, auto set_c, int buf) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value, o = c / 8, ch = c % 8, set = decltype(set_c)::value;
      auto run = [&](float (&v)[8], u32x4 (&pk)[3], char* d) __attribute__((always_inline)) {
        split_chunk<ch>(v, pk);
        if (ch == 3) *reinterpret_cast<u32x4*>(d) = pk[0];
        if (ch == 7) {
          *reinterpret_cast<u32x4*>(d + PLANE) = pk[1];
          *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pk[2];
        }
      };
      if constexpr (o == 0) run(ra[set], pka, lds3 + buf * BUF + wlds);
      else run(rb[set], pkb, lds3 + buf * BUF + BAND + wlds);
    };
    f32x16 hi[TI][TJ], lo[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) hi[i][j][e] = 0.f, lo[i][j][e] = 0.f;
    bf16x8 fa[2][TI][3], fb[2][TJ][3];
    constexpr int NF = 3 * (TI + TJ);
    // fragment f of a 16-deep step s16 (rows 16 s16 .. 16 s16 + 15 of the tile): two transposed reads, rows 8 h .. 8 h + 3 and
    // 8 h + 4 .. 8 h + 7 of the step
    auto frag_one = [&](auto f_c, auto st_c, int buf, int s16) __attribute__((always_inline)) {
      constexpr int f = decltype(f_c)::value, st = decltype(st_c)::value;
      constexpr int FPL[6] = {2, 0, 0, 2, 1, 1};
      constexpr bool isa = f < 6 ? (f % 2 == 0) : (f - 6 < 3 * (TI - 1));
      constexpr int tl = f < 6 ? 0 : (isa ? 1 + (f - 6) / 3 : 1 + (f - 6 - 3 * (TI - 1)) / 3);
      constexpr int pl = f < 6 ? FPL[f] : (isa ? (f - 6) % 3 : (f - 6 - 3 * (TI - 1)) % 3);
      const char* base = lds3 + buf * BUF + s16 * (16 * PROW) + pl * PLANE;
      if constexpr (isa) {
        fa[st][tl][pl] = tr_frag(base + fa_off, base + fa_off + 4 * PROW);
      } else {
        const int off = tl == 0 ? fb_off0 : fb_off1;
        fb[st][tl][pl] = tr_frag(base + off, base + off + 4 * PROW);
      }
    };
    constexpr int S = 12 * G, NR = (NF + R3_FPS - 1) / R3_FPS, SB = S - NR, NC = 16, SC = SB - 1;
    auto ktile = [&](auto par_c, int kt) __attribute__((always_inline)) {
      constexpr int P2 = decltype(par_c)::value;
      using SetN = std::integral_constant<int, P2 ^ 1>;
      fixup(SetN{}, kt + 1);
      static_for<S>([&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        constexpr int step = s / (6 * G), gi = (s % (6 * G)) / 6, q = s % 6;
        mfma_one<DUAL, q>(fa[step][gi / TJ], fb[step][gi % TJ], hi[gi / TJ][gi % TJ], lo[gi / TJ][gi % TJ]);
        static_for<NF>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          if constexpr (f * (6 * G) / NF == s) frag_one(f_c, C1{}, P2, 1);
        });
        static_for<NC>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          if constexpr (c * SC / NC == s) side_chunk(c_c, SetN{}, P2 ^ 1);
        });
        if constexpr (s == (NC - 1) * SC / NC) gload(SetN{}, kt + 3);
        if constexpr (s == SB - 1) __syncthreads();
        if constexpr (s >= SB) {
          static_for<NF>([&](auto f_c) {
            constexpr int f = decltype(f_c)::value;
            if constexpr (f / R3_FPS == s - SB) frag_one(f_c, C0{}, P2 ^ 1, 0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    gload(C0{}, 0);
    gload(C1{}, 1);
    fixup(C0{}, 0);
    static_for<NC>([&](auto c_c) { side_chunk(c_c, C0{}, 0); });
    gload(C0{}, 2);
    __syncthreads();
    static_for<NF>([&](auto f_c) { frag_one(f_c, C0{}, 0, 0); });
    for (int kt = 0; kt < KT; kt += 2) {
      ktile(C0{}, kt);
      if (kt + 1 < KT) ktile(C1{}, kt + 1);
    }
    // ---- the partial tile (whole TM x TN, edges included: the reduction stores what is inside)
    __syncthreads();
    if (sums) {
      float* red = reinterpret_cast<float*>(lds3);     // [32 staged rows][TM]; the tile buffers are free now
#pragma unroll
      for (int e = 0; e < 8; ++e) red[srow * TM + 8 * sch + e] = asum[e];
      __syncthreads();
      if (tid < TM) {
        float t = 0.f;
#pragma unroll 8
        for (int q = 0; q < BKT; ++q) t += red[q * TM + tid];
        if (!g.direct) slot[TM * TN + tid] = t;
        else if (n0 + tid < P.N) P.db[n0 + tid] = t;
      }
      __syncthreads();
    }
    (void)srow_u;
    if (g.direct) {                                    // this block summed the tile over ALL rows: the result, not a partial
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = n0 + (wm * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int cc = k0 + (wn * TJ + j) * 32 + r;
            if (row < P.N && cc < P.K) R3_STORE(&P.dW[(size_t)row * P.K + cc], hi[i][j][e] + lo[i][j][e]);
          }
      continue;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (wm * TI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          R3_STORE(&slot[row * TN + (wn * TJ + j) * 32 + r], hi[i][j][e] + lo[i][j][e]);
        }
  }
}


// ---------------------------------------------------------------------------------------------
// The fused NT GEMMs of the patch embedder and of the set-abstraction MLPs (gemm.hip gemm_nt_kernel's contracts: NtArgs,
// PRO_BNRELU producer with optional 32-row group gather, the group-wise epilogues) on the exact-split arithmetic and
// on gemm3_kernel's pipeline: C[M,N] = epi(pro(A[M,K]) . B[N,K]^T), K % 32 == 0.  Block tile 128 x 128, eight waves
// of 32 x 64: a wave's 32 rows are ONE group of a point cloud (one 32-row MFMA tile), so the group bias is the
// accumulators' start value, the group max is a max over a lane's 16 registers and its partner lane (lane ^ 32), and
// the BatchNorm column statistics meet across the four waves of a column in LDS.
//   PRO_BNRELU            a := relu(a * scale[k] + shift[k]) on the staged A octet right before it is split
//   EPI_BIAS              C = acc + bias
//   EPI_GROUPBIAS_STATS   C = acc + gbias[m / 32]; per-column sum / sum of squares -> stats (atomics per XCD slot, or
//                         plain per-tile-row partials in deterministic mode)
//   EPI_STATS             C = acc (+ bias) with the same statistics
//   EPI_STORE_GROUPMAX    C = acc + bias stored; gmax / garg = max over each group's rows and the first row attaining it
//   EPI_GROUPMAX          the same without the store
//   EPI_GROUP_SCATTER     C[c_groups[m / 32] * 32 + m % 32] = acc + gbias[m / 32] (gbias nullable)
template <int PRO, int EPI>
__global__ __launch_bounds__(512) void conv3_kernel(const NtArgs p) {
  constexpr int TI = 1, TJ = 2, WM = 4, WN = 2, BM = 128, BN = 128, ROWB = 80, BKT = 32;
  constexpr int PLANE = (BM + BN) * ROWB, BUF = 3 * PLANE;
  constexpr int G = TI * TJ;
  constexpr bool DUAL = true;
  extern __shared__ __attribute__((aligned(16))) char lds3[];   // [2][BUF] tile buffers, then [WM][2][BN] floats of statistics
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  // Persistent blocks (as gemm_nt_kernel): block (xcd, slot) walks tiles slot, slot + nslots, ... of its XCD's chunk and
  // the k-tiles of consecutive output tiles form ONE stream through the pipeline -- the loads run three positions
  // ahead across the tile boundary, so a boundary costs the epilogue only (no exposed first loads: ~3 of the ~6 us a
  // tile pays on top of its k-loop).  The host launches one residency of blocks when K / 32 is even and >= 4 (the
  // stream keeps its buffer / register-set parity across tiles), else one block per tile.
  const int chunk = (p.tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, nslots = gridDim.x >> 3;
  int slot = blockIdx.x >> 3;
  auto tile_of = [&](int sl) {
    const int t = xcd * chunk + sl;
    return (sl < chunk && t < p.tiles) ? t : -1;
  };
  int tile = tile_of(slot);
  if (tile < 0) return;
  const int KT = p.K / BKT;
  // this thread's octet of A and of B (one each: 128 rows x 4 octets = 512): conflict-free LDS store map of gemm3_kernel
  const int orow = ((tid >> 5) << 3) + (((tid >> 2) & 1) << 2) + ((tid >> 3) & 3), ocol = tid & 3;
  const int alds = orow * ROWB + ocol * 16, blds = (BM + orow) * ROWB + ocol * 16;
  unsigned aoff[2], boff[2];                           // [0] the tile being multiplied, [1] the next one of this block
  auto offsets_of = [&](int t, unsigned& ao, unsigned& bo) __attribute__((always_inline)) {
    const int tm0 = (t / p.tiles_n) * BM, tn0 = (t % p.tiles_n) * BN;
    int am = min(tm0 + orow, M - 1);
    if (p.a_groups) am = p.a_groups[am >> 5] * 32 + (am & 31);     // whole 32-row groups gathered
    ao = ((unsigned)am * (unsigned)p.lda + ocol * 8) * 4u;
    bo = ((unsigned)min(tn0 + orow, N - 1) * (unsigned)p.ldb + ocol * 8) * 4u;
  };
  offsets_of(tile, aoff[0], boff[0]);
  int next = tile_of(slot + nslots);
  aoff[1] = aoff[0], boff[1] = boff[0];
  if (next >= 0) offsets_of(next, aoff[1], boff[1]);
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  float ra[2][8], rb[2][8], rs[2][8], rh[2][8];         // staged A / B octets; the producer's scale / shift of the A octet's k
  u32x4 pka[3], pkb[3];
  // position `pos` of the stream: k-tile pos of this tile, or k-tile pos - KT of the next one (past the last tile: the
  // last k-tile again, split into the buffer nobody reads)
  auto gload = [&](auto set_c, int pos) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    const bool nx = pos >= KT && next >= 0;
    const size_t kb = (size_t)(nx ? pos - KT : min(pos, KT - 1)) * BKT * 4;
    const unsigned ao = nx ? aoff[1] : aoff[0], bo = nx ? boff[1] : boff[0];
    const float4 a0 = *reinterpret_cast<const float4*>(Ab + kb + ao);
    const float4 a1 = *reinterpret_cast<const float4*>(Ab + kb + ao + 16);
    const float4 b0 = *reinterpret_cast<const float4*>(Bb + kb + bo);
    const float4 b1 = *reinterpret_cast<const float4*>(Bb + kb + bo + 16);
    ra[set][0] = a0.x, ra[set][1] = a0.y, ra[set][2] = a0.z, ra[set][3] = a0.w;
    ra[set][4] = a1.x, ra[set][5] = a1.y, ra[set][6] = a1.z, ra[set][7] = a1.w;
    rb[set][0] = b0.x, rb[set][1] = b0.y, rb[set][2] = b0.z, rb[set][3] = b0.w;
    rb[set][4] = b1.x, rb[set][5] = b1.y, rb[set][6] = b1.z, rb[set][7] = b1.w;
    if constexpr (PRO == PRO_BNRELU) {
      const char* sc = reinterpret_cast<const char*>(p.pro_scale) + kb + ocol * 32;
      const char* sh = reinterpret_cast<const char*>(p.pro_shift) + kb + ocol * 32;
      const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 16);
      const float4 h0 = *reinterpret_cast<const float4*>(sh), h1 = *reinterpret_cast<const float4*>(sh + 16);
      rs[set][0] = s0.x, rs[set][1] = s0.y, rs[set][2] = s0.z, rs[set][3] = s0.w;
      rs[set][4] = s1.x, rs[set][5] = s1.y, rs[set][6] = s1.z, rs[set][7] = s1.w;
      rh[set][0] = h0.x, rh[set][1] = h0.y, rh[set][2] = h0.z, rh[set][3] = h0.w;
      rh[set][4] = h1.x, rh[set][5] = h1.y, rh[set][6] = h1.z, rh[set][7] = h1.w;
    }
  };
  auto side_chunk = [&](auto c_c, auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int c = decltype(c_c)::value, o = c / 8, ch = c % 8, set = decltype(set_c)::value;
    auto run = [&](float (&v)[8], u32x4 (&pk)[3], char* d) __attribute__((always_inline)) {
      split_chunk<ch>(v, pk);
      if (ch == 3) *reinterpret_cast<u32x4*>(d) = pk[0];
      if (ch == 7) {
        *reinterpret_cast<u32x4*>(d + PLANE) = pk[1];
        *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pk[2];
      }
    };
    if constexpr (o == 0) {
      if constexpr (PRO == PRO_BNRELU && ch == 0) {        // the previous layer's BatchNorm + ReLU, never stored
#pragma unroll
        for (int j = 0; j < 8; ++j) ra[set][j] = fmaxf(ra[set][j] * rs[set][j] + rh[set][j], 0.f);
      }
      run(ra[set], pka, lds3 + buf * BUF + alds);
    } else {
      run(rb[set], pkb, lds3 + buf * BUF + blds);
    }
  };
  f32x16 hi[TI][TJ], lo[TI][TJ];
  bf16x8 fa[2][TI][3], fb[2][TJ][3];
  const int fa_off = (wm * TI * 32 + r) * ROWB + 16 * h, fb_off = (BM + wn * TJ * 32 + r) * ROWB + 16 * h;
  constexpr int NF = 3 * (TI + TJ);
  auto frag_one = [&](auto f_c, auto st_c, int buf, int s16) __attribute__((always_inline)) {
    constexpr int f = decltype(f_c)::value, st = decltype(st_c)::value;
    constexpr int FPL[6] = {2, 0, 0, 2, 1, 1};
    constexpr bool isa = f < 6 ? (f % 2 == 0) : (f - 6 < 3 * (TI - 1));
    constexpr int tl = f < 6 ? 0 : (isa ? 1 + (f - 6) / 3 : 1 + (f - 6 - 3 * (TI - 1)) / 3);
    constexpr int pl = f < 6 ? FPL[f] : (isa ? (f - 6) % 3 : (f - 6 - 3 * (TI - 1)) % 3);
    const char* base = lds3 + buf * BUF + s16 * 32 + pl * PLANE;
    if constexpr (isa) fa[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + fa_off + tl * 32 * ROWB);
    else fb[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + fb_off + tl * 32 * ROWB);
  };
  constexpr int S = 12 * G, NR = (NF + R3_FPS - 1) / R3_FPS, SB = S - NR, NC = 16, SC = SB - 1;
  auto ktile = [&](auto par_c, int kt) __attribute__((always_inline)) {
    constexpr int P2 = decltype(par_c)::value;
    using SetN = std::integral_constant<int, P2 ^ 1>;
    static_for<S>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int step = s / (6 * G), gi = (s % (6 * G)) / 6, q = s % 6;
      mfma_one<DUAL, q>(fa[step][gi / TJ], fb[step][gi % TJ], hi[gi / TJ][gi % TJ], lo[gi / TJ][gi % TJ]);
      static_for<NF>([&](auto f_c) {
        constexpr int f = decltype(f_c)::value;
        if constexpr (f * (6 * G) / NF == s) frag_one(f_c, C1{}, P2, 1);
      });
      static_for<NC>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        if constexpr (c * SC / NC == s) side_chunk(c_c, SetN{}, P2 ^ 1);
      });
      if constexpr (s == (NC - 1) * SC / NC) gload(SetN{}, kt + 3);
      if constexpr (s == SB - 1) __syncthreads();
      if constexpr (s >= SB) {
        static_for<NF>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          if constexpr (f / R3_FPS == s - SB) frag_one(f_c, C0{}, P2 ^ 1, 0);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  R3_SETPRIO();
  gload(C0{}, 0);
  gload(C1{}, 1);
  static_for<NC>([&](auto c_c) { side_chunk(c_c, C0{}, 0); });
  gload(C0{}, 2);
  __syncthreads();
  static_for<NF>([&](auto f_c) { frag_one(f_c, C0{}, 0, 0); });

  for (;;) {
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      float init = 0.f;
      if (EPI == EPI_GROUPBIAS_STATS || EPI == EPI_GROUP_SCATTER) {
        const int gr = m0 + wm * 32, gc = n0 + (wn * TJ + j) * 32 + r;
        if (gr < M && gc < N && (EPI == EPI_GROUPBIAS_STATS || p.gbias)) init = p.gbias[(size_t)(gr >> 5) * N + gc];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) hi[0][j][e] = init, lo[0][j][e] = 0.f;
    }
    for (int kt = 0; kt < KT; kt += 2) {
      ktile(C0{}, kt);
      if (kt + 1 < KT) ktile(C1{}, kt + 1);
    }
    // ---- epilogue: this wave's 32 rows are one group; a lane holds 16 rows of one column, its partner (lane ^ 32) the
    // rest.  (LDS already holds the next tile's first k-tile, its fragments are in registers, its loads are in flight.)
    float csum[TJ], csq[TJ];
    const int rbase = m0 + wm * 32;
    const bool rows_in = rbase < M;                     // (M % 32 == 0 for the group epilogues: a group is inside or outside)
    const unsigned ldc = (unsigned)p.ldc;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      csum[j] = 0.f, csq[j] = 0.f;
      const int col = n0 + (wn * TJ + j) * 32 + r;
      const bool colok = col < N;
      const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
      const float add = (EPI == EPI_GROUPBIAS_STATS || EPI == EPI_GROUP_SCATTER) ? 0.f : bv;
      float vmax = -__builtin_huge_valf();
      int amax = 0;
      float* cbase = nullptr;
      if (EPI != EPI_GROUPMAX) {
        int crow = rbase;
        if (EPI == EPI_GROUP_SCATTER && rows_in) crow = p.c_groups[rbase >> 5] * 32;
        cbase = p.C + (size_t)(crow + 4 * h) * ldc + col;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int lr = (e & 3) + 8 * (e >> 2) + 4 * h;
        const bool in = colok && rbase + lr < M;
        const float v = (hi[0][j][e] + lo[0][j][e]) + add;
        if ((EPI == EPI_GROUPBIAS_STATS || EPI == EPI_STATS) && rbase + lr < M) {
          csum[j] += v;
          csq[j] += v * v;
        }
        if (EPI == EPI_GROUPMAX || EPI == EPI_STORE_GROUPMAX) {
          if (v > vmax) vmax = v, amax = lr;            // e ascending => lr ascending within this half
        }
        if (EPI != EPI_GROUPMAX && in) R3_STORE(&cbase[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc], v);
      }
      if (EPI == EPI_GROUPMAX || EPI == EPI_STORE_GROUPMAX) {
        const float ov = __shfl_xor(vmax, 32, kWave);
        const int oa = __shfl_xor(amax, 32, kWave);
        if ((ov > vmax) || (ov == vmax && oa < amax)) vmax = ov, amax = oa;
        if (h == 0 && colok && rows_in) {
          p.gmax[(size_t)(rbase >> 5) * N + col] = vmax;
          p.garg[(size_t)(rbase >> 5) * N + col] = (unsigned char)amax;
        }
      }
    }
    if (EPI == EPI_GROUPBIAS_STATS || EPI == EPI_STATS) {
      // per-tile column sums: one LDS slot per row of waves, added in wave order, then one atomic per column into the
      // partial buffer of this block's XCD slot -- or, in deterministic mode, a plain store into row (tile row)
      float* red = reinterpret_cast<float*>(lds3 + 2 * BUF);          // [WM][2][BN], behind the tile buffers
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float sum = csum[j] + __shfl_xor(csum[j], 32, kWave);
        const float sq = csq[j] + __shfl_xor(csq[j], 32, kWave);
        if (h == 0) {
          red[(wm * 2 + 0) * BN + (wn * TJ + j) * 32 + r] = sum;
          red[(wm * 2 + 1) * BN + (wn * TJ + j) * 32 + r] = sq;
        }
      }
      __syncthreads();
      float* dst = p.stats_det ? p.stats_det + (size_t)(m0 / BM) * 2 * N : p.stats + (size_t)(blockIdx.x & 7) * 2 * N;
      if (tid < 2 * BN) {
        const int half = tid / BN, cc = tid - half * BN;
        if (n0 + cc < N) {
          float t = red[half * BN + cc];
#pragma unroll
          for (int k = 1; k < WM; ++k) t += red[(k * 2 + half) * BN + cc];
          if (p.stats_det) dst[half * N + n0 + cc] = t;
          else atomicAdd(dst + half * N + n0 + cc, t);
        }
      }
      __syncthreads();                                  // red is written again by the next tile's epilogue
    }
    if (next < 0) break;
    slot += nslots;
    tile = next;
    aoff[0] = aoff[1], boff[0] = boff[1];
    next = tile_of(slot + nslots);
    if (next >= 0) offsets_of(next, aoff[1], boff[1]);
  }
}

}  // namespace rows3
}  // namespace pdae
