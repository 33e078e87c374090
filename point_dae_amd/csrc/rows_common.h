// rows_common.h -- argument structs, the grouped weight-gradient work layout and its ordered reduction,
// shared by the fp32-MFMA row GEMMs (rows_gemm.hip) and the exact-split bf16 ones (rows3_gemm.hip).
#pragma once
#include "common.h"

namespace pdae {
namespace rows {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LD = BK + 4;   // LDS rows of 36 floats: ds_read_b128 of 16 lanes hits 64 banks

enum { EPI_STORE = 0, EPI_BIAS_RELU = 1, EPI_BIAS_GELU2 = 2, EPI_MUL_GELUGRAD = 3, EPI_MUL_POS = 4,
       // (exact-split family, [K,N] weights, 128 x 128 tiles: pdae_rows_gemm_bnrelu_stats) the data gradient that flows into a
       // BatchNorm + ReLU: C = relu'(bn(X)) ? acc : 0 stored, and the two column sums BatchNorm's backward needs
       // (sum t, sum t xhat) as one partial row per 128-row band of the product
       EPI_BNRELU_STATS = 5 };

struct Args {
  int M, N, K;
  const float* A;
  int lda;
  const float* B;
  int ldb;
  float* C;
  int ldc;
  float* Z;           // EPI_BIAS_GELU2: GELU'(z) written;  EPI_MUL_GELUGRAD: read;  EPI_MUL_POS: the ReLU output
                      // whose sign masks the result (leading dimension ldc)
  const float* bias;  // [N] or null
  int tiles_n, tiles;
  int kchunk;         // reduction range of one split (blockIdx.y), a multiple of BK
  long long slab;     // elements between the C slabs of consecutive splits
  int stream_blocks;  // > 0: stream-K over (tile, k-tile) units on this many blocks, `slabs` C slabs
  int slabs;
  long long strideA, strideB, strideC;   // batched launch: element strides between the problems of blockIdx.z
  int symmetric;      // fp32-input kernel, square tiles, A == B (Gram matrices): `tiles` counts the tiles on and above the
                      // diagonal; a block also stores its tile transposed below it
  // EPI_BNRELU_STATS: Z = X, the BatchNorm's INPUT rows (leading dimension ldc; row m of the product = row
  // z_groups[m / 32] * 32 + m % 32 of X when a list is given), its folded affine (scale, shift) and statistics, and
  // stats_part [ceil(M / 128)][2][N]
  const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
  const int* z_groups;
  float* stats_part;
};

constexpr int WG_MAX = 48;               // layers per launch: 12 Transformer blocks x 4 (the by-value struct is 3.1 KB)
constexpr int TBK = 16;                 // rows per LDS slab
constexpr int WCH = 32;                 // rows per work unit
constexpr int WTM = 128;                // output tile rows (n of dY)
// output tile columns (k of X): 128 (4 waves, any shape), 256 or 384 (8 waves: the Transformer blocks' K = 384 / 1536 --
// a unit then moves 64 KB for 3.1 MFLOP instead of 32 KB for 1.05, and the stack-level launches stream their operands
// from HBM: 48 instead of 32 FLOP per byte; 256 for K = 256 / 512 / 1024: 43 FLOP per byte)
constexpr int wslot(int TN) { return WTM * TN + WTM; }   // floats per partial: the tile + the column sums of its dY band
struct WgradProb {
  const float* dY;    // [M, N]
  const float* X;     // [M, K]
  float* dW;          // [N, K]
  float* db;          // [N] or null: column sums of dY
  int N, K;
  int tk;             // tiles along K
  int tile0;          // first tile of this problem in the group
  int M, chunks;      // rows of this problem, 32-row chunks per tile
  int lt0;            // this launch covers the problem's tiles lt0 .. (a group split over two launches; else 0)
  long long unit0;    // first unit of this problem
};
struct WgradArgs {
  int nprob, tiles, blocks, slots;              // grid; slots per block
  int direct;                                   // 1: a block owns whole tiles and stores them (and the bias sums) itself: no partials
  long long units;
  float* partials;                              // [blocks][slots][wslot(TN)]
  // the embedder's forms (patch_embed.py; one problem per launch): whole 32-row groups gathered on either operand (row m
  // of the product = row groups[m / 32] * 32 + m % 32 of the stored matrix), and X := relu(X * scale[k] + shift[k])
  // while it is staged (BatchNorm + ReLU recomputed instead of stored); all null for the Transformer blocks.  (Here and
  // not per problem: 48 problems x 4 pointers would push the by-value argument past the 4 KB kernarg limit.)
  const int* a_groups;
  const int* b_groups;
  const float* scale;
  const float* shift;
  WgradProb p[WG_MAX];
};

__device__ __forceinline__ long long wg_start(const WgradArgs& g, int b) {
  return ((long long)b * g.units) / g.blocks;
}

// the problem a unit / a tile belongs to (scalar work: u and tile are block-uniform)
__device__ __forceinline__ int wg_prob_of_unit(const WgradArgs& g, long long u) {
  int pi = 0;
  for (int q = 1; q < g.nprob; ++q)
    if (u >= g.p[q].unit0) pi = q;
  return pi;
}
__device__ __forceinline__ int wg_prob_of_tile(const WgradArgs& g, int tile) {
  int pi = 0;
  for (int q = 1; q < g.nprob; ++q)
    if (tile >= g.p[q].tile0) pi = q;
  return pi;
}
__device__ __forceinline__ int wg_tile_of_unit(const WgradArgs& g, long long u) {
  const WgradProb& P = g.p[wg_prob_of_unit(g, u)];
  return P.tile0 + (int)((u - P.unit0) / P.chunks);
}

// dW tile = sum of its partials in a fixed order (=> bit-identical run to run).  A block owns
// 256 / PL float4 of a tile and PL "partial lanes": lane l adds partials b0 + l, b0 + l + PL, ...
// (eight loads in flight), the lanes are then added in lane order through LDS.  PL = 1 for the
// Transformer blocks' groups (<= 16 partials per tile); a narrow weight with a long reduction has
// hundreds (the embedder's first conv: ONE tile, 512 partials -- a single lane walking them took
// 121 us, PL = 8 takes 15).  Elements outside the weight are skipped.  The blocks with part == 0
// also reduce the bias-gradient column sums.
// float4 per thread of the reduction: 8 where a tile has a few partials (one lane: the stacks' launches, 96 -> 38 us),
// 1 where it has hundreds (the lane-parallel forms want the blocks)
constexpr int wru(int PL) { return PL == 1 ? 8 : 1; }
template <int PL, int TN>
__device__ __forceinline__ void wgrad_reduce_body(const WgradArgs& g, int block) {
  constexpr int WRU = wru(PL);
  constexpr int EPB = 256 / PL;                      // threads per partial lane
  constexpr int PARTS = WTM * TN / 4 / (EPB * WRU);   // blocks per tile
  constexpr int WSLOT = wslot(TN);
  static_assert(EPB >= WTM / 4 && (WTM * TN / 4) % (EPB * WRU) == 0, "a tile divides into blocks; its bias sums fit one");
  __shared__ float4 red[PL > 1 ? PL : 1][PL > 1 ? WRU : 1][EPB];
  __shared__ float4 redb[PL > 1 ? PL : 1][WTM / 4];
  const int tile = block / PARTS, part = block % PARTS;
  const WgradProb& P = g.p[wg_prob_of_tile(g, tile)];
  const int lt = tile - P.tile0;                     // within this launch's share of the problem
  const int bx = (lt + P.lt0) % P.tk, by = (lt + P.lt0) / P.tk;
  const int n0 = by * WTM, k0 = bx * TN;
  // blocks whose ranges meet this tile's units
  const long long u0 = P.unit0 + (long long)lt * P.chunks, u1 = u0 + P.chunks - 1;
  const int b0 = (int)(((u0 + 1) * g.blocks - 1) / g.units), b1 = (int)(((u1 + 1) * g.blocks - 1) / g.units);
  const int tid = threadIdx.x, pl = PL == 1 ? 0 : tid / EPB, el = tid % EPB;
  // this thread's WRU float4 of the WTM x TN tile; the locating arithmetic of a partial (which slot of block b holds
  // this tile: 64-bit divisions and a walk over the problems) is done once per partial and block, not per element
  int e[WRU];
  bool valid[WRU];
  bool any = false;
#pragma unroll
  for (int u = 0; u < WRU; ++u) {
    e[u] = ((part * WRU + u) * EPB + el) * 4;
    valid[u] = n0 + e[u] / TN < P.N && k0 + e[u] % TN < P.K;
    any |= valid[u];
  }
  const bool bias = part == 0 && bx == 0 && P.db && el < WTM / 4 && n0 + el * 4 < P.N;
  float4 s[WRU], bs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < WRU; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  // Which slot of block b holds this tile: the (tile - first tile of b's range)-th.  The range of every block b0 < b <= b1
  // STARTS inside this tile (ranges are contiguous and b0 holds the tile's first unit), so the tile is their first: slot
  // 0.  Only b0 may have come from an earlier tile -- one locating computation (64-bit divisions, a walk over the
  // problems) per reduction block instead of one per partial and thread.
  const int slot_b0 = tile - wg_tile_of_unit(g, wg_start(g, b0));
  for (int b = b0 + pl; b <= b1; b += PL) {          // lane pl adds partials b0 + pl, b0 + pl + PL, ... in that order
    const float* src = g.partials + ((size_t)b * g.slots + (b == b0 ? slot_b0 : 0)) * WSLOT;
    float4 v[WRU], w = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < WRU; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid[u]) v[u] = *reinterpret_cast<const float4*>(src + e[u]);
    }
    if (bias) w = *reinterpret_cast<const float4*>(src + WTM * TN + el * 4);
#pragma unroll
    for (int u = 0; u < WRU; ++u) s[u].x += v[u].x, s[u].y += v[u].y, s[u].z += v[u].z, s[u].w += v[u].w;
    bs.x += w.x, bs.y += w.y, bs.z += w.z, bs.w += w.w;
  }
  (void)any;
  if (PL > 1) {
#pragma unroll
    for (int u = 0; u < WRU; ++u) red[pl][u][el] = s[u];
    if (el < WTM / 4) redb[pl][el] = bs;
    __syncthreads();
    if (pl != 0) return;
#pragma unroll
    for (int q = 1; q < PL; ++q) {
#pragma unroll
      for (int u = 0; u < WRU; ++u) {
        const float4 t = red[q][u][el];
        s[u].x += t.x, s[u].y += t.y, s[u].z += t.z, s[u].w += t.w;
      }
      if (el < WTM / 4) {
        const float4 t = redb[q][el];
        bs.x += t.x, bs.y += t.y, bs.z += t.z, bs.w += t.w;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < WRU; ++u)
    if (valid[u]) *reinterpret_cast<float4*>(P.dW + (size_t)(n0 + e[u] / TN) * P.K + k0 + e[u] % TN) = s[u];
  if (bias) *reinterpret_cast<float4*>(P.db + n0 + el * 4) = bs;
}

template <int PL, int TN>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs g) {
  wgrad_reduce_body<PL, TN>(g, blockIdx.x);
}

}  // namespace rows

// ---- the exact-split bf16 instantiations (rows3_gemm.hip), dispatched to by the C entries of rows_gemm.hip
namespace rows3 {
struct Cfg3 {
  int ti, tj, wm, wn, ks;
};
constexpr int NCFG3 = 4;
constexpr int CFG3_BASE = 16;            // plan / pdae_rows_gemm `cfg` values CFG3_BASE + i select tile shape i of this family
extern const Cfg3 kCfg3[NCFG3];
size_t lds_bytes3(const Cfg3& c);
void launch_gemm3(rows::Args& a, int cfg, bool w_kn, int epi, int splits, int stream_blocks, hipStream_t s);
void launch_wgrad3(const rows::WgradArgs& g, int tn, int pl, hipStream_t s);
}  // namespace rows3
// the arithmetic in force for the forward / data-gradient half of the family (rows_gemm.hip)
int gemm_arith_rows();
}  // namespace pdae
