// rows_gemm.hip -- the dense layers of the Transformer blocks on the gfx950 matrix cores.
//
// The reference runs qkv / proj (models/PointCAE_transformer.py:113-137), fc1 / fc2 (:94-110),
// pos_embed (:329-333) and increase_dim (:653-658) through cuBLAS, with bias, GELU and their
// backward twins as separate passes.  The shapes are SMALL for a 256-CU chip: M = B * T rows
// with T = 13..32 visible tokens (M = 1664..4096) in the encoder and M = 8192 in the decoder,
// N, K in {384, 1152, 1536}: 5-25 us of MFMA time per GEMM, where a tile grid that does not
// divide the chip and the fixed cost of a block's first load / last store decide the result.
// Hence a family of tile shapes (64x64 .. 128x192, always 4 waves so the four SIMDs of a CU
// carry equal work), several independent blocks resident per CU (one block's prologue and
// epilogue under another's MFMAs), optional split-K into slabs that the consuming LayerNorm
// kernel adds up, and a per-shape plan (plan_rows) that picks tile and split by the number of
// MFMA rounds the grid costs.  fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32, the same
// arithmetic class as the reference's.
//
//   rows_gemm_kernel   C[M,N] = epi(A[M,K] . op(B));  op(B) = B[N,K]^T (a Linear's forward) or
//                      B[K,N] (the same weight as the data-gradient operand, no transposed copy)
//     epilogues: store (+bias) | bias+ReLU | z = acc+bias: GELU(z) -> C and GELU'(z) -> Z (fc1 +
//                act: the activation and the factor its backward needs, one erf for both) |
//                C = acc * Z (backward of the former: dz comes out of the dh GEMM as a multiply;
//                evaluating GELU' there cost 27 us per decoder block)
//   wgrad_kernel       dW_p[N_p,K_p] = dY_p[M,N_p]^T . X_p[M,K_p] for a GROUP of layers in one
//                      launch (a block's qkv, proj, fc1, fc2: 108 tiles instead of four grids of
//                      9-36), reduction over M split into slabs, + column sums of dY (bias grad)
//   wgrad_reduce       slabs -> gradients in fixed split order: no atomics anywhere, results are
//                      bit-identical run to run
#include <type_traits>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rows_common.h"

namespace pdae {
namespace rows {




// Block tile (32 TI WM) x (32 TJ WN), WM x WN waves, every wave TI x TJ MFMA tiles of 32x32.
// k permutation inside an 8-deep slab as in gemm.hip: lane (r = l & 31, h = l >> 5) supplies
// k = 8s + 4h + t to MFMA t, so one ds_read_b128 of a K-contiguous operand row feeds four MFMAs.
// BKN: B is stored [K][N] (N contiguous): staged as [k][n] rows, fragments by ds_read_b32
// (lanes r -> consecutive banks).
// Pipeline: global -> registers -> LDS (two buffers), one barrier per 32-deep k-tile.  A k-tile is
// 1-4 k MFMA cycles per wave, an L2 round trip under load ~2 k: TWO k-tiles of loads are in flight
// in two register sets (one on the 128x192 tile, whose accumulators leave no room); the main loop
// is branch-free (the last iterations re-load the last tile instead of testing for the end), so
// the loads and LDS stores of the next tiles interleave with the MFMAs of this one.
template <int TI, int TJ, int WM, int WN, bool BKN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2)))
void rows_gemm_kernel(const Args p) {
  constexpr int BM = 32 * TI * WM, BN = 32 * TJ * WN, NT = 64 * WM * WN;
  constexpr int LA = (BM * 8) / NT, LB = (BN * 8) / NT;   // float4 per thread and k-tile
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "every thread stages LA + LB float4 per k-tile");
  constexpr int ASZ = BM * LD, BSZ = BKN ? BK * BN : BN * LD;
  constexpr int BQ = BN / 4;                               // float4 per staged [k][n] row
  constexpr int PF = TI * TJ <= 4 ? 2 : 1;                 // k-tiles of global loads in flight (register sets)
  extern __shared__ float lds[];                           // [2][ASZ + BSZ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int scol = (tid & 7) * 4;
  const int M = p.M, N = p.N;
  // Work of this block.  Plain mode: ONE tile (XCD-aware order: blocks b, b + 8, ... share an L2;
  // each XCD owns a contiguous chunk of tiles, n fastest, so the tiles_n re-reads of an A row band
  // hit that L2) and the k range of split blockIdx.y.  Stream mode (narrow outputs with a long
  // reduction: proj, fc2, the fc1 / qkv data gradients): (tile, k-tile) units in tile-major order,
  // dealt in EQUAL contiguous ranges to a grid that is resident at once; a range crosses tile
  // boundaries, the piece of tile t that is the q-th one in unit order goes to slab q of the
  // output, the block that ends a tile zero-fills the slabs the tile did not need, and the
  // consumer (a LayerNorm kernel) adds the slabs: no block carries more MFMAs than another, where
  // 276 or 414 equal tiles leave a quarter of the 256 CUs with twice the work.
  const int KTA = (p.K + BK - 1) / BK;
  const long long units = (long long)p.tiles * KTA;
  long long u = 0, uend = 1;
  int sb = 0;
  if (p.stream_blocks) {
    const int P = p.stream_blocks;
    sb = (int)(blockIdx.x & 7) * (P >> 3) + (int)(blockIdx.x >> 3);   // an XCD's blocks take consecutive ranges
    u = sb * units / P, uend = (sb + 1) * units / P;
    if (u >= uend) return;
  } else {
    const int chunk = (p.tiles + 7) >> 3;
    const int slot = blockIdx.x >> 3;
    if (slot >= chunk || (int)(blockIdx.x & 7) * chunk + slot >= p.tiles) return;
  }
  for (;;) {
  int tile, kbeg, kend, piece = 0;
  bool tile_ends = false;
  if (p.stream_blocks) {
    tile = (int)(u / KTA);
    const int k0 = (int)(u % KTA), k1 = (int)min((long long)KTA, k0 + (uend - u));
    u += k1 - k0;
    kbeg = k0 * BK, kend = min(p.K, k1 * BK);
    tile_ends = k1 == KTA;
    // blocks before this one that hold a piece of the tile: the tile's first unit lies in block
    // floor(((u_first + 1) P - 1) / units)
    const long long uf = (long long)tile * KTA;
    piece = sb - (int)(((uf + 1) * p.stream_blocks - 1) / units);
  } else {
    tile = (int)(blockIdx.x & 7) * ((p.tiles + 7) >> 3) + (int)(blockIdx.x >> 3);
    kbeg = blockIdx.y * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    piece = blockIdx.y;
  }
  int ti_ = tile / p.tiles_n, tj_ = tile % p.tiles_n;
  if (p.symmetric) {                                          // the tile-th tile on or above the diagonal, row by row
    int left = tile;
    ti_ = 0;
    while (left >= p.tiles_n - ti_) left -= p.tiles_n - ti_, ++ti_;
    tj_ = ti_ + left;
  }
  const int m0 = ti_ * BM, n0 = tj_ * BN;
  // 32-bit BYTE offsets from the (uniform) operand bases: one VGPR per staged row and the
  // scalar-base + vector-offset form of global_load (the launcher checks the operands are < 4 GB);
  // rows past the matrix edge are clamped (their products are never stored)
  unsigned aoff[LA], boff[LB];
  bool bcol_ok[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i)
    aoff[i] = ((unsigned)min(m0 + ((tid + i * NT) >> 3), M - 1) * (unsigned)p.lda + scol) * 4u;
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    if (BKN) {
      const int c = n0 + ((tid + i * NT) % BQ) * 4;
      bcol_ok[i] = c < N;
      boff[i] = ((unsigned)((tid + i * NT) / BQ) * (unsigned)p.ldb + (unsigned)min(c, N - 4)) * 4u;
    } else {
      bcol_ok[i] = true;
      boff[i] = ((unsigned)min(n0 + ((tid + i * NT) >> 3), N - 1) * (unsigned)p.ldb + scol) * 4u;
    }
  }
  const char* Ab = reinterpret_cast<const char*>(p.A + (size_t)blockIdx.z * p.strideA);
  const char* Bb = reinterpret_cast<const char*>(p.B + (size_t)blockIdx.z * p.strideB);
  const size_t bstep = BKN ? (size_t)p.ldb * 4 : 4;        // bytes per unit of k in B
  const int KT = (kend - kbeg + BK - 1) / BK;
  // guarded loads only where the tile is not whole: a partial last k-tile (K % 32 != 0) or the
  // last column tile of a [K][N] operand
  const bool guarded = (kend - kbeg) % BK != 0 || (BKN && n0 + BN > N);

  auto load_set = [&](float4 (&ra)[LA], float4 (&rb)[LB], int kt, auto guard_c) __attribute__((always_inline)) {
    constexpr bool GUARD = decltype(guard_c)::value;
    const int k = kbeg + min(kt, KT - 1) * BK;             // past the end: the last tile again (never stored... or stored where nobody reads)
    const char* Ak = Ab + (size_t)k * 4;
    const char* Bk = Bb + (size_t)k * bstep;
    if (!GUARD) {
#pragma unroll
      for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const float4*>(Ak + aoff[i]);
#pragma unroll
      for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const float4*>(Bk + boff[i]);
    } else {
      const bool ain = k + scol < kend;
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ain) ra[i] = *reinterpret_cast<const float4*>(Ak + aoff[i]);
      }
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in = BKN ? (bcol_ok[i] && k + (tid + i * NT) / BQ < kend) : ain;
        if (in) rb[i] = *reinterpret_cast<const float4*>(Bk + boff[i]);
      }
    }
  };
  auto store_set = [&](const float4 (&ra)[LA], const float4 (&rb)[LB], int buf) __attribute__((always_inline)) {
    float* As = lds + buf * (ASZ + BSZ);
    float* Bs = As + ASZ;
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<float4*>(As + ((tid + i * NT) >> 3) * LD + scol) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      if (BKN) *reinterpret_cast<float4*>(Bs + (tid + i * NT) * 4) = rb[i];   // [k][n], n contiguous
      else *reinterpret_cast<float4*>(Bs + ((tid + i * NT) >> 3) * LD + scol) = rb[i];
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x16 acc2;                                             // second chain of the single-tile wave (odd k-steps)
#pragma unroll
  for (int e = 0; e < 16; ++e) acc2[e] = 0.f;

  int buf = 0;
  // One k-tile: MFMAs on LDS[buf] (tile kt); meanwhile the loads of tile kt + PF go into the
  // register set `ld` and the set `st` (tile kt + 1) goes to LDS[buf ^ 1]; one barrier.
  // ONE wave per SIMD issues in order: eight global loads in a row in front of the MFMAs leave the
  // matrix pipe idle for ~640 cycles per k-tile, eight ds_write_b128 in a row for ~370 (in-kernel
  // stamps: tools/lab/NOTES.md), while one or two memory instructions behind a group of four
  // MFMAs (256 cycles) are free.  So the order is written out -- after MFMA group (s, i, j) come
  // its share of the global loads (slab 0), of the LDS stores (slab 1 with two register sets: their
  // data landed a k-tile ago) and of the next slab's fragment reads -- and a sched_barrier keeps the
  // compiler from regrouping them.  With two register sets the barrier sits INSIDE the last slab,
  // behind its first MFMAs, and the first fragments of the next k-tile are read right behind it:
  // the LDS round trip at the k-tile boundary (~390 cycles per k-tile of 1024 MFMA cycles on the
  // 64x64 tile) runs under the slab's remaining MFMAs.
  float4 a[2][TI], b[2][TJ];                               // fragments: [slab parity][tile]
  constexpr int G = TI * TJ, NL = LA + LB, NF = TI + TJ;
  auto frag = [&](const float* As, const float* Bs, int st, int s, int q) __attribute__((always_inline)) {   // fragment q of slab s: A rows first, then B
    if (q < TI) {
      a[st][q] = *reinterpret_cast<const float4*>(As + q * 32 * LD + s * 8);
    } else if (BKN) {
      const float* w = Bs + s * 8 * BN + (q - TI) * 32;
      b[st][q - TI] = make_float4(w[0], w[BN], w[2 * BN], w[3 * BN]);
    } else {
      b[st][q - TI] = *reinterpret_cast<const float4*>(Bs + (q - TI) * 32 * LD + s * 8);
    }
  };
  auto frag_base = [&](int bf, const float*& As, const float*& Bs) __attribute__((always_inline)) {
    As = lds + bf * (ASZ + BSZ) + (wm * TI * 32 + r) * LD + 4 * h;
    Bs = lds + bf * (ASZ + BSZ) + ASZ + (BKN ? 4 * h * BN + wn * TJ * 32 + r : (wn * TJ * 32 + r) * LD + 4 * h);
  };
  auto ktile = [&](float4 (&lda_)[LA], float4 (&ldb_)[LB], float4 (&sta_)[LA], float4 (&stb_)[LB], int kt,
                   auto guard_c) __attribute__((always_inline)) {
    constexpr bool GUARD = decltype(guard_c)::value;
    constexpr int SST = PF == 2 ? 1 : BK / 8 - 1;          // the slab whose MFMAs cover the LDS stores
    const int kl = kbeg + min(kt + PF, KT - 1) * BK;       // past the end: the last tile again (stored where nobody reads)
    const char* Ak = Ab + (size_t)kl * 4;
    const char* Bk = Bb + (size_t)kl * bstep;
    if (GUARD) load_set(lda_, ldb_, kt + PF, guard_c);     // (guarded loads sit behind branches: nothing to interleave)
    const float *As, *Bs, *An, *Bn;
    frag_base(buf, As, Bs);
    frag_base(buf ^ 1, An, Bn);
    float* Ad = lds + (buf ^ 1) * (ASZ + BSZ);
    float* Bd = Ad + ASZ;
    if (PF == 1) {
#pragma unroll
      for (int q = 0; q < NF; ++q) frag(As, Bs, 0, 0, q);
    }
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      const bool last = s == BK / 8 - 1;
      // four rounds (the k-steps x, y, z, w of the slab), each ONE MFMA per accumulator tile: back
      // to back MFMAs on the same accumulator issue every ~72 cycles, on alternating accumulators
      // every 64 (measured: the 64x64 tile's single chain ran 13 % under the rate)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int i = g / TJ, j = g % TJ;
          const float av = t == 0 ? a[cur][i].x : t == 1 ? a[cur][i].y : t == 2 ? a[cur][i].z : a[cur][i].w;
          const float bv = t == 0 ? b[cur][j].x : t == 1 ? b[cur][j].y : t == 2 ? b[cur][j].z : b[cur][j].w;
          if (G == 1 && (t & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2, 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
        }
        if (PF == 2 && last && t == 0) {
          __syncthreads();
#pragma unroll
          for (int q = 0; q < NF; ++q) frag(An, Bn, nxt, 0, q);
        }
        if (!last) {
#pragma unroll
          for (int q = 0; q < NF; ++q)
            if ((q * 4) / NF == t) frag(As, Bs, nxt, s + 1, q);
        }
        if (!GUARD && s == 0) {
#pragma unroll
          for (int l = 0; l < NL; ++l)
            if ((l * 4) / NL == t) {
              if (l < LA) lda_[l] = *reinterpret_cast<const float4*>(Ak + aoff[l]);
              else ldb_[l - LA] = *reinterpret_cast<const float4*>(Bk + boff[l - LA]);
            }
        }
        if (s == SST) {
#pragma unroll
          for (int l = 0; l < NL; ++l)
            if ((l * 4) / NL == t) {
              if (l < LA) *reinterpret_cast<float4*>(Ad + ((tid + l * NT) >> 3) * LD + scol) = sta_[l];
              else if (BKN) *reinterpret_cast<float4*>(Bd + (tid + (l - LA) * NT) * 4) = stb_[l - LA];
              else *reinterpret_cast<float4*>(Bd + ((tid + (l - LA) * NT) >> 3) * LD + scol) = stb_[l - LA];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (PF == 1) {
      __syncthreads();
    }
    buf ^= 1;
  };

  float4 r0a[LA], r0b[LB], r1a[LA], r1b[LB];
  auto run = [&](auto guard_c) __attribute__((always_inline)) {
    load_set(r0a, r0b, 0, guard_c);
    if (PF == 2) load_set(r1a, r1b, 1, guard_c);
    store_set(r0a, r0b, 0);
    __syncthreads();
    if (PF == 2) {
      const float *As, *Bs;
      frag_base(0, As, Bs);
#pragma unroll
      for (int q = 0; q < NF; ++q) frag(As, Bs, 0, 0, q);
    }
    if (PF == 1) {
      for (int kt = 0; kt < KT; ++kt) ktile(r0a, r0b, r0a, r0b, kt, guard_c);
    } else {
      for (int kt = 0; kt < KT; kt += 2) {
        ktile(r0a, r0b, r1a, r1b, kt, guard_c);
        if (kt + 1 < KT) ktile(r1a, r1b, r0a, r0b, kt + 1, guard_c);
      }
    }
  };
  if (KT > 0) {
    if (guarded) run(std::true_type{});
    else run(std::false_type{});
  }

  if (TI * TJ == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] += acc2[e];
  }
  // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31,
  // row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): a lane holds 16 rows of ONE column.
  float* Cs = p.C + (size_t)blockIdx.z * p.strideC + (size_t)piece * p.slab;
  auto epilogue = [&](auto full_c, auto zero_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr bool ZERO = decltype(zero_c)::value;     // zero-fill of an unused slab (stream mode)
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
        // the sixteen GELU'(z) factors of this lane in ONE batch of loads (C and Z may alias as far
        // as the compiler knows: read one by one, every load would wait behind the previous store)
        float zv[16];
        if ((EPI == EPI_MUL_GELUGRAD || EPI == EPI_MUL_POS) && !ZERO) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int lr = (e & 3) + 8 * (e >> 2);
            zv[e] = (FULL || (colok && rbase + lr < M)) ? p.Z[off + (unsigned)lr * ldc] : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int lr = (e & 3) + 8 * (e >> 2);
          if (!FULL && !(colok && rbase + lr < M)) continue;
          if (ZERO) {
            Cs[off + (unsigned)lr * ldc] = 0.f;
            continue;
          }
          float v = acc[i][j][e] + bv;
          if (EPI == EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
          if (EPI == EPI_BIAS_GELU2) {
            // one exponential for both: GELU(z) = z Phi(z), GELU'(z) = Phi(z) + z phi(z) (common.h gelu_pair_f)
            float ge, gr;
            gelu_pair_f(v, ge, gr);
            p.Z[off + (unsigned)lr * ldc] = gr;
            v = ge;
          }
          if (EPI == EPI_MUL_GELUGRAD) v *= zv[e];
          if (EPI == EPI_MUL_POS) v = zv[e] > 0.f ? v : 0.f;      // ReLU backward: relu'(0) = 0 as ATen's threshold
          Cs[off + (unsigned)lr * ldc] = v;
        }
      }
    }
  };
  const bool full = m0 + BM <= M && n0 + BN <= N;
  if (full) epilogue(std::true_type{}, std::false_type{});
  else epilogue(std::false_type{}, std::false_type{});
  if (p.symmetric && n0 > m0) {
    // the mirror image below the diagonal: C[col][row] = C[row][col] (the same products in the same order as a block
    // of its own would sum).  A lane's registers 4 g .. 4 g + 3 are four consecutive rows of its column: one 16-byte
    // store per group into row `col` of the transpose
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + (wn * TJ + j) * 32 + r;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int rbase = m0 + (wm * TI + i) * 32 + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int row = rbase + 8 * g4;
          float* dst = Cs + (size_t)col * p.ldc + row;
          if (col >= N) continue;
          if (row + 3 < M) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (row + q < M) dst[q] = acc[i][j][4 * g4 + q];
          }
        }
      }
    }
  }
  if (!p.stream_blocks) break;
  if (tile_ends)
    for (int q = piece + 1; q < p.slabs; ++q) {
      Cs = p.C + (size_t)blockIdx.z * p.strideC + (size_t)q * p.slab;
      if (full) epilogue(std::true_type{}, std::true_type{});
      else epilogue(std::false_type{}, std::true_type{});
    }
  if (u >= uend) break;
  __syncthreads();      // every wave is done with the LDS buffers before the next segment stages into them
  }   // segments
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradients.  dW_p[N_p, K_p] = sum over rows m of dY_p[m, n] X_p[m, k] for every
// layer p of a group in ONE launch; tiles are read "down the columns": lane (r, h) of MFMA step t
// takes dY[m = 2t + h][n0 + r] (ds_read_b32, consecutive banks).
//
// Work = (output tile, 32-row chunk of the reduction) units in tile-major order, dealt in EQUAL
// contiguous ranges to a fixed grid that fills the chip once (stream-K along the reduction): a
// block of 108 tiles x 92 chunks (a block's qkv, proj, fc1, fc2 at M = 2944) is 9936 units = 19.4
// per block of a 512-block grid, instead of 108 x splits tiles that never divide 256 CUs (the
// uniform-split version ran 2 or 4 blocks on a CU: 117 us where the MFMAs need 66).  A block writes
// the partial tile of every output tile its range touches into its own slots; wgrad_reduce adds a
// tile's partials in block order: no atomics, bit-identical run to run.

template <int TN, int NT>
__global__ __launch_bounds__(NT) void wgrad_kernel(const WgradArgs g) {
  constexpr int TM = WTM;
  constexpr int WNV = NT / 128;                       // waves along k (two along n)
  constexpr int WK = TN / WNV;                        // columns of a wave: 64 (TN 128) or 96 (TN 384)
  constexpr int TJ = WK / 32;                         // MFMA tiles of a wave along k
  constexpr int WSLOT = wslot(TN);
  constexpr int ROW4 = (TM + TN) / 4;                 // float4 per staged row
  constexpr int SLOTS = (TBK * ROW4 + NT - 1) / NT;   // float4 per thread per slab
  // staging map.  Joint (NT a multiple of ROW4): thread -> four columns of the [dY | X] row, the SAME in every slot, rows
  // srow0 + i RSTEP.  Split (TN = 256: 96 float4 per row do not divide 512 threads): every thread stages one float4 of
  // the dY band (row tid / 32) AND SLOTS - 1 of the X band (rows tid / XR4 + i XSTEP) -- again the same columns per slot.
  constexpr bool SPLIT = NT % ROW4 != 0;
  constexpr int XR4 = TN / 4, XSTEP = NT / XR4;
  static_assert((TBK * ROW4) % NT == 0 && WK % 32 == 0, "slot layout");
  static_assert(!SPLIT || (NT == TBK * (TM / 4) && NT % XR4 == 0 && XSTEP * (SLOTS - 1) == TBK), "split slot layout");
  constexpr int RSTEP = SPLIT ? TBK : NT / ROW4;      // threads that stage the same dY columns
  extern __shared__ float wg_lds[];                   // [2][TBK * (TM + TN)]
  float* const lds0 = wg_lds;
  float* const lds1 = wg_lds + TBK * (TM + TN);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNV, wn = wave % WNV;
  const int r = lane & 31, h = lane >> 5;
  const int srow0 = SPLIT ? tid / (TM / 4) : tid / ROW4;   // a thread stages the SAME four columns in every slot
  const int scol = SPLIT ? (tid % (TM / 4)) * 4 : (tid % ROW4) * 4;
  const bool isb = !SPLIT && scol >= TM;
  const int xrow0 = tid / XR4, xcol = (tid % XR4) * 4;     // (split map) this thread's X rows / columns
  // blocks b, b + 8, ... share an XCD (its L2): give an XCD CONSECUTIVE ranges, i.e. a run of tiles with
  // all their chunks, so the tk re-reads of a dY column band and the tn re-reads of an X band stay in one L2
  // (the profile showed 408 MB fetched per launch for 40-110 MB of operands with ranges dealt round robin)
  const int nb8 = g.blocks >> 3;
  const int sb = g.blocks % 8 == 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  long long u = wg_start(g, sb);
  const long long uend = wg_start(g, sb + 1);
  float* slot = g.partials + (size_t)sb * g.slots * WSLOT;
  for (; u < uend; slot += WSLOT) {
    const WgradProb& P = g.p[wg_prob_of_unit(g, u)];
    const long long rel = u - P.unit0;
    const int lt = P.lt0 + (int)(rel / P.chunks), c0 = (int)(rel % P.chunks);
    const int c1 = (int)min((long long)P.chunks, c0 + (uend - u));
    u += c1 - c0;
    const int bx = lt % P.tk, by = lt / P.tk;
    const int n0 = by * TM, k0 = bx * TN;
    const int N = P.N, K = P.K;
    const int mbeg = c0 * WCH, mend = min(P.M, c1 * WCH);
    const int gcol = isb ? k0 + scol - TM : n0 + scol;
    const bool ok = isb ? gcol < K : gcol < N;
    const float* src = isb ? P.X + gcol : P.dY + gcol;
    const int ld = isb ? K : N;
    const bool xok = k0 + xcol < K;                   // (split map)
    const float* xsrc = P.X + k0 + xcol;
    float4 rg[SLOTS];
    float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);    // column sums of this thread's dY elements
    const bool sum_a = P.db != nullptr && bx == 0 && !isb;
    // listed operands: a slab of TBK = 16 rows lies inside ONE 32-row group (units are 32-row chunks), so the stored
    // row of product row m is m + rowoff with ONE block-uniform offset per slab and operand
    const int* const grp_a = g.a_groups;
    const int* const grp_b = g.b_groups;
    const bool listed = grp_a != nullptr || grp_b != nullptr;
    // BatchNorm + ReLU producer on this thread's X columns (the same in every slot)
    const bool bnrelu = g.scale != nullptr;
    float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bnrelu) {
      const int xc = SPLIT ? k0 + xcol : gcol;
      if ((SPLIT ? xok : (isb && ok))) {
        bsc = *reinterpret_cast<const float4*>(g.scale + xc);
        bsh = *reinterpret_cast<const float4*>(g.shift + xc);
      }
    }
    auto producer = [&](float4 v) {
      v.x = fmaxf(v.x * bsc.x + bsh.x, 0.f), v.y = fmaxf(v.y * bsc.y + bsh.y, 0.f);
      v.z = fmaxf(v.z * bsc.z + bsh.z, 0.f), v.w = fmaxf(v.w * bsc.w + bsh.w, 0.f);
      return v;
    };
    bool rv[SLOTS];                                   // slot holds a row inside the matrix (the producer skips the zero fill)
    auto gload = [&](int mt) {
      int offa = 0, offb = 0;                         // stored row - product row, this slab
      if (listed) {
        const int gq = mt >> 5;
        if (grp_a) offa = (grp_a[gq] - gq) * 32;
        if (grp_b) offb = (grp_b[gq] - gq) * 32;
      }
      if (SPLIT) {
        const int gm = mt + srow0;
        rg[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && gm < mend) rg[0] = *reinterpret_cast<const float4*>(src + (size_t)(gm + offa) * ld);
#pragma unroll
        for (int i = 1; i < SLOTS; ++i) {
          const int xm = mt + xrow0 + (i - 1) * XSTEP;
          rg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          rv[i] = xok && xm < mend;
          if (rv[i]) rg[i] = *reinterpret_cast<const float4*>(xsrc + (size_t)(xm + offb) * K);
        }
        return;
      }
      const int off = isb ? offb : offa;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int gm = mt + srow0 + i * RSTEP;
        rg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        rv[i] = ok && gm < mend;
        if (rv[i]) rg[i] = *reinterpret_cast<const float4*>(src + (size_t)(gm + off) * ld);
      }
    };
    auto lstore = [&](float* buf) {
      if (SPLIT) {
        const float4 v = rg[0];
        if (sum_a) asum.x += v.x, asum.y += v.y, asum.z += v.z, asum.w += v.w;
        *reinterpret_cast<float4*>(&buf[srow0 * (TM + TN) + scol]) = v;
#pragma unroll
        for (int i = 1; i < SLOTS; ++i) {
          if (bnrelu && rv[i]) rg[i] = producer(rg[i]);   // (rows past the end stay zero: relu(shift) is not)
          *reinterpret_cast<float4*>(&buf[(xrow0 + (i - 1) * XSTEP) * (TM + TN) + TM + xcol]) = rg[i];
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        if (bnrelu && isb && rv[i]) rg[i] = producer(rg[i]);
        const float4 v = rg[i];
        if (sum_a) asum.x += v.x, asum.y += v.y, asum.z += v.z, asum.w += v.w;
        *reinterpret_cast<float4*>(&buf[(srow0 + i * RSTEP) * (TM + TN) + scol]) = v;
      }
    };
    f32x16 acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    gload(mbeg);
    lstore(lds0);
    __syncthreads();
    int buf = 0;
    for (int mt = mbeg; mt < mend; mt += TBK) {
      const bool more = mt + TBK < mend;
      if (more) gload(mt + TBK);
      const float* T = (buf ? lds1 : lds0) + h * (TM + TN);
      float fa[2][2][2], fb[2][2][TJ];   // [stage][step within stage][tile]
#define PDAE_WG_FREAD(st, t)                                        \
      {                                                             \
        const float* row = T + 2 * (t) * (TM + TN);                 \
        fa[st][(t) & 1][0] = row[wm * 64 + r];                      \
        fa[st][(t) & 1][1] = row[wm * 64 + 32 + r];                 \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j)              \
          fb[st][(t) & 1][j] = row[TM + wn * WK + j * 32 + r];      \
      }
      PDAE_WG_FREAD(0, 0)
      PDAE_WG_FREAD(0, 1)
#pragma unroll
      for (int tt = 0; tt < TBK / 4; ++tt) {
        const int cur = tt & 1, nxt = cur ^ 1;
        if (tt + 1 < TBK / 4) {
          PDAE_WG_FREAD(nxt, 2 * tt + 2)
          PDAE_WG_FREAD(nxt, 2 * tt + 3)
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][i], fb[cur][q][j], acc[i][j], 0, 0, 0);
        }
        if (tt == 0) __builtin_amdgcn_sched_group_barrier(0x020, SLOTS, 0);
#pragma unroll
        for (int q = 0; q < 4 * TJ; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (tt + 1 < TBK / 4) {
            if (q < 2 * (2 + TJ)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          } else if (q < SLOTS) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        if (tt == TBK / 4 - 2 && more) lstore(buf ? lds0 : lds1);
      }
#undef PDAE_WG_FREAD
      __syncthreads();
      buf ^= 1;
    }
    // ---- the partial tile (whole TM x TN, edges included: the reduction stores what is inside)
    if (P.db != nullptr && bx == 0) {
      // the RSTEP threads that staged the same four columns add up in thread order (fixed)
      float* red = lds0;                                // [RSTEP][TM]; the slab buffers are free now
      if (!isb) *reinterpret_cast<float4*>(red + srow0 * TM + scol) = asum;
      __syncthreads();
      if (tid < TM) {
        float sum = red[tid];
#pragma unroll
        for (int q = 1; q < RSTEP; ++q) sum += red[q * TM + tid];
        slot[TM * TN + tid] = sum;
      }
      __syncthreads();                                  // red is the next segment's slab buffer
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          slot[row * TN + wn * WK + j * 32 + r] = acc[i][j][e];
        }
  }
}

// ---- host side ------------------------------------------------------------------------------
struct Cfg {
  int ti, tj, wm, wn;
};
constexpr int NCFG = 8;
static const Cfg kCfg[NCFG] = {
    {2, 2, 2, 2},   // 0: 128 x 128
    {1, 2, 2, 2},   // 1:  64 x 128
    {2, 1, 2, 2},   // 2: 128 x  64
    {1, 1, 2, 2},   // 3:  64 x  64
    {1, 3, 2, 2},   // 4:  64 x 192
    {3, 1, 1, 4},   // 5:  96 x 128
    {1, 3, 4, 1},   // 6: 128 x  96
    {2, 3, 2, 2},   // 7: 128 x 192
};

static size_t lds_bytes(const Cfg& c, bool bkn) {
  const int bm = 32 * c.ti * c.wm, bn = 32 * c.tj * c.wn;
  return 2 * sizeof(float) * ((size_t)bm * LD + (bkn ? (size_t)BK * bn : (size_t)bn * LD));
}

// Cost of a plan in MFMA issue slots (64 cycles) of the busiest SIMD.  Blocks are dealt to the 256
// CUs round robin and the blocks resident on a CU share its four SIMDs, so the busiest CU decides:
// ceil(blocks / 256) blocks of TI TJ K/2 MFMAs per SIMD each.  Calibrated on tools/lab_rows.py
// sweeps (M = 1664 .. 8192, the eight GEMMs of a block): small tiles win -- four or more blocks
// stay resident per CU and cover each other's load / store issue and epilogues -- unless a larger
// tile saves a whole round; a block alone on its CU pays its memory-instruction issue in full
// (+33 % on the 64x64 tile, +12 % on 128x128, in-kernel stamps); every launch pays ~3 us of ramp
// and drain, every split-K slab a little consumer time.
static double plan_cost(int M, int N, int K, int cfg, int splits, bool bkn) {
  const Cfg& c = kCfg[cfg];
  const int bm = 32 * c.ti * c.wm, bn = 32 * c.tj * c.wn;
  const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const int kc = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  const long long per_cu = (tiles * splits + 255) / 256;
  const int g = c.ti * c.tj;
  const double mfma = (double)g * (kc / 2);                       // per wave = per SIMD and block
  const double big = g == 1 ? 1.0 : g == 2 ? 1.04 : g == 3 ? 1.07 : g == 4 ? 1.15 : 1.25;
  const double alone = g == 1 ? 1.33 : g == 2 ? 1.2 : 1.12;      // nobody to cover the issue bubbles
  double cost = per_cu * mfma * (per_cu == 1 ? alone : big) + 100.;
  if (splits > 1) cost += 8. * splits;                            // the consumer reads `splits` slabs
  (void)bkn;
  return cost;
}

// slabs a stream-K plan needs: the most pieces any tile is cut into when `units` = tiles * kt
// units are dealt to P blocks in ranges [b units / P, (b + 1) units / P)
static int stream_slabs(long long tiles, int kt, int P) {
  const long long units = tiles * kt;
  int most = 1;
  for (long long t = 0; t < tiles; ++t) {
    const long long u0 = t * kt, u1 = u0 + kt - 1;
    const int n = (int)(((u1 + 1) * P - 1) / units - ((u0 + 1) * P - 1) / units) + 1;
    if (n > most) most = n;
  }
  return most;
}

static void plan_rows(int M, int N, int K, bool bkn, bool may_split, int* cfg, int* splits, int* stream_blocks) {
  static const int order[] = {3, 1, 2, 6, 4, 0, 5, 7};           // ties go to the smaller tile
  double best = 1e300;
  *cfg = 3, *splits = 1, *stream_blocks = 0;
  for (int c : order)
    for (int s = 1; s <= (may_split ? 4 : 1); ++s) {
      if (s > 1 && K / s < 128) continue;
      const double t = plan_cost(M, N, K, c, s, bkn);
      if (t < best * 0.995) best = t, *cfg = c, *splits = s;
    }
  if (may_split) {
    // stream-K on the 64x64 tile: every SIMD carries units / P * 16 MFMAs per resident block
    const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
    const int kt = (K + BK - 1) / BK;
    // the first grid whose tiles are cut into at most THREE pieces: a slab is a 5.5 MB write for the producer and a
    // 5.5 MB read for the LayerNorm that adds it (tools/lab/NOTES.md: 4 slabs 12.71, 3 slabs 12.68, 2 slabs
    // 12.71 ms/step; 768 / 512 blocks at most 12.71-12.73)
    constexpr int lab_pmax = 1024, lab_smax = 3;
    constexpr bool lab_768 = false;
    for (int P : {1024, 768, 512, 256}) {
      if (P > lab_pmax || (P == 768 && !lab_768)) continue;
      const long long units = tiles * kt;
      if (units < 2LL * P) continue;
      const int S = stream_slabs(tiles, kt, P);
      if (S > lab_smax) continue;
      const double per_simd = (double)((units + P - 1) / P) * 16. * (P / 256);
      const double t = per_simd * (P == 256 ? 1.33 : 1.0) + 110. + 8. * S;
      if (t < best * 0.97) best = t, *cfg = 3, *splits = S, *stream_blocks = P;
      break;
    }
  }
}

template <int TI, int TJ, int WM, int WN, bool BKN, int EPI>
static void launch_cfg(Args& a, int splits, int stream_blocks, hipStream_t s, int batch = 1) {
  constexpr int BM = 32 * TI * WM, BN = 32 * TJ * WN;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
  if (a.symmetric) a.tiles = a.tiles_n * (a.tiles_n + 1) / 2;   // (square tiles, M == N: the caller checked)
  a.kchunk = ((a.K + splits - 1) / splits + BK - 1) / BK * BK;
  const Cfg c = {TI, TJ, WM, WN};
  const size_t lds = lds_bytes(c, BKN);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rows_gemm_kernel<TI, TJ, WM, WN, BKN, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  const int chunk = (a.tiles + 7) / 8;
  a.stream_blocks = stream_blocks, a.slabs = splits;
  const dim3 grid = stream_blocks ? dim3(stream_blocks, 1, batch) : dim3(8 * chunk, splits, batch);
  hipLaunchKernelGGL((rows_gemm_kernel<TI, TJ, WM, WN, BKN, EPI>), grid, dim3(WM * WN * 64), lds, s, a);
}

template <bool BKN, int EPI>
static void launch_rows(Args& a, int cfg, int splits, int sb, hipStream_t s) {
  switch (cfg) {
    case 0: launch_cfg<2, 2, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 1: launch_cfg<1, 2, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 2: launch_cfg<2, 1, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 3: launch_cfg<1, 1, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 4: launch_cfg<1, 3, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 5: launch_cfg<3, 1, 1, 4, BKN, EPI>(a, splits, sb, s); break;
    case 6: launch_cfg<1, 3, 4, 1, BKN, EPI>(a, splits, sb, s); break;
    default: launch_cfg<2, 3, 2, 2, BKN, EPI>(a, splits, sb, s); break;
  }
}

}  // namespace rows
}  // namespace pdae

using namespace pdae;
using namespace pdae::rows;


// ---- GEMM arithmetic of the row-GEMM family: exact-split bf16 (rows3_gemm.hip) unless PDAE_GEMM=f32mfma or
// pdae_set_gemm_arith(PDAE_GEMM_F32MFMA) asks for the fp32-input MFMA kernels above
// (the setting lives in the calling thread's context: det.hip)
static int gemm_arith() {
  int& a = ctx_gemm_arith();
  if (a < 0) {
    const char* e = getenv("PDAE_GEMM");
    a = (e && (!strcmp(e, "f32mfma") || !strcmp(e, "f32") || !strcmp(e, "fp32"))) ? PDAE_GEMM_F32MFMA : PDAE_GEMM_BF16X3;
  }
  return a;
}
extern "C" int pdae_set_gemm_arith(int arith) {
  if (arith != PDAE_GEMM_F32MFMA && arith != PDAE_GEMM_BF16X3) return bad_arg("set_gemm_arith: PDAE_GEMM_F32MFMA or PDAE_GEMM_BF16X3");
  ctx_gemm_arith() = arith;
  return PDAE_OK;
}
extern "C" int pdae_gemm_arith(void) { return gemm_arith(); }
// (lab) PDAE_GEMM_ONLY=rows | wgrad: the exact-split arithmetic for that half of the family only
static int arith_of(bool wgrad) {
  static const char* only = getenv("PDAE_GEMM_ONLY");
  if (only && gemm_arith() == PDAE_GEMM_BF16X3 && (wgrad ? strcmp(only, "wgrad") : strcmp(only, "rows")) != 0) return PDAE_GEMM_F32MFMA;
  return gemm_arith();
}

namespace pdae {
int gemm_arith_rows() { return arith_of(false); }
}  // namespace pdae

// shapes the exact-split kernels take: the reduction in whole 32-deep tiles (every layer of the models but the K = 3
// / K = 4 ones), 32-bit byte offsets as the fp32 kernels; a [K, N] weight is staged in octets of its rows (16-byte loads)
static bool gemm3_takes(int N, int K, bool bkn) { return K % 32 == 0 && K >= 32 && (!bkn || N % 4 == 0); }

// Plan of the exact-split family (cfg = CFG3_BASE + tile shape).  Cost in microseconds, calibrated on tools/lab/
// rows3_lab.py sweeps (M = 1664 .. 65536, the Transformer blocks' N, K): one block per CU at a time, so a launch is
// ceil(blocks / 256) ROUNDS of [k-tiles x time per 32-deep k-tile + the block's exposed first loads and epilogue],
// + ~4 us of launch ramp and drain, + a little consumer time per extra slab.
static double plan3_cost(int M, int N, int K, int c3, int splits) {
  static const double t_ktile[rows3::NCFG3] = {1.0, 0.66, 1.5, 0.75}, t_fixed[rows3::NCFG3] = {6.0, 4.0, 7.5, 1.5};
  const rows3::Cfg3& c = rows3::kCfg3[c3];
  const int bm = 32 * c.ti * c.wm, bn = 32 * c.tj * c.wn;
  // the grid is 8 ceil(tiles / 8) x splits blocks (whole XCD chunks); a grid within a few blocks of 256 does not land
  // one block per CU (measured: 264 launched / 252 working blocks ran as two rounds), so a round is 248 blocks
  const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const long long blocks = (tiles + 7) / 8 * 8 * splits;
  const int kc = ((K + splits - 1) / splits + 31) / 32 * 32;
  const long long rounds = (blocks + 247) / 248;
  double cost = 4.0 + rounds * (kc / 32 * t_ktile[c3] + t_fixed[c3]);
  if (splits > 1) cost += 0.7 * (splits - 1);
  return cost;
}
// may_split: 0 = one slab; 1 = up to 4 slabs (the LayerNorm kernels add them on their way); 2..8 = up to that many (a
// caller with its own slab consumer: a handful of rows against a long reduction, nn_ops.mlp_chain + pdae_slab_sum_epi)
static void plan_rows3(int M, int N, int K, int may_split, int* cfg, int* splits, int* stream_blocks) {
  double best = 1e300;
  *cfg = rows3::CFG3_BASE, *splits = 1, *stream_blocks = 0;
  static const char* force = getenv("PDAE_ROWS3_FORCE");       // lab: "cfg,splits" overrides every plan
  int fc = -1, fs = -1;
  if (force) sscanf(force, "%d,%d", &fc, &fs);
  const int smax = may_split <= 0 ? 1 : (may_split == 1 ? 4 : (may_split > 8 ? 8 : may_split));
  for (int c = 0; c < rows3::NCFG3; ++c)
    for (int s = 1; s <= smax; ++s) {
      if (s > 1 && K / s < 128) continue;
      const double t = plan3_cost(M, N, K, c, s);
      if (t < best * 0.995) best = t, *cfg = rows3::CFG3_BASE + c, *splits = s;
    }
  if (fc >= 0) *cfg = rows3::CFG3_BASE + fc;
  if (fs >= 1 && may_split) {                                  // (a forced split count still leaves every slab a tile)
    while (fs > 1 && (long long)(fs - 1) * (((K + fs - 1) / fs + 31) / 32 * 32) >= K) --fs;
    *splits = fs;
  }
}

extern "C" int pdae_rows_gemm_plan(int M, int N, int K, int w_kn, int may_split, int* cfg, int* splits,
                                   int* stream_blocks) {
  if (M < 0 || N <= 0 || K <= 0 || !cfg || !splits || !stream_blocks) return bad_arg("rows_gemm_plan: bad argument");
  if (arith_of(false) == PDAE_GEMM_BF16X3 && gemm3_takes(N, K, w_kn != 0))
    plan_rows3(M > 0 ? M : 1, N, K, may_split, cfg, splits, stream_blocks);
  else plan_rows(M > 0 ? M : 1, N, K, w_kn != 0, may_split != 0, cfg, splits, stream_blocks);
  return PDAE_OK;
}

extern "C" int pdae_rows_gemm(int M, int N, int K, const float* X, const float* W, int w_kn,
                              const float* bias, int epi, float* Z, float* Y, int cfg, int splits,
                              int stream_blocks, pdae_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return bad_arg("rows_gemm: bad size");
  if (K % 4 != 0 || (w_kn && N % 4 != 0)) return unsupported("rows_gemm: K (and N for a [K,N] weight) must be multiples of 4");
  if (epi < 0 || epi > 4) return bad_arg("rows_gemm: epi must be 0..4");
  if ((long long)M * (K > N ? K : N) >= (1LL << 30) || (long long)N * K >= (1LL << 30))
    return unsupported("rows_gemm: operands of 4 GB or more (32-bit byte offsets)");
  const bool fam3 = cfg >= rows3::CFG3_BASE;
  if ((fam3 ? cfg - rows3::CFG3_BASE >= rows3::NCFG3 : cfg >= NCFG) || splits > 8 || splits == 0) return bad_arg("rows_gemm: bad plan");
  if (fam3 && !gemm3_takes(N, K, w_kn != 0)) return unsupported("rows_gemm: the exact-split bf16 tile shapes need K % 32 == 0");
  if (fam3 && splits > 1) {
    // every slab of the reduction needs at least one 32-deep tile: a block whose slab starts at or past K would index
    // its loads from a negative tile count (the kernel clamps positions to KT - 1)
    const int kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    if ((long long)(splits - 1) * kchunk >= K) return bad_arg("rows_gemm: more split-K slabs than 32-deep tiles of the reduction");
  }
  if (cfg < 0 || splits < 0) {
    int c, s, b;
    if (arith_of(false) == PDAE_GEMM_BF16X3 && gemm3_takes(N, K, w_kn != 0)) plan_rows3(M > 0 ? M : 1, N, K, 0, &c, &s, &b);
    else plan_rows(M > 0 ? M : 1, N, K, w_kn != 0, false, &c, &s, &b);
    if (cfg < 0) cfg = c;
    if (splits < 0) splits = 1, stream_blocks = 0;
  }
  if (stream_blocks < 0 || stream_blocks % 8 != 0) return bad_arg("rows_gemm: stream_blocks must be a multiple of 8");
  if (stream_blocks && cfg >= rows3::CFG3_BASE) return unsupported("rows_gemm: the exact-split tile shapes have no stream-K form");
  if (stream_blocks) {
    const bool f3 = cfg >= rows3::CFG3_BASE;
    const int bm = f3 ? 32 * rows3::kCfg3[cfg - rows3::CFG3_BASE].ti * rows3::kCfg3[cfg - rows3::CFG3_BASE].wm : 32 * kCfg[cfg].ti * kCfg[cfg].wm;
    const int bn = f3 ? 32 * rows3::kCfg3[cfg - rows3::CFG3_BASE].tj * rows3::kCfg3[cfg - rows3::CFG3_BASE].wn : 32 * kCfg[cfg].tj * kCfg[cfg].wn;
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (M > 0 && (tiles * ((K + BK - 1) / BK) < stream_blocks ||
                  stream_slabs(tiles, (K + BK - 1) / BK, stream_blocks) > splits))
      return bad_arg("rows_gemm: too few slabs (or units) for this stream-K grid");
    if (epi != EPI_STORE || bias) return bad_arg("rows_gemm: stream-K slabs take the plain store epilogue without bias");
  }
  if (splits > 1 && (epi != EPI_STORE || bias)) return bad_arg("rows_gemm: split-K slabs take the plain store epilogue without bias");
  if (M == 0) return PDAE_OK;
  if (!X || !W || !Y) return bad_arg("rows_gemm: null pointer");
  if ((epi == EPI_BIAS_GELU2 || epi == EPI_MUL_GELUGRAD || epi == EPI_MUL_POS) && !Z)
    return bad_arg("rows_gemm: this epilogue needs Z");
  if (epi == EPI_BIAS_GELU2 && w_kn) return unsupported("rows_gemm: bias+GELU epilogue on a [K,N] weight");
  if (epi == EPI_BIAS_RELU && w_kn) return unsupported("rows_gemm: bias+ReLU epilogue on a [K,N] weight");
  if (epi == EPI_MUL_GELUGRAD && !w_kn) return unsupported("rows_gemm: GELU' epilogue on an [N,K] weight");
  Args a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = w_kn ? N : K, a.C = Y, a.ldc = N;
  a.Z = Z, a.bias = bias, a.slab = (long long)M * N;
  hipStream_t s = as_stream(stream);
  if (cfg >= rows3::CFG3_BASE) {
    rows3::launch_gemm3(a, cfg - rows3::CFG3_BASE, w_kn != 0, epi, splits, stream_blocks, s);
    return check_launch("rows_gemm");
  }
  if (!w_kn) {
    if (epi == EPI_STORE) launch_rows<false, EPI_STORE>(a, cfg, splits, stream_blocks, s);
    else if (epi == EPI_BIAS_RELU) launch_rows<false, EPI_BIAS_RELU>(a, cfg, splits, 0, s);
    else if (epi == EPI_MUL_POS) launch_rows<false, EPI_MUL_POS>(a, cfg, splits, 0, s);
    else launch_rows<false, EPI_BIAS_GELU2>(a, cfg, splits, 0, s);
  } else {
    if (epi == EPI_STORE) launch_rows<true, EPI_STORE>(a, cfg, splits, stream_blocks, s);
    else if (epi == EPI_MUL_POS) launch_rows<true, EPI_MUL_POS>(a, cfg, splits, 0, s);
    else launch_rows<true, EPI_MUL_GELUGRAD>(a, cfg, splits, 0, s);
  }
  return check_launch("rows_gemm");
}

// ---- data gradient into a BatchNorm + ReLU, with the sums of BatchNorm's backward out of the SAME launch (round 6) ----
// T[M,N] = relu'(bn(X)) ? (dY[M,K] . W[K,N]) : 0,   S[0][n] = sum_m T[m][n],   S[1][n] = sum_m T[m][n] xhat[m][n]
// The sweep that used to compute S re-read T and X (bnrelu_backward_reduce: 2 x 89 us of the cfg3 step, half of cfg2's 2.7
// ms of BatchNorm backward): here the product tile is still in the accumulators when it is masked and summed
// (gemm3_kernel, EPI_BNRELU_STATS: one partial row per 128-row band, no atomics), and a small pass adds the bands in order
// in fp64.  On the fp32-input arithmetic (or a shape the exact-split family does not take) the entry runs the plain data
// gradient and the old sweep: same results up to summation order.
// S[c] = sum over the blocks that had tiles (grid of gx blocks, XCD-chunked tile order: block b owns tiles iff
// (b & 7) * chunk + (b >> 3) < tiles and (b >> 3) < chunk) of their partial rows, in block order, fp64
__global__ __launch_bounds__(1024) void bn_stats_finish_kernel(int gx, int tiles, int W2, const float* __restrict__ part,
                                                                float* __restrict__ S) {
  __shared__ double red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int chunk = (tiles + 7) >> 3;
  double t = 0.0;
  if (c < W2)
    for (int b = ph; b < gx; b += 16) {
      const int sl = b >> 3, tl = (b & 7) * chunk + sl;
      if (sl < chunk && tl < tiles) t += (double)part[(size_t)b * W2 + c];
    }
  red[ph][threadIdx.x & 63] = t;
  __syncthreads();
  if (ph == 0 && c < W2) {
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][threadIdx.x & 63];
    S[c] = (float)t;
  }
}

// one partial row per block of the launch: at most one block per tile, at most one residency of the chip (+ the XCD padding)
// first level of the finishing pass for launches of thousands of one-tile blocks (a reduction too short for persistent
// blocks: K = 64): block (x, y) adds rows [256 y, 256 y + 256) of the blocks that had tiles, in order, fp64 -> mid[y][c]
__global__ __launch_bounds__(1024) void bn_stats_finish1_kernel(int gx, int tiles, int W2, const float* __restrict__ part,
                                                                 float* __restrict__ mid) {
  __shared__ double red[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int chunk = (tiles + 7) >> 3, b0 = blockIdx.y * 256, b1 = min(gx, b0 + 256);
  double t = 0.0;
  if (c < W2)
    for (int b = b0 + ph; b < b1; b += 16) {
      const int sl = b >> 3, tl = (b & 7) * chunk + sl;
      if (sl < chunk && tl < tiles) t += (double)part[(size_t)b * W2 + c];
    }
  red[ph][threadIdx.x & 63] = t;
  __syncthreads();
  if (ph == 0 && c < W2) {
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][threadIdx.x & 63];
    mid[(size_t)blockIdx.y * W2 + c] = (float)t;
  }
}

// one partial row per block of the launch (at most one block per tile) + the first-level sums of the two-level finish
extern "C" long long pdae_rows_gemm_bnrelu_stats_workspace(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const long long rows = (long long)((M + 127) / 128) * ((N + 63) / 64) + 8;
  return (rows + (rows + 255) / 256) * 2 * N;
}

extern "C" int pdae_rows_gemm_bnrelu_stats(int M, int N, int K, const float* dY, const float* W, const float* X,
                                           const int32_t* groups, const float* scale, const float* shift, const float* mean,
                                           const float* invstd, float* T, float* S, float* workspace, pdae_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0 || N % 4 != 0 || K % 4 != 0) return bad_arg("rows_gemm_bnrelu_stats: N, K positive multiples of 4");
  if (!S) return bad_arg("rows_gemm_bnrelu_stats: null pointer");
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    (void)hipMemsetAsync(S, 0, sizeof(float) * 2 * (size_t)N, s);
    return check_launch("rows_gemm_bnrelu_stats");
  }
  if (!dY || !W || !X || !scale || !shift || !mean || !invstd || !T) return bad_arg("rows_gemm_bnrelu_stats: null pointer");
  if (groups && M % 32 != 0) return bad_arg("rows_gemm_bnrelu_stats: a group list needs M % 32 == 0");
  if ((long long)M * (K > N ? K : N) >= (1LL << 30) || (long long)N * K >= (1LL << 30))
    return unsupported("rows_gemm_bnrelu_stats: 32-bit byte offsets");
  // the grid launch_cfg3 will choose: persistent (8 x 32 blocks) when the stream of LDS tiles keeps its parity, else one
  // block per tile; 128 x 64 tiles for the narrow layers, 128 x 128 else (the two shapes that carry this epilogue)
  const int c3 = N <= 64 ? 3 : 0, bn = N <= 64 ? 64 : 128;
  const int tiles = ((M + 127) / 128) * ((N + bn - 1) / bn), chunk = (tiles + 7) / 8, kt = K / 32;
  const bool pers = kt % 2 == 0 && kt >= 4 && chunk > 32;
  // (one partial row per BLOCK; a one-tile-per-block launch of thousands of tiles -- a reduction too short for the
  // persistent form -- hands the finishing pass as many rows: it then runs in two levels)
  if (arith_of(false) == PDAE_GEMM_BF16X3 && gemm3_takes(N, K, true) && workspace && N <= 2048) {
    Args a = {};
    a.M = M, a.N = N, a.K = K, a.A = dY, a.lda = K, a.B = W, a.ldb = N, a.C = T, a.ldc = N;
    a.Z = const_cast<float*>(X), a.slab = (long long)M * N;
    a.bn_scale = scale, a.bn_shift = shift, a.bn_mean = mean, a.bn_invstd = invstd, a.z_groups = groups, a.stats_part = workspace;
    rows3::launch_gemm3(a, c3, true, EPI_BNRELU_STATS, 1, 0, s);
    const int gx = 8 * (pers ? 32 : chunk), W2 = 2 * N;
    if (gx > 1024) {
      const int P = (gx + 255) / 256;
      float* mid = workspace + (size_t)gx * W2;
      hipLaunchKernelGGL(bn_stats_finish1_kernel, dim3((W2 + 63) / 64, P), dim3(1024), 0, s, gx, tiles, W2, workspace, mid);
      // (every row of `mid` counts: tiles = 8 P makes the second level's tile test pass for rows 0 .. P - 1)
      hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((W2 + 63) / 64), dim3(1024), 0, s, P, 8 * P, W2, mid, S);
    } else {
      hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((W2 + 63) / 64), dim3(1024), 0, s, gx, tiles, W2, workspace, S);
    }
    return check_launch("rows_gemm_bnrelu_stats");
  }
  // the two-launch form: plain data gradient, then the sums' own sweep (fp64 chains; deterministic mode: ordered partials)
  int rc = pdae_rows_gemm(M, N, K, dY, W, 1, nullptr, EPI_STORE, nullptr, T, -1, 1, 0, stream);
  if (rc) return rc;
  return bnrelu_backward_sums(s, M, N, T, X, scale, shift, mean, invstd, S, groups);
}

// the widest tile every layer's K is a multiple of: 384 (the Transformer blocks: 384, 1536), 256 (FoldingNet, the
// PointNet++ levels: 256, 512, 1024), else 128
static int wgrad_tile_width(int nprob, const int* Ks) {
  if (arith_of(true) == PDAE_GEMM_BF16X3) return 128;         // exact-split kernel: 128 x 128 tiles (two accumulator sets)
  static const char* force = getenv("PDAE_WGRAD_TN");       // A/B switch (tools/lab/ab.sh): caps the tile width
  const int cap = force ? atoi(force) : 384;
  bool w384 = cap >= 384, w256 = cap >= 256;
  for (int q = 0; q < nprob; ++q) {
    if (Ks[q] % 384 != 0) w384 = false;
    if (Ks[q] % 256 != 0) w256 = false;
  }
  return w384 ? 384 : (w256 ? 256 : 128);
}

// Work layout of a grouped launch.  first / count: the window of the group's tiles (in the order of the problems, tiles
// of a problem in (by, bx) order) this launch covers (count < 0: all); whole: one block per tile instead of equal unit
// ranges over one residency of the chip (every problem must then have the same row count).
static int wgrad_layout(int nprob, const int* Ms, const int* Ns, const int* Ks, WgradArgs* g, int WTN, int first = 0,
                        int count = -1, bool whole = false) {
  int tiles = 0, seen = 0;
  long long units = 0;
  for (int q = 0; q < nprob; ++q) {
    if (Ns[q] <= 0 || Ks[q] <= 0 || Ns[q] % 4 != 0 || Ks[q] % 4 != 0)
      return unsupported("rows_wgrad: N, K positive multiples of 4");
    if (Ms[q] <= 0) return bad_arg("rows_wgrad: every layer of a group needs rows");
    WgradProb& P = g->p[q];
    P.N = Ns[q], P.K = Ks[q], P.M = Ms[q];
    P.tk = (Ks[q] + WTN - 1) / WTN;
    P.tile0 = tiles;
    P.chunks = (Ms[q] + WCH - 1) / WCH;
    P.unit0 = units;
    const int all = ((Ns[q] + WTM - 1) / WTM) * P.tk;
    int lo = first - seen, hi = count < 0 ? all : first + count - seen;
    lo = lo < 0 ? 0 : (lo > all ? all : lo);
    hi = hi < lo ? lo : (hi > all ? all : hi);
    P.lt0 = lo;
    seen += all;
    tiles += hi - lo;
    units += (long long)(hi - lo) * P.chunks;
  }
  g->nprob = nprob, g->tiles = tiles, g->units = units;
  if (whole) {                                    // a block per tile: ranges b units / blocks are whole tiles (equal chunks)
    g->blocks = tiles, g->slots = 1;
    return PDAE_OK;
  }
  // one residency of the chip: 256 CUs x 2 blocks of 4 waves (128-wide tiles), x 1 block of 8 waves (384-wide)
  int wg_blocks = (WTN >= 256 || arith_of(true) == PDAE_GEMM_BF16X3) ? 256 : 512;   // (the exact-split kernel: 8 waves, 122 KB of LDS)
  // (lab) PDAE_WGRAD_BPT=k: k blocks per output tile instead of one residency of the chip
  static const int bpt = getenv("PDAE_WGRAD_BPT") ? atoi(getenv("PDAE_WGRAD_BPT")) : 0;
  if (bpt > 0) wg_blocks = (bpt * tiles + 7) / 8 * 8;
  g->blocks = (int)(units < wg_blocks ? units : wg_blocks);
  if (g->blocks == 0) {
    g->slots = 1;
    return PDAE_OK;
  }
  // slots per block: the most tiles one block's unit range touches (ranges are [b units / B, (b + 1) units / B))
  auto tile_of = [&](long long u) {
    int pi = 0;
    for (int q = 1; q < nprob; ++q)
      if (u >= g->p[q].unit0) pi = q;
    return g->p[pi].tile0 + (int)((u - g->p[pi].unit0) / g->p[pi].chunks);
  };
  int slots = 1;
  for (int b = 0; b < g->blocks; ++b) {
    const long long u0 = (long long)b * units / g->blocks, u1 = (long long)(b + 1) * units / g->blocks;
    if (u1 > u0) {
      const int n = tile_of(u1 - 1) - tile_of(u0) + 1;
      if (n > slots) slots = n;
    }
  }
  g->slots = slots;
  return PDAE_OK;
}

// A stack's grouped launch as TWO launches when its tiles fill the chip several times over (exact-split kernel, every
// layer with the same rows): the first `head` tiles -- whole rounds of one tile per block, in tile order -- and the rest
// as equal unit ranges.  With a block per tile the 32 blocks an XCD runs at a time work on CONSECUTIVE tiles of one
// layer, which share their dY band (same by) or their X band (same bx) and stream them in step: the bands come out of
// that XCD's L2 instead of crossing the fabric once per tile (contiguous ranges put the concurrent blocks ~5 tiles
// apart: 3.7 GB per encoder-stack launch against 1.06 GB of operands, profiles/pmc_r04.json).
// The layers of a group in launch order: those with the group's most frequent row count first (stable), the others
// behind them -- the leading run is what wgrad_head_tiles can give whole rounds to (the decoder stack's last block runs
// on the returned rows only and arrives FIRST in the backward's order).  Outputs are per layer: order changes nothing else.
static void wgrad_order(int nprob, const int* Ms, int* perm) {
  int mode = Ms[0], best = 0;
  for (int q = 0; q < nprob; ++q) {
    int c = 0;
    for (int r = 0; r < nprob; ++r) c += Ms[r] == Ms[q];
    if (c > best) best = c, mode = Ms[q];
  }
  int n = 0;
  for (int q = 0; q < nprob; ++q)
    if (Ms[q] == mode) perm[n++] = q;
  for (int q = 0; q < nprob; ++q)
    if (Ms[q] != mode) perm[n++] = q;
}

static int wgrad_head_tiles(int nprob, const int* Ms, const int* Ns, const int* Ks, int WTN) {
  static const char* off = getenv("PDAE_WGRAD_SPLIT");        // lab: 0 switches the two-launch schedule off
  if ((off && atoi(off) == 0) || arith_of(true) != PDAE_GEMM_BF16X3) return 0;
  long long lead = 0, tiles = 0;
  bool run = true;
  for (int q = 0; q < nprob; ++q) {
    if (Ns[q] <= 0 || Ks[q] <= 0) return 0;
    const long long t = (long long)((Ns[q] + WTM - 1) / WTM) * ((Ks[q] + WTN - 1) / WTN);
    run = run && Ms[q] == Ms[0];
    if (run) lead += t;                                        // the leading layers that share their row count
    tiles += t;
  }
  const int rounds = (int)(lead / 256);
  return rounds >= 1 && tiles >= 384 ? rounds * 256 : 0;
}

namespace pdae {
// (the weight-gradient reductions are no longer parked: a step's grouped launches are few and large)
int rows_wgrad_flush(hipStream_t) { return PDAE_OK; }
}  // namespace pdae

extern "C" int pdae_rows_wgrad_multi_workspace(int nprob, const int* Ms, const int* Ns, const int* Ks, long long* floats) {
  if (nprob <= 0 || nprob > WG_MAX || !Ms || !Ns || !Ks || !floats) return bad_arg("rows_wgrad_multi_workspace: bad argument");
  static thread_local WgradArgs g;         // (3 KB: off the stack)
  const int tn = wgrad_tile_width(nprob, Ks);
  int perm[WG_MAX], Mp[WG_MAX], Np[WG_MAX], Kp[WG_MAX];
  wgrad_order(nprob, Ms, perm);
  for (int q = 0; q < nprob; ++q) Mp[q] = Ms[perm[q]], Np[q] = Ns[perm[q]], Kp[q] = Ks[perm[q]];
  Ms = Mp, Ns = Np, Ks = Kp;
  const int head = wgrad_head_tiles(nprob, Ms, Ns, Ks, tn);
  int rc = wgrad_layout(nprob, Ms, Ns, Ks, &g, tn, head);
  if (rc) return rc;
  *floats = (long long)g.blocks * g.slots * wslot(tn);        // (the head tiles are stored by their blocks: no partials)
  return PDAE_OK;
}

template <int TN>
static void wgrad_launch(const WgradArgs& g, int pl, hipStream_t s) {
  constexpr int NT = TN >= 256 ? 512 : 256;
  constexpr int PARTS1 = WTM * TN / 4 / (256 * wru(1));   // reduction blocks per tile with one partial lane
  constexpr int PARTSL = WTM * TN / 4 / 256;               // ... per lane-block of the lane-parallel forms
  const size_t lds = sizeof(float) * 2 * TBK * (WTM + TN);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<TN, NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<TN, NT>), dim3(g.blocks), dim3(NT), lds, s, g);
  if (pl == 1) hipLaunchKernelGGL((wgrad_reduce_kernel<1, TN>), dim3(g.tiles * PARTS1), dim3(256), 0, s, g);
  else if (pl == 4) hipLaunchKernelGGL((wgrad_reduce_kernel<4, TN>), dim3(g.tiles * PARTSL * 4), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<8, TN>), dim3(g.tiles * PARTSL * 8), dim3(256), 0, s, g);
}

extern "C" int pdae_rows_wgrad_multi(int nprob, const int* Ms, const float* const* dY, const float* const* X,
                                     float* const* dW, float* const* db, const int* Ns, const int* Ks,
                                     float* workspace, pdae_stream_t stream) {
  if (nprob <= 0 || nprob > WG_MAX || !Ms || !dY || !X || !dW || !Ns || !Ks) return bad_arg("rows_wgrad_multi: bad argument");
  WgradArgs g = {};
  const int tn = wgrad_tile_width(nprob, Ks);
  for (int q = 0; q < nprob; ++q)
    if (!dW[q] || !dY[q] || !X[q]) return bad_arg("rows_wgrad_multi: null pointer");
  if (!workspace) return bad_arg("rows_wgrad_multi: null workspace");
  int perm[WG_MAX], Mp[WG_MAX], Np[WG_MAX], Kp[WG_MAX];
  wgrad_order(nprob, Ms, perm);
  for (int q = 0; q < nprob; ++q) Mp[q] = Ms[perm[q]], Np[q] = Ns[perm[q]], Kp[q] = Ks[perm[q]];
  Ms = Mp, Ns = Np, Ks = Kp;
  const int head = wgrad_head_tiles(nprob, Ms, Ns, Ks, tn);
  hipStream_t s = as_stream(stream);
  auto bind = [&](WgradArgs& a) {
    for (int q = 0; q < nprob; ++q)
      a.p[q].dY = dY[perm[q]], a.p[q].X = X[perm[q]], a.p[q].dW = dW[perm[q]], a.p[q].db = db ? db[perm[q]] : nullptr;
  };
  bool one_chunk = arith_of(true) == PDAE_GEMM_BF16X3 && head == 0;
  for (int q = 0; q < nprob; ++q) one_chunk = one_chunk && Ms[q] <= WCH;
  if (one_chunk) {
    // every layer on at most 32 rows (the coarse heads' Linear layers on a batch of 32 clouds): a tile IS one unit, so a
    // block per tile stores it itself -- no partials, no reduction launch
    WgradArgs a = {};
    int rc = wgrad_layout(nprob, Ms, Ns, Ks, &a, tn, 0, -1, true);
    if (rc) return rc;
    if (a.tiles == 0) return check_launch("rows_wgrad_multi");
    bind(a);
    a.partials = workspace, a.direct = 1;
    rows3::launch_wgrad3(a, tn, 1, s);
    return check_launch("rows_wgrad_multi");
  }
  if (head > 0) {                          // whole rounds of one tile per block first (wgrad_head_tiles)
    WgradArgs a = {};
    int rc = wgrad_layout(nprob, Ms, Ns, Ks, &a, tn, 0, head, true);
    if (rc) return rc;
    bind(a);
    a.partials = workspace, a.direct = 1;  // (a block per tile: it stores the tile and the bias sums itself, no reduction pass)
    rows3::launch_wgrad3(a, tn, 1, s);
  }
  int rc = wgrad_layout(nprob, Ms, Ns, Ks, &g, tn, head);
  if (rc) return rc;
  if (g.tiles == 0) return check_launch("rows_wgrad_multi");
  bind(g);
  g.partials = workspace;
  // partial lanes of the reduction by the most partials a tile can have: the longest reduction spans the most
  // blocks, the block ranges are units / blocks long
  int max_chunks = g.p[0].chunks;
  for (int q = 1; q < nprob; ++q) max_chunks = g.p[q].chunks > max_chunks ? g.p[q].chunks : max_chunks;
  const long long most = (max_chunks * (long long)g.blocks + g.units - 1) / g.units + 1;
  const int pl = most <= 16 ? 1 : (most <= 64 ? 4 : 8);
  if (arith_of(true) == PDAE_GEMM_BF16X3) rows3::launch_wgrad3(g, tn, pl, s);
  else if (tn == 384) wgrad_launch<384>(g, pl, s);
  else if (tn == 256) wgrad_launch<256>(g, pl, s);
  else wgrad_launch<128>(g, pl, s);
  return check_launch("rows_wgrad_multi");
}

// One weight gradient with the embedder's operand forms (point_dae_amd/patch_embed.py): whole 32-row groups gathered on
// either operand and / or BatchNorm + ReLU recomputed on X while it is staged.  Same kernel, same ordered reduction (no
// atomics, no memset) as the Transformer blocks' grouped launches; workspace = pdae_rows_wgrad_workspace(M, 1, &N, &K).
extern "C" int pdae_rows_wgrad_listed(int M, int N, int K, const float* dY, const int32_t* a_groups, const float* X,
                                      const int32_t* b_groups, const float* scale, const float* shift, float* dW,
                                      float* db, float* workspace, pdae_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !dY || !X || !dW || !workspace) return bad_arg("rows_wgrad_listed: bad argument");
  if ((a_groups || b_groups) && M % 32 != 0) return bad_arg("rows_wgrad_listed: listed operands need M % 32 == 0 (whole groups)");
  if ((scale == nullptr) != (shift == nullptr)) return bad_arg("rows_wgrad_listed: scale and shift come together");
  WgradArgs g = {};
  const int tn = wgrad_tile_width(1, &K);
  int rc = wgrad_layout(1, &M, &N, &K, &g, tn);
  if (rc) return rc;
  g.p[0].dY = dY, g.p[0].X = X, g.p[0].dW = dW, g.p[0].db = db;
  g.a_groups = a_groups, g.b_groups = b_groups, g.scale = scale, g.shift = shift;
  g.partials = workspace;
  const long long most = (g.p[0].chunks * (long long)g.blocks + g.units - 1) / g.units + 1;
  const int pl = most <= 16 ? 1 : (most <= 64 ? 4 : 8);
  hipStream_t s = as_stream(stream);
  if (arith_of(true) == PDAE_GEMM_BF16X3) rows3::launch_wgrad3(g, tn, pl, s);
  else if (tn == 384) wgrad_launch<384>(g, pl, s);
  else if (tn == 256) wgrad_launch<256>(g, pl, s);
  else wgrad_launch<128>(g, pl, s);
  return check_launch("rows_wgrad_listed");
}

extern "C" int pdae_rows_wgrad_workspace(int M, int nprob, const int* Ns, const int* Ks, long long* floats) {
  if (M < 0 || nprob <= 0 || nprob > WG_MAX || !Ns || !Ks || !floats)
    return bad_arg("rows_wgrad_workspace: bad argument");
  int Ms[WG_MAX];
  for (int q = 0; q < nprob; ++q) Ms[q] = M > 0 ? M : 1;
  return pdae_rows_wgrad_multi_workspace(nprob, Ms, Ns, Ks, floats);
}

extern "C" int pdae_rows_wgrad(int M, int nprob, const float* const* dY, const float* const* X,
                               float* const* dW, float* const* db, const int* Ns, const int* Ks,
                               float* workspace, pdae_stream_t stream) {
  if (M < 0 || nprob <= 0 || nprob > WG_MAX || !dY || !X || !dW || !Ns || !Ks)
    return bad_arg("rows_wgrad: bad argument");
  if (M == 0) {
    hipStream_t s = as_stream(stream);
    for (int q = 0; q < nprob; ++q) {
      if (!dW[q]) return bad_arg("rows_wgrad: null pointer");
      (void)hipMemsetAsync(dW[q], 0, sizeof(float) * (size_t)Ns[q] * Ks[q], s);
      if (db && db[q]) (void)hipMemsetAsync(db[q], 0, sizeof(float) * (size_t)Ns[q], s);
    }
    return check_launch("rows_wgrad");
  }
  int Ms[WG_MAX];
  for (int q = 0; q < nprob; ++q) Ms[q] = M;
  return pdae_rows_wgrad_multi(nprob, Ms, dY, X, dW, db, Ns, Ks, workspace, stream);
}

// Batched product Y_b[M,N] = X_b[M,K] . W_b[N,K]^T for b < batch (blockIdx.z), element strides between the
// problems; W_b = X_b gives the Gram matrices of the DGCNN feature-space kNN (models/dgcnn_util.py:7-12).
extern "C" int pdae_rows_gemm_batched(int batch, int M, int N, int K, const float* X, long long strideX,
                                      const float* W, long long strideW, float* Y, long long strideY,
                                      pdae_stream_t stream) {
  if (batch < 0 || M < 0 || N <= 0 || K <= 0) return bad_arg("rows_gemm_batched: bad size");
  if (K % 4 != 0) return unsupported("rows_gemm_batched: K must be a multiple of 4");
  if (batch > 65535) return unsupported("rows_gemm_batched: more than 65535 problems");
  if ((long long)M * (K > N ? K : N) >= (1LL << 30) || (long long)N * K >= (1LL << 30))
    return unsupported("rows_gemm_batched: per-problem operands of 4 GB or more");
  if (batch == 0 || M == 0) return PDAE_OK;
  if (!X || !W || !Y) return bad_arg("rows_gemm_batched: null pointer");
  int cfg, splits, sb;
  plan_rows(M * (batch < 16 ? batch : 16), N, K, false, false, &cfg, &splits, &sb);   // the grid is batch x tiles
  Args a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = N;
  a.slab = (long long)M * N, a.strideA = strideX, a.strideB = strideW, a.strideC = strideY;
  hipStream_t s = as_stream(stream);
  static const bool sym_off = getenv("PDAE_GRAM_SYMMETRIC") && atoi(getenv("PDAE_GRAM_SYMMETRIC")) == 0;   // lab: A/B switch
  if (X == W && strideX == strideW && M == N && N % 4 == 0 && strideY % 4 == 0 && !sym_off &&
      (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
    // a Gram matrix X X^T: the 64 x 64 tiles on and above the diagonal, each stored a second time transposed (half the
    // products; G[j][i] is bit for bit the G[i][j] a tile of its own would compute)
    a.symmetric = 1;
    launch_cfg<1, 1, 2, 2, false, EPI_STORE>(a, 1, 0, s, batch);
    return check_launch("rows_gemm_batched");
  }
  switch (cfg) {
    case 0: launch_cfg<2, 2, 2, 2, false, EPI_STORE>(a, 1, 0, s, batch); break;
    case 1: launch_cfg<1, 2, 2, 2, false, EPI_STORE>(a, 1, 0, s, batch); break;
    case 2: launch_cfg<2, 1, 2, 2, false, EPI_STORE>(a, 1, 0, s, batch); break;
    default: launch_cfg<1, 1, 2, 2, false, EPI_STORE>(a, 1, 0, s, batch); break;
  }
  return check_launch("rows_gemm_batched");
}
