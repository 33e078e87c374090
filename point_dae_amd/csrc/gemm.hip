// gemm.hip -- fp32 GEMMs on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32),
// with the producer / epilogue fusions the patch embedder needs.
//
// The reference runs its dense layers (nn.Linear / 1x1 nn.Conv1d: patch
// embedder models/PointCAE_transformer.py:24-51, attention and MLP :113-137,
// :94-110, heads :329-333, :653-658) through cuBLAS/cuDNN in fp32, with
// BatchNorm / ReLU / max-pool / concat as separate full passes over
// (B*G*32, 512) tensors (537 MB each at B=128).  The fp32 MFMA of CDNA4 is exact
// fp32 (a k-ordered fmaf chain) at the fp32 vector peak (157 TFLOP/s), so the
// same arithmetic class is kept -- no TF32/bf16.
//
// Kernel "NT":  C[M,N] = epi( pro(A)[M,K] . B[N,K]^T )          (both K-contiguous)
//   - block tile BM x BN (256x256 with 16 waves, 128x384 with 12, or 128x128 with 4),
//     BK = 32; every wave owns 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulators/lane).
//     Measured on MI355X (B=128 shapes): 256x256 tiles 112-126
//     TFLOP/s (0.70 MFMA-busy), hipBLASLt 129-139 on the same shapes.
//   - The k index inside an 8-deep slab is permuted so that one ds_read_b128 per
//     operand row feeds four MFMAs: lane (r = l&31, h = l>>5) supplies
//     k = 8s + 4h + t to MFMA t (same permutation on A and B).  LDS rows are
//     padded to 36 floats: the four 16-lane groups of a ds_read_b128 then hit 64
//     distinct banks.
//   - global -> registers -> LDS staging, two LDS buffers, ONE barrier per
//     k-tile.  The issue order inside a k-tile is pinned with sched_group_barrier:
//     the loads of tile t+1 behind the first MFMAs of tile t, one fragment read of
//     the next 8-deep group behind every four MFMAs, the LDS stores of tile t+1
//     behind the last ones.
//   - persistent blocks: one residency of the chip; a block walks the tiles
//     slot, slot + nslots, ... of its XCD's contiguous chunk (blocks b, b+8, ...
//     share an L2; n fastest), and the slabs of consecutive tiles form one stream
//     through the LDS double buffer, so a tile boundary costs only the epilogue.
//   producers (applied to A while it is staged):
//     PRO_BNRELU   a := max(0, a * scale[k] + shift[k])   -- BatchNorm + ReLU of
//                  the previous layer never materialised
//   epilogues:
//     EPI_BIAS[_RELU|_GELU]  C = act(acc + bias[n])
//     EPI_GROUPBIAS_STATS    C = acc + gbias[m/32][n]; per-column sum / sum of
//                            squares of C accumulated into stats (BatchNorm
//                            batch statistics without another pass over C)
//     EPI_GROUPMAX           out[m/32][n] = max over the 32 rows of a group of
//                            (acc + bias[n]), arg[m/32][n] = first row attaining
//                            it; C itself is never written (max-pool fused)
//     EPI_STORE_GROUPMAX     both C = acc + bias and the group max / argmax
// Kernel "TN":  C[N,K] = A[M,N]^T . B[M,K]   (weight gradients; reduction over
//     the slow index M, split across blocks -- exactly one residency of the chip,
//     all tiles of a split on one XCD -- fp32 atomics into C; optional producer on
//     B, group-list row gather, fused column sums of A = the bias gradient).
#include <type_traits>
#include <cstdlib>

#include "common.h"
#include "nt_args.h"

namespace pdae {
int gemm_arith_rows();                                   // rows_gemm.hip: PDAE_GEMM_BF16X3 unless the fp32-input kernels were asked for
namespace rows3 {
bool conv3_takes(const NtArgs& a, int pro, int epi);    // rows3_gemm.hip: the same contracts on exact-split bf16
void launch_conv3(NtArgs& a, int pro, int epi, hipStream_t s);
}  // namespace rows3
}  // namespace pdae

namespace pdae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBK = 32, GLD = GBK + 4;

__device__ __forceinline__ float act_relu(float v) { return v > 0.f ? v : 0.f; }
__device__ __forceinline__ float act_gelu(float v) {
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

template <int BM, int BN, int PRO, int EPI>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void gemm_nt_kernel(const NtArgs p) {
  constexpr int WN = BN / 64;
  constexpr int NT = (BM / 64) * (BN / 64) * 64;
  constexpr int LA = (BM * 8 + NT - 1) / NT, LB = (BN * 8 + NT - 1) / NT;  // float4 per thread
  extern __shared__ float lds[];    // [2][(BM + BN) * GLD]
  // Persistent blocks: block (xcd, slot) walks tiles slot, slot + nslots, ... of its XCD's
  // chunk.  The slabs of consecutive tiles form ONE stream through the double-buffered LDS:
  // while a tile's last slab is multiplied the next tile's first slab is loaded and stored,
  // so a tile boundary costs the epilogue only -- no block relaunch, no exposed first load
  // (measured before: ~12-17 us of fixed cost per 256x256 tile, 20 % of conv3's time).
  const int chunk = (p.tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, nslots = gridDim.x >> 3;
  int slot = blockIdx.x >> 3;
  auto tile_of = [&](int sl) {
    const int t = xcd * chunk + sl;
    return (sl < chunk && t < p.tiles) ? t : -1;
  };
  int tile = tile_of(slot);
  if (tile < 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int r_ = r, h_ = h, wm_ = wm, wn_ = wn;
  const int scol = (tid & 7) * 4;
  const int M = p.M, N = p.N, K = p.K;

  // staging slot i of this thread is tile row (tid + i*NT) / 8; rows past the
  // matrix edge are clamped (their results are never stored), slots past the
  // tile (only when NT does not divide the tile) are skipped
  int m0 = 0, n0 = 0;
  // 32-bit element offsets from the (scalar) operand bases: one VGPR per staged row and the
  // base + offset addressing form of global_load (the launcher checks the operands fit)
  unsigned arow[LA], brow[LB];
  auto set_tile = [&](int t) {
    m0 = (t / p.tiles_n) * BM, n0 = (t % p.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int m = min(m0 + ((tid + i * NT) >> 3), M - 1);
      if (p.a_groups) m = p.a_groups[m >> 5] * 32 + (m & 31);   // gather whole 32-row groups
      arow[i] = (unsigned)m * (unsigned)p.lda + scol;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i)
      brow[i] = (unsigned)min(n0 + ((tid + i * NT) >> 3), N - 1) * (unsigned)p.ldb + scol;
  };
  set_tile(tile);
  auto a_ok = [&](int i) { return (BM * 8) % NT == 0 || ((tid + i * NT) >> 3) < BM; };
  auto b_ok = [&](int i) { return (BN * 8) % NT == 0 || ((tid + i * NT) >> 3) < BN; };

  float4 ra[LA], rb[LB];
  auto gload = [&](int kt) {
    const int k = kt * GBK;
    if (k + GBK <= K) {
#pragma unroll
      for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const float4*>(p.A + (arow[i] + (unsigned)k));
#pragma unroll
      for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const float4*>(p.B + (brow[i] + (unsigned)k));
    } else {  // partial last k-tile (K % 4 == 0): zero-fill
      const bool in = k + scol < K;
#pragma unroll
      for (int i = 0; i < LA; ++i)
        ra[i] = in ? *reinterpret_cast<const float4*>(p.A + (arow[i] + (unsigned)k)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < LB; ++i)
        rb[i] = in ? *reinterpret_cast<const float4*>(p.B + (brow[i] + (unsigned)k)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (PRO == PRO_BNRELU) {
      const int kk = k + scol;
      if (kk < K) {
        const float4 sc = *reinterpret_cast<const float4*>(p.pro_scale + kk);
        const float4 sh = *reinterpret_cast<const float4*>(p.pro_shift + kk);
#pragma unroll
        for (int i = 0; i < LA; ++i) {
          ra[i].x = act_relu(ra[i].x * sc.x + sh.x);
          ra[i].y = act_relu(ra[i].y * sc.y + sh.y);
          ra[i].z = act_relu(ra[i].z * sc.z + sh.z);
          ra[i].w = act_relu(ra[i].w * sc.w + sh.w);
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    float* As = lds + buf * (BM + BN) * GLD;
    float* Bs = As + BM * GLD;
#pragma unroll
    for (int i = 0; i < LA; ++i)
      if (a_ok(i)) *reinterpret_cast<float4*>(As + ((tid + i * NT) >> 3) * GLD + scol) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i)
      if (b_ok(i)) *reinterpret_cast<float4*>(Bs + ((tid + i * NT) >> 3) * GLD + scol) = rb[i];
  };

  f32x16 acc[2][2];
  const int KT = (K + GBK - 1) / GBK;
  gload(0);
  lstore(0);
  __syncthreads();
  int buf = 0;
  while (true) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // EPI_GROUPBIAS_STATS: the accumulators START at the group's bias (a 32-row MFMA tile is one group,
      // a lane holds one column of it: one value for its 16 registers) -- the epilogue then has no add and
      // four live registers fewer at the kernel's 128-VGPR limit
      float init = 0.f;
      if (EPI == EPI_GROUPBIAS_STATS || EPI == EPI_GROUP_SCATTER) {
        const int gr = m0 + wm * 64 + i * 32, gc = n0 + wn * 64 + j * 32 + r;
        if (gr < M && gc < N && (EPI == EPI_GROUPBIAS_STATS || p.gbias)) init = p.gbias[(size_t)(gr >> 5) * N + gc];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = init;
    }
  int em0 = m0, en0 = n0, next = -1;   // the tile being multiplied (m0/n0 move on to the next one early)
  for (int kt = 0; kt < KT; ++kt) {
    bool more = kt + 1 < KT;
    if (more) {
      gload(kt + 1);
    } else {
      next = tile_of(slot + nslots);
      // (opaque: otherwise the next tile's address arithmetic is hoisted above the k loop and
      //  spilled across it)
      asm volatile("" : "+s"(next));
      if (next >= 0) {
        set_tile(next);
        gload(0);
        more = true;
      }
    }
    const float* As = lds + buf * (BM + BN) * GLD + (wm * 64 + r) * GLD + 4 * h;
    const float* Bs = lds + buf * (BM + BN) * GLD + BM * GLD + (wn * 64 + r) * GLD + 4 * h;
    // fragment reads run one 8-deep slab ahead of the MFMAs that consume them
    float4 a[2][2], b[2][2];
    a[0][0] = *reinterpret_cast<const float4*>(As);
    a[0][1] = *reinterpret_cast<const float4*>(As + 32 * GLD);
    b[0][0] = *reinterpret_cast<const float4*>(Bs);
    b[0][1] = *reinterpret_cast<const float4*>(Bs + 32 * GLD);
#pragma unroll
    for (int s = 0; s < GBK / 8; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < GBK / 8) {
        a[nxt][0] = *reinterpret_cast<const float4*>(As + (s + 1) * 8);
        a[nxt][1] = *reinterpret_cast<const float4*>(As + 32 * GLD + (s + 1) * 8);
        b[nxt][0] = *reinterpret_cast<const float4*>(Bs + (s + 1) * 8);
        b[nxt][1] = *reinterpret_cast<const float4*>(Bs + 32 * GLD + (s + 1) * 8);
      }
#ifdef PDAE_NT_NO_SGB
      // keep the compiler from sinking those reads back to their first use
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].x, b[cur][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].y, b[cur][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].z, b[cur][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].w, b[cur][j].w, acc[i][j], 0, 0, 0);
        }
#ifndef PDAE_NT_NO_SGB
      // issue order: one fragment read of the next k-group behind every four MFMAs (the reads
      // then never queue up in front of an MFMA that needs them; measured 99 -> 104 TFLOP/s in
      // tools/lab/NOTES.md), the next slab's global loads behind the first MFMAs
      if (s == 0) __builtin_amdgcn_sched_group_barrier(0x020, LA + LB, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (s + 1 < GBK / 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else __builtin_amdgcn_sched_group_barrier(0x200, (LA + LB + 3) / 4, 0);   // the LDS stores of the next slab
      }
#endif
      // the next slab goes to the other LDS buffer (free since the last barrier) while this
      // slab's last MFMAs run, not after them
      if (s == GBK / 8 - 2 && more) lstore(buf ^ 1);
    }
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31,
  // row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): a lane holds 16 rows of ONE column.
  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
  // FULL = the whole tile lies inside the matrix (the only case on the step's shapes): no
  // per-element guards, one 64-bit base pointer per MFMA tile and 32-bit row offsets (the
  // guarded form spent ~6 us of VALU address arithmetic and 64 branches per tile)
  auto epilogue = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    const unsigned ldc = (unsigned)p.ldc;
    // the lane coordinates pass through an opaque asm so that nothing of the epilogue's
    // address arithmetic is hoisted out of the persistent loop (it would be live, i.e.
    // spilled, across the whole MFMA loop: the 256x256 tile runs at the 128-VGPR limit)
    int r = r_, h = h_, wm = wm_, wn = wn_;
    asm volatile("" : "+v"(r), "+v"(h), "+v"(wm), "+v"(wn));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = en0 + wn * 64 + j * 32 + r;
      const bool colok = FULL || col < N;
      const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rbase = em0 + wm * 64 + i * 32;  // one 32-row group per MFMA tile
        const float add = (EPI == EPI_GROUPBIAS_STATS || EPI == EPI_GROUP_SCATTER) ? 0.f : bv;
        float vmax = -__builtin_huge_valf();
        int amax = 0;
        float* cbase = (EPI != EPI_GROUPMAX) ? p.C + (size_t)(rbase + 4 * h) * ldc + col : nullptr;
        if (EPI == EPI_GROUP_SCATTER && (FULL || rbase < M))      // the 32-row MFMA tile is one group: one id
          cbase = p.C + (size_t)(p.c_groups[rbase >> 5] * 32 + 4 * h) * ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int lr = (e & 3) + 8 * (e >> 2) + 4 * h;
          const int row = rbase + lr;
          float v = acc[i][j][e] + add;
          if (EPI == EPI_BIAS_RELU) v = act_relu(v);
          if (EPI == EPI_BIAS_GELU) v = act_gelu(v);
          if ((EPI == EPI_GROUPBIAS_STATS || EPI == EPI_STATS) && (FULL || row < M)) {
            csum[j] += v;
            csq[j] += v * v;
          }
          if (EPI == EPI_GROUPMAX || EPI == EPI_STORE_GROUPMAX) {
            if (v > vmax) {  // e ascending => lr ascending within this half
              vmax = v;
              amax = lr;
            }
          }
          if (EPI != EPI_GROUPMAX && (FULL || (colok && row < M)))
            cbase[(unsigned)((e & 3) + 8 * (e >> 2)) * ldc] = v;
        }
        if (EPI == EPI_GROUPMAX || EPI == EPI_STORE_GROUPMAX) {
          // the other 16 rows of the group live in lane ^ 32
          const float ov = __shfl_xor(vmax, 32, kWave);
          const int oa = __shfl_xor(amax, 32, kWave);
          const bool take = (ov > vmax) || (ov == vmax && oa < amax);
          if (take) {
            vmax = ov;
            amax = oa;
          }
          if (h == 0 && colok && (FULL || rbase < M)) {
            p.gmax[(size_t)(rbase >> 5) * N + col] = vmax;
            p.garg[(size_t)(rbase >> 5) * N + col] = (unsigned char)amax;
          }
        }
      }
    }
  };
  if (em0 + BM <= M && en0 + BN <= N)
    epilogue(std::true_type{});
  else
    epilogue(std::false_type{});
  if (EPI == EPI_GROUPBIAS_STATS || EPI == EPI_STATS) {
    // per-tile column sums: one LDS slot per row of waves (plain stores), added in wave order,
    // then one atomic per column into the partial buffer of this block's XCD slot -- or, in
    // deterministic mode, a plain store into row (tile row) of p.stats_det
    constexpr int WM = BM / 64;
    float* red = lds + 2 * (BM + BN) * GLD;  // [WM][2][BN], behind the slab buffers (which already
                                             // hold the next tile's first slab)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = csum[j] + __shfl_xor(csum[j], 32, kWave);
      float q = csq[j] + __shfl_xor(csq[j], 32, kWave);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + wn * 64 + j * 32 + r] = s;
        red[(wm * 2 + 1) * BN + wn * 64 + j * 32 + r] = q;
      }
    }
    __syncthreads();
    float* dst = p.stats_det ? p.stats_det + (size_t)(em0 / BM) * 2 * N : p.stats + (size_t)(blockIdx.x & 7) * 2 * N;
    for (int c = tid; c < 2 * BN; c += NT) {
      const int half = c / BN, cc = c - half * BN;
      if (en0 + cc < N) {
        float t = red[half * BN + cc];
#pragma unroll
        for (int k = 1; k < WM; ++k) t += red[(k * 2 + half) * BN + cc];
        if (p.stats_det) dst[half * N + en0 + cc] = t;
        else atomicAdd(dst + half * N + en0 + cc, t);
      }
    }
    __syncthreads();   // red is written again by the next tile's epilogue
  }
  if (next < 0) break;
  slot += nslots;
  tile = next;
  }   // persistent tile loop
}

// ---------------------------------------------------------------------------
// C[N,K] += sum_m A[m,n] B[m,k] over this block's slice of M.  Tiles are read
// "down the columns": lane (r,h) of MFMA step t takes A[m = 2t + h][n = i0 + r].
// Optional producer on B (PRO_BNRELU over B's columns k) recomputes the
// activation that was never stored.
#ifndef PDAE_TBK
#define PDAE_TBK 16
#endif
constexpr int TBK = PDAE_TBK;
constexpr int TN_RES = TBK == 16 ? 3 : 2;   // TN blocks resident per CU (LDS: 32 KB / 64 KB each)
struct TnArgs {
  int M, N, K;
  const float* A;
  int lda;
  const float* B;
  int ldb;
  float* C;
  int ldc;
  const float* pro_scale;  // [K]
  const float* pro_shift;
  const int* b_groups;     // nullable: row m of B is source row b_groups[m/32]*32 + m%32
  const int* a_groups;     // nullable: the same for A
  float* colsum_a;         // nullable: [N] += column sums of A (the bias gradient), from the k-tile-0 blocks
  int rows_per_split;
  int tk, tn, splits;      // tiles along K and N, M-splits
  float* part_c;           // deterministic mode: [splits][N][K] plain-store partials (else null)
  float* part_s;           //                     [splits][N] partials of colsum_a
};

// Block tile TM (columns n of A) x TN_ (columns k of B); every wave 64x64.  Bigger
// tiles = fewer bytes staged per MAC (the 128x128 version ran at 69 TFLOP/s).
template <int TM, int TN_, int PRO>
__global__ __launch_bounds__((TM / 64) * (TN_ / 64) * 64) __attribute__((amdgpu_waves_per_eu(TN_RES, TN_RES)))
void gemm_tn_kernel(const TnArgs p) {
  constexpr int WN = TN_ / 64;
  constexpr int NT = (TM / 64) * (TN_ / 64) * 64;
  constexpr int ROW4 = (TM + TN_) / 4;              // float4 per staged row
  constexpr int SLOTS = (TBK * ROW4 + NT - 1) / NT;  // float4 per thread per tile
  __shared__ float lds[2][TBK * (TM + TN_)];
  // XCD-aware order (1-D grid; blocks b, b+8, ... share an L2): all tn*tk tiles of one
  // M-split are consecutive slots of ONE XCD, so they run together and the tk-fold re-read
  // of the A band and the tn-fold re-read of the B band hit that L2 instead of HBM
  // (PMC before: dW3 fetched 2.2 GB for 0.8 GB of operands).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tiles = p.tk * p.tn;
  const int tile = slot % tiles, split = (slot / tiles) * 8 + xcd;
  if (split >= p.splits) return;
  const int bx = tile % p.tk, by = tile / p.tk;
  const int n0 = by * TM, k0 = bx * TN_;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  if (mbeg >= mend) return;
  const int N = p.N, K = p.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  // NT is a multiple of ROW4, so a thread stages the SAME four columns in every slot
  // (one scale/shift float4, not one per slot: 56 VGPRs less, three blocks per CU)
  static_assert(NT % ROW4 == 0 && (TBK * ROW4) % NT == 0, "slot layout");
  constexpr int RSTEP = NT / ROW4;
  const int srow0 = tid / ROW4, scol = (tid % ROW4) * 4;
  const bool isb = scol >= TM;
  const int gcol = isb ? k0 + scol - TM : n0 + scol;
  const bool ok = isb ? gcol < K : gcol < N;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PRO == PRO_BNRELU && isb && ok) {
    sc = *reinterpret_cast<const float4*>(p.pro_scale + gcol);
    sh = *reinterpret_cast<const float4*>(p.pro_shift + gcol);
  }
  const float* src = isb ? p.B + gcol : p.A + gcol;
  const int ld = isb ? p.ldb : p.lda;
  float4 rg[SLOTS];
  bool rin[SLOTS];
  float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of this thread's A elements
  const bool sum_a = p.colsum_a != nullptr && bx == 0 && !isb;
  static_assert(32 % TBK == 0, "a staged slab lies inside one 32-row group");
  auto gload = [&](int mt) {
    // a slab is TBK consecutive rows starting at a multiple of TBK: inside ONE group of the
    // list -> one uniform (scalar) id load per slab instead of a dependent load per slot
    const int shift = p.b_groups ? (__builtin_amdgcn_readfirstlane(p.b_groups[mt >> 5]) * 32 - (mt & ~31)) : 0;
    const int shift_a = p.a_groups ? (__builtin_amdgcn_readfirstlane(p.a_groups[mt >> 5]) * 32 - (mt & ~31)) : 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int gm = mt + srow0 + i * RSTEP;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && gm < mend) {
        const int sm = isb ? gm + shift : gm + shift_a;
        v = *reinterpret_cast<const float4*>(src + (size_t)sm * ld);
      }
      rg[i] = v;
      rin[i] = ok && gm < mend;
    }
  };
  // producer math and the bias-gradient sums run here, AFTER the MFMA loop: the loads issued
  // by gload stay in flight behind the MFMAs instead of being waited for right away
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      float4 v = rg[i];
      if (PRO == PRO_BNRELU && isb && rin[i]) {
        v.x = act_relu(v.x * sc.x + sh.x);
        v.y = act_relu(v.y * sc.y + sh.y);
        v.z = act_relu(v.z * sc.z + sh.z);
        v.w = act_relu(v.w * sc.w + sh.w);
      }
      if (sum_a) asum.x += v.x, asum.y += v.y, asum.z += v.z, asum.w += v.w;
      *reinterpret_cast<float4*>(&lds[buf][(srow0 + i * RSTEP) * (TM + TN_) + scol]) = v;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gload(mbeg);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int mt = mbeg; mt < mend; mt += TBK) {
    if (mt + TBK < mend) gload(mt + TBK);
    const float* T = lds[buf] + h * (TM + TN_);
    // fragment reads run two k-steps ahead of the MFMAs that consume them
    float fa[2][2][2], fb[2][2][2];   // [stage][step within stage][tile]
    auto fread = [&](int st, int t) {
      const float* row = T + 2 * t * (TM + TN_);
      fa[st][t & 1][0] = row[wm * 64 + r];
      fa[st][t & 1][1] = row[wm * 64 + 32 + r];
      fb[st][t & 1][0] = row[TM + wn * 64 + r];
      fb[st][t & 1][1] = row[TM + wn * 64 + 32 + r];
    };
    fread(0, 0);
    fread(0, 1);
#pragma unroll
    for (int tt = 0; tt < TBK / 4; ++tt) {
      const int cur = tt & 1, nxt = cur ^ 1;
      if (tt + 1 < TBK / 4) {
        fread(nxt, 2 * tt + 2);
        fread(nxt, 2 * tt + 3);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][u][0], fb[cur][u][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][u][0], fb[cur][u][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][u][1], fb[cur][u][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][u][1], fb[cur][u][1], acc[1][1], 0, 0, 0);
      }
      // issue order (see gemm_nt_kernel): the next slab's global loads behind the first MFMAs,
      // one fragment read behind every MFMA, the LDS stores of the next slab behind the last ones
      if (tt == 0) __builtin_amdgcn_sched_group_barrier(0x020, SLOTS, 0);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (tt + 1 < TBK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (g < SLOTS) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      if (tt == TBK / 4 - 2 && mt + TBK < mend) lstore(buf ^ 1);
    }
    __syncthreads();
    buf ^= 1;
  }
  if (p.colsum_a != nullptr && bx == 0) {
    // the RSTEP threads that staged the same four columns meet in LDS (free after the loop's
    // last barrier) and are added in row-phase order
    float4* ls = reinterpret_cast<float4*>(&lds[0][0]);   // [RSTEP][TM / 4]
    if (!isb) ls[srow0 * (TM / 4) + scol / 4] = asum;
    __syncthreads();
    if (!isb && srow0 == 0 && ok) {
      float4 t = ls[scol / 4];
#pragma unroll
      for (int k = 1; k < RSTEP; ++k) {
        const float4 u = ls[k * (TM / 4) + scol / 4];
        t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
      }
      if (p.part_s) {
        *reinterpret_cast<float4*>(p.part_s + (size_t)split * N + gcol) = t;
      } else {
        atomicAdd(p.colsum_a + gcol + 0, t.x), atomicAdd(p.colsum_a + gcol + 1, t.y);
        atomicAdd(p.colsum_a + gcol + 2, t.z), atomicAdd(p.colsum_a + gcol + 3, t.w);
      }
    }
  }
  float* pc = p.part_c ? p.part_c + (size_t)split * N * K : nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wn * 64 + j * 32 + r;
    if (col >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < N) {
          if (pc) pc[(size_t)row * K + col] = acc[i][j][e];
          else atomicAdd(p.C + (size_t)row * p.ldc + col, acc[i][j][e]);
        }
      }
    }
  }
}

template <int BM, int BN, int PRO, int EPI>
static void launch_nt_cfg(NtArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM;
  a.tile_rows = tiles_m;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.tiles = tiles_m * a.tiles_n;
  constexpr int NTH = (BM / 64) * (BN / 64) * 64;
  const size_t lds = (2 * (size_t)(BM + BN) * GLD + 2 * BN * (BM / 64)) * sizeof(float);
  // persistent blocks: one residency of the chip (32 CUs per XCD x blocks that fit a CU's LDS)
  const int chunk = (a.tiles + 7) / 8;
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  const int nslots = chunk < 32 * per_cu ? chunk : 32 * per_cu;
  const int grid = 8 * nslots;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<BM, BN, PRO, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, PRO, EPI>), dim3(grid), dim3(NTH), lds, s, a);
}

// 256x256 tiles when they fill the chip (>= 3/4 of the 256 CUs busy in the last
// wave of blocks is not worth modelling: big problems only), else 128x128.
template <int PRO, int EPI>
static bool launch_conv3_if(NtArgs& a, hipStream_t s) {
  if (gemm_arith_rows() != PDAE_GEMM_BF16X3 || !rows3::conv3_takes(a, PRO, EPI)) return false;
  rows3::launch_conv3(a, PRO, EPI, s);
  return true;
}

template <int PRO, int EPI>
static int launch_nt(NtArgs& a, hipStream_t s) {
  if (launch_conv3_if<PRO, EPI>(a, s)) return check_launch("conv3");
  const long long big_tiles = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  if (big_tiles >= 512 && a.N % 256 != 0 && a.N % 384 == 0) launch_nt_cfg<128, 384, PRO, EPI>(a, s);
  else if (big_tiles >= 512) launch_nt_cfg<256, 256, PRO, EPI>(a, s);
  else launch_nt_cfg<128, 128, PRO, EPI>(a, s);
  return check_launch("gemm_nt");
}

static int check_nt(const char* who, int M, int N, int K) {
  if (M < 0 || N <= 0 || K <= 0) return bad_arg(who);
  if (K % 4 != 0) return unsupported("gemm: the reduction length must be a multiple of 4");
  // rows are addressed with 32-bit element offsets (lda, ldb <= max(K, N) here)
  if ((long long)M * (K > N ? K : N) >= (1LL << 32) || (long long)N * K >= (1LL << 32))
    return unsupported("gemm: operands of 2^32 elements or more");
  return PDAE_OK;
}

}  // namespace pdae

using namespace pdae;

extern "C" int pdae_linear_forward(int M, int N, int K, const float* X, const float* W,
                                   const float* bias, int act, float* Y, pdae_stream_t stream) {
  int rc = check_nt("linear_forward: bad size", M, N, K);
  if (rc) return rc;
  if (act < 0 || act > 2) return bad_arg("linear_forward: act must be 0, 1 or 2");
  if (M == 0) return PDAE_OK;
  if (!X || !W || !Y) return bad_arg("linear_forward: null pointer");
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = N, a.bias = bias;
  hipStream_t s = as_stream(stream);
  if (act == 0) return launch_nt<PRO_NONE, EPI_BIAS>(a, s);
  if (act == 1) return launch_nt<PRO_NONE, EPI_BIAS_RELU>(a, s);
  return launch_nt<PRO_NONE, EPI_BIAS_GELU>(a, s);
}

extern "C" int pdae_linear_backward_data(int M, int N, int K, const float* dY, const float* Wt,
                                         float* dX, pdae_stream_t stream) {
  // dX[M,K] = dY[M,N] . W[N,K]  ==  NT product with W^T [K,N] (N contiguous)
  int rc = check_nt("linear_backward_data: bad size", M, K, N);
  if (rc) return rc;
  if (M == 0) return PDAE_OK;
  if (!dY || !Wt || !dX) return bad_arg("linear_backward_data: null pointer");
  NtArgs a = {};
  a.M = M, a.N = K, a.K = N, a.A = dY, a.lda = N, a.B = Wt, a.ldb = N, a.C = dX, a.ldc = K;
  return launch_nt<PRO_NONE, EPI_BIAS>(a, as_stream(stream));
}

template <int TM, int TN_>
static int launch_tn_cfg(TnArgs& t, bool bnrelu, hipStream_t s) {
  const int tn = (t.N + TM - 1) / TM, tk = (t.K + TN_ - 1) / TN_;
  constexpr int NTH = (TM / 64) * (TN_ / 64) * 64;
  // M-splits so that the grid is ONE full residency of the chip (256 CUs x 3 blocks):
  // 768 blocks ran dW4 in 576 us, 1024 blocks (1.33 rounds) in 663 us; each split >= 256 rows
  int splits = (TN_RES * 256 + tn * tk / 2) / (tn * tk);
  if (splits < 1) splits = 1;
  int rows = (t.M + splits - 1) / splits;
  rows = ((rows + TBK - 1) / TBK) * TBK;
  if (rows < 256) rows = 256;
  splits = (t.M + rows - 1) / rows;
  t.rows_per_split = rows;
  t.tk = tk, t.tn = tn, t.splits = splits;
  const unsigned grid = 8u * ((splits + 7) / 8) * tk * tn;
  // deterministic mode: the splits store their tiles side by side and one pass adds them in
  // split order (the zero-filled dW / dbias receive the sums)
  int rc = PDAE_OK;
  const size_t nk = (size_t)t.N * t.K;
  float* ws = static_cast<float*>(det_workspace(sizeof(float) * splits * (nk + t.N), &rc));
  if (rc) return rc;
  if (ws) t.part_c = ws, t.part_s = ws + splits * nk;
  if (bnrelu)
    hipLaunchKernelGGL((gemm_tn_kernel<TM, TN_, PRO_BNRELU>), dim3(grid), dim3(NTH), 0, s, t);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<TM, TN_, PRO_NONE>), dim3(grid), dim3(NTH), 0, s, t);
  if (ws) {
    if (t.ldc != t.K) return unsupported("deterministic mode: weight gradients with a dense leading dimension only");
    if ((rc = det_reduce(s, splits, (int)nk, t.part_c, t.C, (int)nk))) return rc;
    if (t.colsum_a && (rc = det_reduce(s, splits, t.N, t.part_s, t.colsum_a, t.N))) return rc;
  }
  return PDAE_OK;
}

static int launch_tn(TnArgs& t, bool bnrelu, hipStream_t s) {
  // measured on the embedder's weight gradients (M = 262144): the 128x128 tile at two
  // blocks per CU (654 us for dW4) beats 192x256 / 256x256 at one block per CU (767 us)
  const int rc = launch_tn_cfg<128, 128>(t, bnrelu, s);
  return rc ? rc : check_launch("gemm_tn");
}

extern "C" int pdae_linear_backward_weight(int M, int N, int K, const float* dY, const float* X,
                                           float* dW, float* dbias, pdae_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return bad_arg("linear_backward_weight: bad size");
  if (!dW) return bad_arg("linear_backward_weight: null pointer");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, s);
  if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)N, s);
  if (M == 0) return check_launch("linear_backward_weight");
  if (!dY || !X) return bad_arg("linear_backward_weight: null pointer");
  if (N % 4 != 0 || K % 4 != 0) return unsupported("linear_backward_weight: N, K multiples of 4");
  TnArgs t = {};
  t.M = M, t.N = N, t.K = K, t.A = dY, t.lda = N, t.B = X, t.ldb = K, t.C = dW, t.ldc = K;
  t.colsum_a = dbias;      // the bias gradient rides on the A tiles the k-tile-0 blocks stage anyway
  return launch_tn(t, false, s);
}

// ---- fused patch-embedder layers (models/PointCAE_transformer.py:37-51) ------
extern "C" int pdae_embed_conv_store_groupmax(int M, int N, int K, const float* X, const float* W,
                                              const float* bias, float* Y, float* gmax,
                                              unsigned char* garg, pdae_stream_t stream) {
  int rc = check_nt("embed_conv_store_groupmax: bad size", M, N, K);
  if (rc) return rc;
  if (M % 32 != 0) return bad_arg("embed_conv_store_groupmax: M must be a multiple of 32");
  if (M == 0) return PDAE_OK;
  if (!X || !W || !Y || !gmax || !garg) return bad_arg("embed_conv_store_groupmax: null pointer");
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = N, a.bias = bias;
  a.gmax = gmax, a.garg = garg;
  return launch_nt<PRO_NONE, EPI_STORE_GROUPMAX>(a, as_stream(stream));
}

extern "C" int pdae_embed_conv_groupbias_stats(int M, int N, int K, const float* X, const float* W,
                                               const float* gbias, float* Y, float* stats,
                                               pdae_stream_t stream) {
  int rc = check_nt("embed_conv_groupbias_stats: bad size", M, N, K);
  if (rc) return rc;
  if (M % 32 != 0) return bad_arg("embed_conv_groupbias_stats: M must be a multiple of 32");
  if (!stats) return bad_arg("embed_conv_groupbias_stats: null pointer");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(stats, 0, sizeof(float) * 16 * (size_t)N, s);
  if (M == 0) return check_launch("embed_conv_groupbias_stats");
  if (!X || !W || !Y || !gbias) return bad_arg("embed_conv_groupbias_stats: null pointer");
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = N;
  a.gbias = gbias, a.stats = stats;
  // deterministic mode: one partial row per tile row (128-row tiles at the least), added in
  // row order into slot 0 of `stats`
  const int max_tile_rows = (M + 127) / 128;
  a.stats_det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)max_tile_rows * 2 * N, &rc));
  if (rc) return rc;
  rc = launch_nt<PRO_NONE, EPI_GROUPBIAS_STATS>(a, s);
  if (rc || !a.stats_det) return rc;
  return det_reduce(s, a.tile_rows, 2 * N, a.stats_det, stats, 2 * N);
}

// The first layer of a set-abstraction level on coordinates alone (K = 4: xyz - centre and the zero pad column; sa1 of
// Point_CAE_PointNetv2: 2.1 M rows x 64 channels): 8 FLOPs per stored float, nothing for a matrix pipe to do -- the pass is
// its (M, N) store.  Thread = 4 adjacent channels of one row phase, weights in registers, y = fma(x3, w3, fma(x2, w2, fma(x1, w1, x0 w0)))
// (the k-ascending FMA chain of a CPU sgemm micro-kernel: what the reference's conv computes on the host);
// statistics as in the GEMM epilogue (float atomics into the 8 slots, or one partial row per block in deterministic mode).
constexpr int K4_ROWS = 2048;   // rows per block
__global__ __launch_bounds__(256) void conv_k4_stats_kernel(int M, int N, const float4* __restrict__ x, const float4* __restrict__ W,
                                                            float* __restrict__ y, float* __restrict__ stats,
                                                            float* __restrict__ det) {
  extern __shared__ float k4_red[];          // [phases][2][N]
  const int q = N >> 2, phases = 256 / q;
  const int cq = threadIdx.x % q, ph = threadIdx.x / q;
  const int c = cq * 4;
  float4 w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = W[c + j];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int mbeg = blockIdx.x * K4_ROWS, mend = min(M, mbeg + K4_ROWS);
  if (ph < phases) {
#pragma unroll 4
    for (int m = mbeg + ph; m < mend; m += phases) {
      const float4 xv = x[m];
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = __builtin_fmaf(xv.w, w[j].w, __builtin_fmaf(xv.z, w[j].z, __builtin_fmaf(xv.y, w[j].y, xv.x * w[j].x)));
        s1[j] += v[j];
        s2[j] += v[j] * v[j];
      }
      *reinterpret_cast<float4*>(y + (size_t)m * N + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      k4_red[(ph * 2 + 0) * N + c + j] = s1[j];
      k4_red[(ph * 2 + 1) * N + c + j] = s2[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * N; i += 256) {
    float t = 0.f;
    for (int p = 0; p < phases; ++p) t += k4_red[p * 2 * N + i];
    if (det) det[(size_t)blockIdx.x * 2 * N + i] = t;
    else atomicAdd(stats + (size_t)(blockIdx.x & 7) * 2 * N + i, t);
  }
}

// ---- one layer of a shared MLP (Conv 1x1, no bias -> BatchNorm -> ReLU; pointnet2_modules'
// SharedMLP): Y = act(X) . W^T with act = the PREVIOUS layer's BatchNorm + ReLU applied while X
// is staged (scale == null: X as is), and this layer's BatchNorm statistics from the epilogue.
extern "C" int pdae_conv_stats(int M, int N, int K, const float* X, const float* scale, const float* shift,
                               const float* W, float* Y, float* stats, pdae_stream_t stream) {
  int rc = check_nt("conv_stats: bad size", M, N, K);
  if (rc) return rc;
  if (!stats) return bad_arg("conv_stats: null pointer");
  if ((scale == nullptr) != (shift == nullptr)) return bad_arg("conv_stats: scale and shift go together");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(stats, 0, sizeof(float) * 16 * (size_t)N, s);
  if (M == 0) return check_launch("conv_stats");
  if (!X || !W || !Y) return bad_arg("conv_stats: null pointer");
  static const bool k4_off = [] { const char* e = getenv("PDAE_CONV_K4"); return e && e[0] == '0'; }();     // (A/B)
  if (K == 4 && !scale && !k4_off && N % 4 == 0 && N <= 1024 && 256 % (N / 4) == 0 && !(((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15)) {
    const int blocks = (M + K4_ROWS - 1) / K4_ROWS, phases = 256 / (N / 4);
    float* det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)blocks * 2 * N, &rc));
    if (rc) return rc;
    hipLaunchKernelGGL(conv_k4_stats_kernel, dim3(blocks), dim3(256), sizeof(float) * phases * 2 * N, s, M, N,
                       reinterpret_cast<const float4*>(X), reinterpret_cast<const float4*>(W), Y, stats, det);
    rc = check_launch("conv_stats");
    if (rc || !det) return rc;
    return det_reduce(s, blocks, 2 * N, det, stats, 2 * N);
  }
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = N;
  a.pro_scale = scale, a.pro_shift = shift, a.stats = stats;
  const int max_tile_rows = (M + 127) / 128;
  a.stats_det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)max_tile_rows * 2 * N, &rc));
  if (rc) return rc;
  // narrow outputs (the first set-abstraction level: 64 / 128 channels): the 128-wide tile
  const bool narrow = N <= 128;
  if (scale ? launch_conv3_if<PRO_BNRELU, EPI_STATS>(a, s) : launch_conv3_if<PRO_NONE, EPI_STATS>(a, s)) {
  } else if (scale) {
    if (narrow) launch_nt_cfg<128, 128, PRO_BNRELU, EPI_STATS>(a, s);
    else launch_nt_cfg<256, 256, PRO_BNRELU, EPI_STATS>(a, s);
  } else {
    if (narrow) launch_nt_cfg<128, 128, PRO_NONE, EPI_STATS>(a, s);
    else launch_nt_cfg<256, 256, PRO_NONE, EPI_STATS>(a, s);
  }
  rc = check_launch("conv_stats");
  if (rc || !a.stats_det) return rc;
  return det_reduce(s, a.tile_rows, 2 * N, a.stats_det, stats, 2 * N);
}

extern "C" int pdae_embed_bnrelu_conv_groupmax(int M, int N, int K, const float* X,
                                               const float* scale, const float* shift,
                                               const float* W, const float* bias, float* gmax,
                                               unsigned char* garg, const int32_t* groups,
                                               pdae_stream_t stream) {
  int rc = check_nt("embed_bnrelu_conv_groupmax: bad size", M, N, K);
  if (rc) return rc;
  if (M % 32 != 0) return bad_arg("embed_bnrelu_conv_groupmax: M must be a multiple of 32");
  if (M == 0) return PDAE_OK;
  if (!X || !scale || !shift || !W || !gmax || !garg)
    return bad_arg("embed_bnrelu_conv_groupmax: null pointer");
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.bias = bias;
  a.pro_scale = scale, a.pro_shift = shift, a.gmax = gmax, a.garg = garg, a.a_groups = groups;
  return launch_nt<PRO_BNRELU, EPI_GROUPMAX>(a, as_stream(stream));
}

extern "C" int pdae_bnrelu_linear_backward_weight(int M, int N, int K, const float* dY,
                                                  const float* X, const float* scale,
                                                  const float* shift, float* dW, float* dbias,
                                                  const int32_t* groups, pdae_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return bad_arg("bnrelu_linear_backward_weight: bad size");
  if (!dW) return bad_arg("bnrelu_linear_backward_weight: null pointer");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, s);
  if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)N, s);
  if (M == 0) return check_launch("bnrelu_linear_backward_weight");
  if (!dY || !X || !scale || !shift) return bad_arg("bnrelu_linear_backward_weight: null pointer");
  if (N % 4 != 0 || K % 4 != 0) return unsupported("bnrelu_linear_backward_weight: N, K multiples of 4");
  TnArgs t = {};
  t.M = M, t.N = N, t.K = K, t.A = dY, t.lda = N, t.B = X, t.ldb = K, t.C = dW, t.ldc = K;
  t.pro_scale = scale, t.pro_shift = shift, t.b_groups = groups, t.colsum_a = dbias;
  if (groups && M % 32 != 0) return bad_arg("bnrelu_linear_backward_weight: M must be a multiple of 32 with a group list");
  return launch_tn(t, true, s);
}

// dW[N,K] = sum over listed rows of dY[rowA(m)]^T X[rowB(m)]: the TN kernel with whole 32-row groups gathered
// on either operand (a_groups / b_groups nullable: that operand is compact).  dY = X = f with the same list on
// both sides is the Gram matrix of the listed rows.
extern "C" int pdae_linear_backward_weight_listed(int M, int N, int K, const float* dY, const int32_t* a_groups,
                                                  const float* X, const int32_t* b_groups, float* dW,
                                                  float* dbias, pdae_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return bad_arg("linear_backward_weight_listed: bad size");
  if (!dW) return bad_arg("linear_backward_weight_listed: null pointer");
  if (M % 32 != 0) return bad_arg("linear_backward_weight_listed: M must be a multiple of 32 (whole groups)");
  hipStream_t s = as_stream(stream);
  (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)N * K, s);
  if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)N, s);
  if (M == 0) return check_launch("linear_backward_weight_listed");
  if (!dY || !X) return bad_arg("linear_backward_weight_listed: null pointer");
  if (N % 4 != 0 || K % 4 != 0) return unsupported("linear_backward_weight_listed: N, K multiples of 4");
  TnArgs t = {};
  t.M = M, t.N = N, t.K = K, t.A = dY, t.lda = N, t.B = X, t.ldb = K, t.C = dW, t.ldc = K;
  t.a_groups = a_groups, t.b_groups = b_groups, t.colsum_a = dbias;
  return launch_tn(t, false, s);
}

// Y[c_groups[m/32]*32 + m%32, :] = X[a_groups[m/32]*32 + m%32, :] . W[N,K]^T + gbias[m/32, :] for the M rows
// (whole 32-row groups) of a compact product: gather on the way in (a_groups nullable: X is compact), one bias
// row per group (nullable), scatter on the way out.  The other rows of Y are not touched.
extern "C" int pdae_group_gemm_scatter(int M, int N, int K, const float* X, const int32_t* a_groups,
                                       const float* W, const float* gbias, float* Y, int ldy,
                                       const int32_t* c_groups, pdae_stream_t stream) {
  int rc = check_nt("group_gemm_scatter: bad size", M, N, K);
  if (rc) return rc;
  if (M % 32 != 0) return bad_arg("group_gemm_scatter: M must be a multiple of 32 (whole groups)");
  if (M == 0) return PDAE_OK;
  if (!X || !W || !Y || !c_groups || ldy < N) return bad_arg("group_gemm_scatter: null pointer / bad ldy");
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = ldy;
  a.gbias = gbias, a.a_groups = a_groups, a.c_groups = c_groups;
  if (launch_conv3_if<PRO_NONE, EPI_GROUP_SCATTER>(a, as_stream(stream))) return check_launch("group_gemm_scatter");
  // (the 256-row tile from 32 k rows on: measured on the embedder's two calls)
  if (M >= 32768 && N % 256 == 0) {
    launch_nt_cfg<256, 256, PRO_NONE, EPI_GROUP_SCATTER>(a, as_stream(stream));
    return check_launch("group_gemm_scatter");
  }
  return launch_nt<PRO_NONE, EPI_GROUP_SCATTER>(a, as_stream(stream));
}

extern "C" int pdae_embed_bnrelu_conv_store_groupmax(int M, int N, int K, const float* X,
                                                     const float* scale, const float* shift,
                                                     const float* W, const float* bias, float* Y,
                                                     float* gmax, unsigned char* garg,
                                                     pdae_stream_t stream) {
  int rc = check_nt("embed_bnrelu_conv_store_groupmax: bad size", M, N, K);
  if (rc) return rc;
  if (M % 32 != 0) return bad_arg("embed_bnrelu_conv_store_groupmax: M must be a multiple of 32");
  if (M == 0) return PDAE_OK;
  if (!X || !scale || !shift || !W || !Y || !gmax || !garg)
    return bad_arg("embed_bnrelu_conv_store_groupmax: null pointer");
  NtArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A = X, a.lda = K, a.B = W, a.ldb = K, a.C = Y, a.ldc = N, a.bias = bias;
  a.pro_scale = scale, a.pro_shift = shift, a.gmax = gmax, a.garg = garg;
  return launch_nt<PRO_BNRELU, EPI_STORE_GROUPMAX>(a, as_stream(stream));
}
