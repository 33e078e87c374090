// rows3p_kernel.h -- the exact-split row GEMM (rows3_kernel.h: fp32 results from six bf16 x bf16 MFMA products per
// fp32 product, two accumulator sets) on operands whose three bf16 planes were written ONCE by their producer.
//
// rows3::gemm3_kernel splits every staged fp32 element inside its k-loop -- an activation element once per column tile
// that reads it, a weight element once per row tile -- 5.5 vector instructions per element beside the MFMAs, on their
// issue port.  Here an operand arrives as PLANES: P3(X) = bf16 [3][rows][ld], plane 0 = h = bf16(x), plane 1 = m =
// bf16(x - h), plane 2 = l = bf16(x - h - m) -- exactly what split_chunk computes, so a product of planes is bit for bit
// the product gemm3_kernel forms from the fp32 operands (tools/lab/p3_lab.py).  The k-loop then has no vector
// arithmetic at all: tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), the
// matrix pipe and the fragment reads are all a wave issues.
//
// LDS image of a 32-deep tile: [plane][row][64 B], rows UNPADDED (an LDS-DMA wave-instruction writes 64 lanes x 16 B
// linearly: 16 rows), conflict-free for the ds_read_b128 fragment reads through an XOR on the SOURCE side: 16-byte slot
// c' of row r holds k-octet c' ^ ((r >> 2) & 3) (the lane groups of a ds_read_b128 -- {0-3, 12-15, 20-27}, ... -- then
// cover sixteen different slots of the 256-byte bank row).  A ring of NBUF tiles (3 where 160 KB allows: two tiles in
// flight behind the one being multiplied); one barrier per tile, in front of it a COUNTED s_waitcnt vmcnt that leaves
// the younger tiles in flight.  The LDS-DMAs are inline asm (the compiler's wait-count pass would put vmcnt(0) in front
// of every ds_read behind a builtin one); no other vector memory instruction lives in the loop, so the count is exact.
#pragma once
#include "../../point_dae_amd/csrc/rows3_kernel.h"

namespace pdae {
namespace rows3p {

using rows::f32x16;
using rows3::bf16x8;
using rows3::mfma_one;
using rows3::static_for;
typedef unsigned short bf16_t;

struct PArgs {
  int M, N, K;
  const bf16_t* A3;      // planes of A: [3][M][lda], reduction contiguous
  long long planeA;      // elements between the planes of A
  int lda;
  const bf16_t* B3;      // planes of B: [3][N][ldb], reduction contiguous (a weight (out, in); or the planes of its
  long long planeB;      // transpose for a data gradient)
  int ldb;
  float* C;              // [M, N] fp32 (split-K: `slab` elements between the slabs); may be null when C3 is given
  int ldc;
  float* Z;              // as rows::Args
  const float* bias;
  bf16_t* C3;            // optional: the planes of the stored result, [3][M][ldc3]
  long long planeC;
  int ldc3;
  int tiles_n, tiles;
  int kchunk;
  long long slab;
#ifdef P3_STAMPS
  unsigned long long* stamps;   // (lab) [blocks][4]: s_memtime / s_memrealtime around the k-loop of wave 0
#endif
};

// one LDS-DMA wave-instruction: 64 lanes x 16 B from sbase + voff[lane] to LDS[m0 .. m0 + 1024)
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase)
               : "memory");
}

// C[M,N] = epi(A . B^T) from planes.  Block tile 128 x (64 TJ), eight waves of 32 x (32 TJ) (rows3::gemm3_kernel's wave
// tiles <1, TJ, 4, 2>), 32-deep LDS tiles.  Slots as in gemm3_kernel: one MFMA each, the side work pinned behind it.
//   slots of step 0          + the fragment reads of step 1 (same LDS tile)
//   slot SB - 1              + s_waitcnt vmcnt(tiles still allowed in flight) ; s_barrier   (tile t + 1 has landed, every
//                              wave has read tile t's last fragments)
//   slots [SB, S)            + the LDS-DMAs of tile t + NBUF into the buffer tile t occupied, the fragment reads of
//                              tile t + 1, step 0
template <int TJ, int EPI, int NBUF, int SCHED = 0, int ABL = 0>   // ABL (lab, SCHED 1): 1 no barrier, 2 no fragment reads, 4 no DMAs, 8 no vmcnt wait
__global__ __launch_bounds__(512) void gemm3p_kernel(const PArgs p) {
  constexpr int TI = 1, WN = 2, BM = 128, BN = 64 * TJ, G = TI * TJ;
  constexpr int ROWB = 64, BKT = 32;
  constexpr int RCH = (BM + BN) / 16;                  // 16-row chunks (one LDS-DMA each) per plane
  constexpr int NCH = 3 * RCH, CW = (NCH + 7) / 8;     // ... per tile, per wave
  constexpr int PLANE = (BM + BN) * ROWB, BUF = 3 * PLANE;
  constexpr bool DUAL = true;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  const int chunk = (p.tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = xcd * chunk + slot;
  if (slot >= chunk || tile >= p.tiles) return;
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  const int kbeg = blockIdx.y * p.kchunk, kend = min(p.K, kbeg + p.kchunk), piece = blockIdx.y;
  const int KT = (kend - kbeg) / BKT;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds3;

  // this wave's LDS-DMAs of a tile: chunk c = wave + 8 j (past the end: the last chunk again -- same bytes to the same
  // place); chunk -> plane c / RCH, rows 16 (c % RCH) of the [A rows | B rows] image
  const char* sbase[CW];
  unsigned voff[CW], ldst[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j) {
    const int c = min(wave + 8 * j, NCH - 1), pl = c / RCH, rc = c % RCH;
    const int lrow = lane >> 2, oc = (lane & 3) ^ ((lrow >> 2) & 3);     // slot lane & 3 of its row holds octet oc
    const bool isa = rc < BM / 16;
    const int row = isa ? min(m0 + 16 * rc + lrow, M - 1) : min(n0 + 16 * (rc - BM / 16) + lrow, N - 1);
    const bf16_t* base = isa ? p.A3 + (size_t)pl * p.planeA : p.B3 + (size_t)pl * p.planeB;
    sbase[j] = reinterpret_cast<const char*>(base + kbeg);
    voff[j] = ((unsigned)row * (unsigned)(isa ? p.lda : p.ldb) + oc * 8) * 2u;
    ldst[j] = lds_base + pl * PLANE + rc * 1024;
  }
  auto issue = [&](auto j_c, int kt, int buf) __attribute__((always_inline)) {
    constexpr int j = decltype(j_c)::value;
    glds16(sbase[j] + (size_t)kt * (BKT * 2), voff[j], ldst[j] + buf * BUF);
  };

  f32x16 hi[TI][TJ], lo[TI][TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) hi[0][j][e] = 0.f, lo[0][j][e] = 0.f;
  bf16x8 fa[2][TI][3], fb[2][TJ][3];
  // fragment of (row, 16-deep step s16, lane half h): octet 2 s16 + h, stored in slot octet ^ ((row >> 2) & 3)
  const int sw = (h ^ ((r >> 2) & 3)) * 16;
  const int fa_off = (wm * TI * 32 + r) * ROWB + sw, fb_off = (BM + wn * TJ * 32 + r) * ROWB + sw;
  constexpr int NF = 3 * (TI + TJ);
  auto frag_one = [&](auto f_c, auto st_c, int buf, int s16) __attribute__((always_inline)) {
    constexpr int f = decltype(f_c)::value, st = decltype(st_c)::value;
    constexpr int FPL[6] = {2, 0, 0, 2, 1, 1};
    constexpr bool isa = f < 6 ? (f % 2 == 0) : (f - 6 < 3 * (TI - 1));
    constexpr int tl = f < 6 ? 0 : (isa ? 1 + (f - 6) / 3 : 1 + (f - 6 - 3 * (TI - 1)) / 3);
    constexpr int pl = f < 6 ? FPL[f] : (isa ? (f - 6) % 3 : (f - 6 - 3 * (TI - 1)) % 3);
    const char* base = lds3 + buf * BUF + pl * PLANE;
    if constexpr (isa) fa[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + ((fa_off + tl * 32 * ROWB) ^ (s16 * 32)));
    else fb[st][tl][pl] = *reinterpret_cast<const bf16x8*>(base + ((fb_off + tl * 32 * ROWB) ^ (s16 * 32)));
  };
  constexpr int S = 12 * G, NR = (NF + 2) / 3, SB = S - NR;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

#ifdef P3_STAMPS
  unsigned long long st0 = 0, sr0 = 0;
#endif
  if constexpr (SCHED == 0) {
  if (KT > 0) {
#pragma unroll
    for (int t = 0; t < NBUF; ++t)
      if (t < KT) static_for<CW>([&](auto j_c) { issue(j_c, t, t); });
    // tile 0 has landed when at most the younger tiles' DMAs are outstanding
    if (KT >= NBUF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * CW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    static_for<NF>([&](auto f_c) { frag_one(f_c, C0{}, 0, 0); });
  }
#ifdef P3_STAMPS
  st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
    static_for<S>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int step = s / (6 * G), g = (s % (6 * G)) / 6, q = s % 6;
      mfma_one<DUAL, q>(fa[step][g / TJ], fb[step][g % TJ], hi[g / TJ][g % TJ], lo[g / TJ][g % TJ]);
      static_for<NF>([&](auto f_c) {
        constexpr int f = decltype(f_c)::value;
        if constexpr (f * (6 * G) / NF == s) frag_one(f_c, C1{}, cur, 1);
      });
      if constexpr (s == SB - 1) {
        // tile kt + 1 must have landed: the tiles issued behind it (kt + 2 .. kt + NBUF - 1, where they exist) may fly on
        if (NBUF >= 3 && kt + NBUF - 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * CW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (s >= SB) {
        if (kt + NBUF < KT) {
          static_for<CW>([&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
            if constexpr (j * NR / CW == s - SB) issue(j_c, kt + NBUF, cur);
          });
        }
        static_for<NF>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          if constexpr (f / 3 == s - SB) frag_one(f_c, C0{}, nxt, 0);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    cur = nxt;
  }
  } else {
  // SCHED 1 (NBUF = 3): the barrier sits in the MIDDLE of a tile (behind step 0: tile kt's last fragment reads are the
  // step-1 reads issued beside step 0), the next tile's step-0 fragment reads are spread over step 1, and the LDS-DMAs
  // of tile kt + 2 -- into the buffer tile kt - 1 left at the previous barrier -- over the whole tile: no burst of six
  // DMA issues and nine reads behind the barrier with both waves of a SIMD in the same phase.
  static_assert(SCHED == 0 || NBUF == 3, "SCHED 1 needs three buffers");
  constexpr int HS = 6 * G, SBAR = HS - 1;
  constexpr int FS = HS - 2 > 0 ? HS - 2 : 1;                 // the next tile's reads end two slots before the tile does
  constexpr int NI = (SBAR * CW) / S + 1;                     // DMAs of tile kt + 2 issued up to slot SBAR: j S / CW <= SBAR
  if (KT > 0) {
    static_for<CW>([&](auto j_c) { issue(j_c, 0, 0); });
    if (KT > 1) {
      static_for<CW>([&](auto j_c) { issue(j_c, 1, 1); });
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    static_for<NF>([&](auto f_c) { frag_one(f_c, C0{}, 0, 0); });
  }
#ifdef P3_STAMPS
  st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const int nxt = cur + 1 == 3 ? 0 : cur + 1, fill = nxt + 1 == 3 ? 0 : nxt + 1;
    const bool more = kt + 2 < KT;
    static_for<S>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      constexpr int step = s / HS, g = (s % HS) / 6, q = s % 6;
      mfma_one<DUAL, q>(fa[step][g / TJ], fb[step][g % TJ], hi[g / TJ][g % TJ], lo[g / TJ][g % TJ]);
      if constexpr (s < HS && !(ABL & 2)) {
        static_for<NF>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          if constexpr (f * HS / NF == s) frag_one(f_c, C1{}, cur, 1);
        });
      }
      if (more && !(ABL & 4)) {
        static_for<CW>([&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          if constexpr ((j * S) / CW == s) issue(j_c, kt + 2, fill);
        });
      }
      if constexpr (s == SBAR) {
        if constexpr (!(ABL & 8)) {
          if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!(ABL & 1)) __builtin_amdgcn_s_barrier();
      }
      if constexpr (s >= HS && !(ABL & 2)) {
        static_for<NF>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          if constexpr (f * FS / NF == s - HS) frag_one(f_c, C0{}, nxt, 0);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    cur = nxt;
  }
  }
#ifdef P3_STAMPS
  if (p.stamps && tid == 0) {
    unsigned long long* d = p.stamps + 4 * (blockIdx.x + gridDim.x * blockIdx.y);
    d[0] = st0, d[1] = sr0, d[2] = __builtin_amdgcn_s_memtime(), d[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // ---- epilogue (gemm3_kernel's).  C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float* Cs = p.C + (size_t)piece * p.slab;
  auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const unsigned ldc = (unsigned)p.ldc;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = n0 + (wn * TJ + j) * 32 + r;
      const bool colok = FULL || col < N;
      const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
      const int rbase = m0 + wm * 32 + 4 * h;
      const size_t off = (size_t)rbase * ldc + col;
      float zv[16];
      if (EPI == rows::EPI_MUL_GELUGRAD || EPI == rows::EPI_MUL_POS) {
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
        float v = (hi[0][j][e] + lo[0][j][e]) + bv;
        if (EPI == rows::EPI_BIAS_RELU) v = v > 0.f ? v : 0.f;
        if (EPI == rows::EPI_BIAS_GELU2) {
          const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
          const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
          p.Z[off + (unsigned)lr * ldc] = cdf + v * pdf;
          v = v * cdf;
        }
        if (EPI == rows::EPI_MUL_GELUGRAD) v *= zv[e];
        if (EPI == rows::EPI_MUL_POS) v = zv[e] > 0.f ? v : 0.f;
        Cs[off + (unsigned)lr * ldc] = v;
      }
    }
  };
  if (m0 + BM <= M && n0 + BN <= N) epilogue(std::true_type{});
  else epilogue(std::false_type{});
}

// x -> P3(x): fp32 [R][C] (leading dimension ld) -> bf16 planes [3][R][ldo]; eight elements per thread (C % 8 == 0)
__global__ __launch_bounds__(256) void split3_kernel(const float* x, long long R, int C, int ld, bf16_t* out, long long plane,
                                                      int ldo) {
  const int c8 = C >> 3;
  const long long n = R * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c8;
    const int oc = (int)(i - row * c8);
    const float4 a = *reinterpret_cast<const float4*>(x + row * ld + oc * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + row * ld + oc * 8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    rows3::u32x4 pk[3];
    static_for<8>([&](auto c_c) { rows3::split_chunk<decltype(c_c)::value>(v, pk); });
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<rows3::u32x4*>(out + pl * plane + row * ldo + oc * 8) = pk[pl];
  }
}

}  // namespace rows3p
}  // namespace pdae
