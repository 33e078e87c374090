// nt_args.h -- argument struct and producer / epilogue codes of the fused NT GEMMs (gemm.hip: fp32-input MFMA;
// rows3_kernel.h conv3_kernel: the same contracts on exact-split bf16).
#pragma once
#include "common.h"

namespace pdae {

enum { PRO_NONE = 0, PRO_BNRELU = 1 };
enum {
  EPI_BIAS = 0,
  EPI_BIAS_RELU = 1,
  EPI_BIAS_GELU = 2,
  EPI_GROUPBIAS_STATS = 3,
  EPI_GROUPMAX = 4,
  EPI_STORE_GROUPMAX = 5,
  EPI_STATS = 6,           // C = acc (+ bias); per-column sum / sum of squares like EPI_GROUPBIAS_STATS
  EPI_GROUP_SCATTER = 7    // C[c_groups[m/32]*32 + m%32] = acc + gbias[m/32] (gbias nullable): whole 32-row groups
                           // of a compact product land at listed groups of a larger matrix
};

struct NtArgs {
  int M, N, K;
  const float* A;
  int lda;
  const float* B;
  int ldb;
  float* C;
  int ldc;
  const float* bias;       // [N] or null
  const float* pro_scale;  // [K]  PRO_BNRELU
  const float* pro_shift;  // [K]
  const float* gbias;      // [M/32][N]  EPI_GROUPBIAS_STATS
  float* stats;            // [8][2][N]  per-XCD-slot partial sum / sumsq
  float* stats_det;        // deterministic mode: [tile rows][2][N] plain-store partials (else null)
  float* gmax;             // [M/32][N]  EPI_*GROUPMAX
  unsigned char* garg;     // [M/32][N]
  const int* a_groups;     // nullable: row m of A is source row a_groups[m/32]*32 + m%32
  const int* c_groups;     // EPI_GROUP_SCATTER: destination group of tile-row group m/32
  int tiles_n, tiles, tile_rows;
};

}  // namespace pdae
