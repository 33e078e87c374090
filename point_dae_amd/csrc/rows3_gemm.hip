// rows3_gemm.hip -- the exact-split bf16 ("bf16x3, six products") instantiations of the row-GEMM family: the same C
// entries (pdae_rows_gemm, pdae_rows_wgrad_multi, pdae_rows_wgrad_listed; rows_gemm.hip) dispatch here when the
// library's GEMM arithmetic is PDAE_GEMM_BF16X3 (the default; pdae_set_gemm_arith / PDAE_GEMM=f32mfma select the
// fp32-input MFMA kernels).  Kernels: rows3_kernel.h.
#include "rows3_kernel.h"

namespace pdae {
namespace rows3 {

// tile shapes of the GEMM family: {TI, TJ, WM, WN, KS}; 32-deep LDS tiles, one block per CU
//   0: 128 x 128, 8 waves of 32 x 64
//   1:  64 x 128, 4 waves of 32 x 64
//   2: 128 x 192, 8 waves of 32 x 96
//   3: 128 x  64, 8 waves of 32 x 32
const Cfg3 kCfg3[NCFG3] = {{1, 2, 4, 2, 2}, {1, 2, 2, 2, 2}, {1, 3, 4, 2, 2}, {1, 1, 4, 2, 2}};

size_t lds_bytes3(const Cfg3& c) {
  const int bm = 32 * c.ti * c.wm, bn = 32 * c.tj * c.wn;
  return (size_t)2 * 3 * (bm + bn) * (c.ks == 2 ? 80 : 48);
}

template <int TI, int TJ, int WM, int WN, int KS, bool BKN, int EPI>
static void launch_cfg3(Args& a, int splits, int stream_blocks, hipStream_t s) {
  constexpr int BM = 32 * TI * WM, BN = 32 * TJ * WN;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
  a.kchunk = ((a.K + splits - 1) / splits + BK3 - 1) / BK3 * BK3;
  const Cfg3 c = {TI, TJ, WM, WN, KS};
  const size_t lds = lds_bytes3(c);
  auto k = gemm3_kernel<TI, TJ, WM, WN, KS, BKN, EPI, true>;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  const int chunk = (a.tiles + 7) / 8;
  a.stream_blocks = stream_blocks, a.slabs = splits;
  const dim3 grid = stream_blocks ? dim3(stream_blocks, 1, 1) : dim3(8 * chunk, splits, 1);
  hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), lds, s, a);
}

template <bool BKN, int EPI>
static void launch_rows3(Args& a, int cfg, int splits, int sb, hipStream_t s) {
  switch (cfg) {
    case 0: launch_cfg3<1, 2, 4, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 1: launch_cfg3<1, 2, 2, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    case 2: launch_cfg3<1, 3, 4, 2, 2, BKN, EPI>(a, splits, sb, s); break;
    default: launch_cfg3<1, 1, 4, 2, 2, BKN, EPI>(a, splits, sb, s); break;
  }
}

void launch_gemm3(Args& a, int cfg, bool w_kn, int epi, int splits, int stream_blocks, hipStream_t s) {
  if (!w_kn) {
    if (epi == rows::EPI_STORE) launch_rows3<false, rows::EPI_STORE>(a, cfg, splits, stream_blocks, s);
    else if (epi == rows::EPI_BIAS_RELU) launch_rows3<false, rows::EPI_BIAS_RELU>(a, cfg, splits, 0, s);
    else if (epi == rows::EPI_MUL_POS) launch_rows3<false, rows::EPI_MUL_POS>(a, cfg, splits, 0, s);
    else launch_rows3<false, rows::EPI_BIAS_GELU2>(a, cfg, splits, 0, s);
  } else {
    if (epi == rows::EPI_STORE) launch_rows3<true, rows::EPI_STORE>(a, cfg, splits, stream_blocks, s);
    else if (epi == rows::EPI_MUL_POS) launch_rows3<true, rows::EPI_MUL_POS>(a, cfg, splits, 0, s);
    else launch_rows3<true, rows::EPI_MUL_GELUGRAD>(a, cfg, splits, 0, s);
  }
}

template <bool FORMS>
static void wgrad3_launch(const rows::WgradArgs& g, int pl, hipStream_t s) {
  using namespace rows;
  constexpr int TN = 128;
  constexpr int PARTS1 = WTM * TN / 4 / (256 * wru(1));
  constexpr int PARTSL = WTM * TN / 4 / 256;
  const size_t lds = (size_t)2 * 3 * (WTM + TN) * 80;
  auto k = wgrad3b_kernel<FORMS>;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipLaunchKernelGGL(k, dim3(g.blocks), dim3(512), lds, s, g);
  if (pl == 1) hipLaunchKernelGGL((wgrad_reduce_kernel<1, TN>), dim3(g.tiles * PARTS1), dim3(256), 0, s, g);
  else if (pl == 4) hipLaunchKernelGGL((wgrad_reduce_kernel<4, TN>), dim3(g.tiles * PARTSL * 4), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<8, TN>), dim3(g.tiles * PARTSL * 8), dim3(256), 0, s, g);
}

void launch_wgrad3(const rows::WgradArgs& g, int tn, int pl, hipStream_t s) {
  (void)tn;                                             // (one tile width: 128)
  if (g.a_groups || g.b_groups || g.scale) wgrad3_launch<true>(g, pl, s);
  else wgrad3_launch<false>(g, pl, s);
}

template <int PRO, int EPI>
static void conv3_launch(NtArgs& a, hipStream_t s) {
  a.tile_rows = (a.M + 127) / 128;
  a.tiles_n = (a.N + 127) / 128;
  a.tiles = a.tile_rows * a.tiles_n;
  const size_t lds = (size_t)2 * 3 * 256 * 80 + 4 * 2 * 128 * sizeof(float);
  auto k = conv3_kernel<PRO, EPI>;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  // persistent blocks (one residency: 32 CUs per XCD) when the stream of k-tiles keeps its parity across output tiles
  const int chunk = (a.tiles + 7) / 8, kt = a.K / 32;
  const int nslots = (kt % 2 == 0 && kt >= 4 && chunk > 32) ? 32 : chunk;
  hipLaunchKernelGGL(k, dim3(8 * nslots), dim3(512), lds, s, a);
}

bool conv3_takes(const NtArgs& a, int pro, int epi) {
  (void)pro;
  if (a.K % 32 != 0 || a.K < 32 || a.M <= 0) return false;
  const bool group_epi = epi == EPI_GROUPBIAS_STATS || epi == EPI_GROUPMAX || epi == EPI_STORE_GROUPMAX || epi == EPI_GROUP_SCATTER;
  if ((group_epi || a.a_groups) && a.M % 32 != 0) return false;
  return epi == EPI_BIAS || epi == EPI_STATS || group_epi;
}

void launch_conv3(NtArgs& a, int pro, int epi, hipStream_t s) {
  if (pro == PRO_NONE) {
    switch (epi) {
      case EPI_BIAS: conv3_launch<PRO_NONE, EPI_BIAS>(a, s); break;
      case EPI_GROUPBIAS_STATS: conv3_launch<PRO_NONE, EPI_GROUPBIAS_STATS>(a, s); break;
      case EPI_STATS: conv3_launch<PRO_NONE, EPI_STATS>(a, s); break;
      case EPI_STORE_GROUPMAX: conv3_launch<PRO_NONE, EPI_STORE_GROUPMAX>(a, s); break;
      case EPI_GROUPMAX: conv3_launch<PRO_NONE, EPI_GROUPMAX>(a, s); break;
      default: conv3_launch<PRO_NONE, EPI_GROUP_SCATTER>(a, s); break;
    }
  } else {
    switch (epi) {
      case EPI_BIAS: conv3_launch<PRO_BNRELU, EPI_BIAS>(a, s); break;
      case EPI_STATS: conv3_launch<PRO_BNRELU, EPI_STATS>(a, s); break;
      case EPI_STORE_GROUPMAX: conv3_launch<PRO_BNRELU, EPI_STORE_GROUPMAX>(a, s); break;
      case EPI_GROUPMAX: conv3_launch<PRO_BNRELU, EPI_GROUPMAX>(a, s); break;
      case EPI_GROUPBIAS_STATS: conv3_launch<PRO_BNRELU, EPI_GROUPBIAS_STATS>(a, s); break;
      default: conv3_launch<PRO_BNRELU, EPI_GROUP_SCATTER>(a, s); break;
    }
  }
}

}  // namespace rows3
}  // namespace pdae
