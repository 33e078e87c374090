// rows3_conv.hip -- the fused NT GEMMs of the embedder / set-abstraction MLPs on exact-split bf16 (rows3_kernel.h
// conv3_kernel), dispatched to from gemm.hip
#include "rows3_kernel.h"

namespace pdae {
namespace rows3 {

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
