// rows3_wgrad.hip -- the grouped weight gradients on exact-split bf16 (rows3_kernel.h wgrad3b_kernel)
#include "rows3_kernel.h"

namespace pdae {
namespace rows3 {

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
  if (g.direct) return;                              // whole tiles stored by their blocks: nothing to reduce
  if (pl == 1) hipLaunchKernelGGL((wgrad_reduce_kernel<1, TN>), dim3(g.tiles * PARTS1), dim3(256), 0, s, g);
  else if (pl == 4) hipLaunchKernelGGL((wgrad_reduce_kernel<4, TN>), dim3(g.tiles * PARTSL * 4), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<8, TN>), dim3(g.tiles * PARTSL * 8), dim3(256), 0, s, g);
}

void launch_wgrad3(const rows::WgradArgs& g, int tn, int pl, hipStream_t s) {
  (void)tn;                                             // (one tile width: 128)
  if (g.a_groups || g.b_groups || g.scale) wgrad3_launch<true>(g, pl, s);
  else wgrad3_launch<false>(g, pl, s);
}

}  // namespace rows3
}  // namespace pdae
