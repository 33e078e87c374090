// rows3_gemm.hip -- the exact-split bf16 ("bf16x3, six products") instantiations of the row-GEMM family: the same C
// entries (pdae_rows_gemm, pdae_rows_wgrad_multi, pdae_rows_wgrad_listed; rows_gemm.hip) dispatch here when the
// library's GEMM arithmetic is PDAE_GEMM_BF16X3 (the default; pdae_set_gemm_arith / PDAE_GEMM=f32mfma select the
// fp32-input MFMA kernels).  Kernels: rows3_kernel.h.
#include "rows_common.h"

namespace pdae {
namespace rows3 {
using rows::Args;

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

void launch_rows3_cfg0(Args& a, bool w_kn, int epi, int splits, hipStream_t s);
void launch_rows3_cfg1(Args& a, bool w_kn, int epi, int splits, hipStream_t s);
void launch_rows3_cfg2(Args& a, bool w_kn, int epi, int splits, hipStream_t s);
void launch_rows3_cfg3(Args& a, bool w_kn, int epi, int splits, hipStream_t s);

void launch_gemm3(Args& a, int cfg, bool w_kn, int epi, int splits, int stream_blocks, hipStream_t s) {
  (void)stream_blocks;
  switch (cfg) {
    case 0: launch_rows3_cfg0(a, w_kn, epi, splits, s); break;
    case 1: launch_rows3_cfg1(a, w_kn, epi, splits, s); break;
    case 2: launch_rows3_cfg2(a, w_kn, epi, splits, s); break;
    default: launch_rows3_cfg3(a, w_kn, epi, splits, s); break;
  }
}

}  // namespace rows3
}  // namespace pdae
