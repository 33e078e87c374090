// LAB: variants of the exact-split bf16 row GEMM (point_dae_amd/csrc/rows3_kernel.h) behind one C entry, for
// tools/lab/rows3_lab.py (timing + error against fp64).  Not part of the library.
#include "../../point_dae_amd/csrc/rows3_kernel.h"

using namespace pdae;
using namespace pdae::rows3;

template <int TI, int TJ, int WM, int WN, int KS, bool BKN, bool DUAL, int ABL = 0>
static int launch(rows::Args a, hipStream_t s) {
  constexpr int BM = 32 * TI * WM, BN = 32 * TJ * WN;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
  a.kchunk = a.K;
  const size_t lds = 2 * 3 * (size_t)(BM + BN) * (KS == 2 ? 80 : 48);
  auto k = gemm3_kernel<TI, TJ, WM, WN, KS, BKN, rows::EPI_STORE, DUAL, ABL>;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int chunk = (a.tiles + 7) / 8;
  hipLaunchKernelGGL(k, dim3(8 * chunk, 1, 1), dim3(WM * WN * 64), lds, s, a);
  return (int)hipGetLastError();
}

// (built -DR3_STAMP: tools/lab/gemm3_anatomy.py) the same launch with the stamp buffer in Args::Z: 4 x u64 per block
extern "C" int lab_gemm3_stamped(int variant, int M, int N, int K, const float* A, const float* B, int bkn, float* C,
                                 unsigned long long* stamps, void* stream) {
  rows::Args a = {};
  a.M = M, a.N = N, a.K = K, a.A = A, a.lda = K, a.B = B, a.ldb = bkn ? N : K, a.C = C, a.ldc = N, a.slab = (long long)M * N;
  a.Z = reinterpret_cast<float*>(stamps);
  hipStream_t s = (hipStream_t)stream;
  if (variant == 0) return bkn ? launch<1, 2, 4, 2, 2, true, true>(a, s) : launch<1, 2, 4, 2, 2, false, true>(a, s);
  if (variant == 2) return bkn ? launch<1, 3, 4, 2, 2, true, true>(a, s) : launch<1, 3, 4, 2, 2, false, true>(a, s);
  if (variant == 3) return bkn ? launch<1, 1, 4, 2, 2, true, true>(a, s) : launch<1, 1, 4, 2, 2, false, true>(a, s);
  return -1;
}

extern "C" int lab_gemm3(int variant, int M, int N, int K, const float* A, const float* B, int bkn, float* C, void* stream) {
  rows::Args a = {};
  a.M = M, a.N = N, a.K = K, a.A = A, a.lda = K, a.B = B, a.ldb = bkn ? N : K, a.C = C, a.ldc = N, a.slab = (long long)M * N;
  hipStream_t s = (hipStream_t)stream;
#define V(id, TI, TJ, WM, WN, KS, DUAL)                                  \
  if (variant == id) return bkn ? launch<TI, TJ, WM, WN, KS, true, DUAL>(a, s) : launch<TI, TJ, WM, WN, KS, false, DUAL>(a, s);
  V(0, 1, 2, 4, 2, 2, true)     // 128 x 128, 8 waves of 32 x 64, 32-deep LDS tiles (one block per CU)
  V(1, 1, 2, 4, 2, 2, false)
  V(2, 1, 3, 4, 2, 2, true)     // 128 x 192
  V(3, 1, 1, 4, 2, 2, true)     // 128 x 64
  V(4, 1, 2, 2, 2, 1, true)     // 64 x 128, 4 waves of 32 x 64, 16-deep LDS tiles (two blocks per CU)
  V(5, 2, 1, 2, 2, 1, true)     // 128 x 64, 4 waves of 64 x 32
  V(6, 2, 2, 2, 2, 1, true)     // 128 x 128, 4 waves of 64 x 64
  V(7, 1, 2, 4, 2, 1, true)     // 128 x 128, 8 waves, 16-deep
  V(8, 1, 2, 4, 1, 1, true)     // 128 x 64, 4 waves of 32 x 64
  V(9, 1, 2, 2, 2, 2, true)     // 64 x 128, 4 waves, 32-deep
#define VA(id, ABL) if (variant == id) return launch<1, 2, 4, 2, 2, false, true, ABL>(a, s);
  VA(10, 1) VA(11, 2) VA(12, 4) VA(13, 8) VA(14, 16) VA(17, 31)
  VA(30, 30) VA(29, 29) VA(27, 27) VA(23, 23) VA(15, 15)     // ONE piece of side work left: loads | split | LDS stores | barrier | fragment reads
#define VB(id, ABL) if (variant == id) return launch<1, 2, 2, 2, 1, false, true, ABL>(a, s);
  return -1;
}
