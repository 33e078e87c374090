// LAB: the plane-operand exact-split row GEMM (point_dae_amd/csrc/rows3p_kernel.h) behind C entries for
// tools/lab/p3_lab.py (bit-equality with pdae_rows_gemm, time per launch).  Not part of the library.
#include "rows3p_kernel.h"

using namespace pdae;
using namespace pdae::rows3p;

extern "C" int lab_split3(const float* x, long long R, int C, unsigned short* out, void* stream) {
  const long long n = R * (C / 8);
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, R, C, C, out, R * (long long)C, C);
  return (int)hipGetLastError();
}

template <int TJ, int NBUF, int SCHED = 0, int ABL = 0>
static int launch(PArgs a, int splits, hipStream_t s) {
  constexpr int BM = 128, BN = 64 * TJ;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.tiles = ((a.M + BM - 1) / BM) * a.tiles_n;
  a.kchunk = ((a.K / splits + 31) / 32) * 32;
  const size_t lds = (size_t)NBUF * 3 * (BM + BN) * 64;
  auto k = gemm3p_kernel<TJ, rows::EPI_STORE, NBUF, SCHED, ABL>;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int chunk = (a.tiles + 7) / 8;
  hipLaunchKernelGGL(k, dim3(8 * chunk, splits, 1), dim3(512), lds, s, a);
  return (int)hipGetLastError();
}

extern "C" int lab_gemm3p(int variant, int M, int N, int K, const unsigned short* A3, const unsigned short* B3, float* C,
                          int splits, void* stream, unsigned long long* stamps) {
  PArgs a = {};
  a.M = M, a.N = N, a.K = K, a.A3 = A3, a.planeA = (long long)M * K, a.lda = K, a.B3 = B3, a.planeB = (long long)N * K, a.ldb = K;
  a.C = C, a.ldc = N, a.slab = (long long)M * N;
#ifdef P3_STAMPS
  a.stamps = stamps;
#endif
  hipStream_t s = (hipStream_t)stream;
  if (variant == 0) return launch<2, 3>(a, splits, s);
  if (variant == 1) return launch<2, 2>(a, splits, s);
  if (variant == 2) return launch<3, 2>(a, splits, s);
  if (variant == 3) return launch<1, 3>(a, splits, s);
  if (variant == 4) return launch<1, 2>(a, splits, s);
  if (variant == 5) return launch<2, 3, 1>(a, splits, s);
  if (variant == 6) return launch<1, 3, 1>(a, splits, s);
  if (variant == 11) return launch<2, 3, 1, 1>(a, splits, s);
  if (variant == 12) return launch<2, 3, 1, 2>(a, splits, s);
  if (variant == 14) return launch<2, 3, 1, 4>(a, splits, s);
  if (variant == 18) return launch<2, 3, 1, 8>(a, splits, s);
  if (variant == 15) return launch<2, 3, 1, 15>(a, splits, s);
  if (variant == 13) return launch<2, 3, 1, 3>(a, splits, s);
  return -1;
}
