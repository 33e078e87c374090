// Scratch harness #2 for NT GEMM variants (not part of the product): per-wave tile size sweep.
// hipcc --offload-arch=gfx950 -O3 -DWTM=128 -DWTN=64 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef BM
#define BM 256
#endif
#ifndef BN
#define BN 256
#endif
#ifndef WTM
#define WTM 64
#endif
#ifndef WTN
#define WTN 64
#endif
#ifndef WPE
#define WPE 0
#endif
#ifndef SGB
#define SGB 0
#endif
#ifndef NOSTORE
#define NOSTORE 0
#endif
constexpr int GBK = 32, GLD = 36;
constexpr int WM = BM / WTM, WN = BN / WTN, NT = WM * WN * 64;
constexpr int TI = WTM / 32, TJ = WTN / 32;
constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;  // float4 per thread per slab
__device__ __forceinline__ int xcd_tile(int tiles) {
  const int chunk = (tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t = xcd * chunk + slot;
  return (slot < chunk && t < tiles) ? t : -1;
}
#if WPE
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
#else
__global__ __launch_bounds__(NT)
#endif
void k(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int tiles_n, int tiles) {
  extern __shared__ float lds[];
  const int tile = xcd_tile(tiles);
  if (tile < 0) return;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  constexpr int RS = NT / 8;
  float4 ra[LA], rb[LB];
  auto gload = [&](int kt) {
    const int kk = kt * GBK + scol;
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + srow + RS * i) * K + kk);
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + srow + RS * i) * K + kk);
  };
  auto lstore = [&](int buf) {
    float* As = lds + buf * (BM + BN) * GLD;
    float* Bs = As + BM * GLD;
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<float4*>(As + (srow + RS * i) * GLD + scol) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i) *reinterpret_cast<float4*>(Bs + (srow + RS * i) * GLD + scol) = rb[i];
  };
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int KT = K / GBK;
  gload(0); lstore(0); __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) gload(kt + 1);
    const float* As = lds + buf * (BM + BN) * GLD + (wm * WTM + r) * GLD + 4 * h;
    const float* Bs = lds + buf * (BM + BN) * GLD + BM * GLD + (wn * WTN + r) * GLD + 4 * h;
    float4 a[2][TI], b[2][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) a[0][i] = *reinterpret_cast<const float4*>(As + i * 32 * GLD);
#pragma unroll
    for (int j = 0; j < TJ; ++j) b[0][j] = *reinterpret_cast<const float4*>(Bs + j * 32 * GLD);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < 4) {
#pragma unroll
        for (int i = 0; i < TI; ++i) a[nxt][i] = *reinterpret_cast<const float4*>(As + i * 32 * GLD + (s + 1) * 8);
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[nxt][j] = *reinterpret_cast<const float4*>(Bs + j * 32 * GLD + (s + 1) * 8);
      }
#if !SGB
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].x, b[cur][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].y, b[cur][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].z, b[cur][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i].w, b[cur][j].w, acc[i][j], 0, 0, 0);
        }
#if SGB
      // interleave: one LDS fragment read (of the next k-group) behind every few MFMAs; this
      // slab's share of the global loads up front
      {
        constexpr int NM = TI * TJ * 4, NR = TI + TJ, PER = NM / NR;
        __builtin_amdgcn_sched_group_barrier(0x020, (LA + LB + 3) / 4, 0);
#pragma unroll
        for (int g = 0; g < NR; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR, 0);
      }
#endif
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn * WTN + j * 32 + r;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (!NOSTORE || acc[i][j][e] == 12345.678f) C[(size_t)row * N + col] = acc[i][j][e];
      }
  }
}
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 256;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  for (auto& v : h) { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), w = rand() / (float)RAND_MAX; v = sqrtf(-2.f * logf(u)) * cosf(6.2831853f * w); }
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  const int tm = M / BM, tn = N / BN, tiles = tm * tn, grid = 8 * ((tiles + 7) / 8);
  const size_t lds = 2 * (BM + BN) * GLD * 4;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, NT, lds);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, 0, M, N, K, A, B, C, tn, tiles);
  hipEventRecord(s);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, 0, M, N, K, A, B, C, tn, tiles);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= it;
  // spot check
  std::vector<float> c(N); hipMemcpy(c.data(), C + (size_t)777 * N, N * 4, hipMemcpyDeviceToHost);
  double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)h[(size_t)777 * K + kk] * h[(size_t)5 * K + kk];
  printf("BM%d BN%d WT %dx%d NT%d wpe%d regs %d spill %zu occ/CU %d : %.1f us  %.1f TFLOP/s  chk %.2e (%s)\n", BM, BN, WTM, WTN, NT, WPE, fa.numRegs, (size_t)fa.localSizeBytes, occ, ms * 1e3,
         2.0 * M * N * K / ms / 1e9, fabs(c[5] - ref), hipGetErrorString(hipGetLastError()));
  return 0;
}
