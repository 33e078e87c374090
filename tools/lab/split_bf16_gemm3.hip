// LAB ONLY: split_bf16_gemm2 with the issue order written out (global loads, the bf16 split + LDS stores and the fragment
// reads interleaved behind groups of four MFMAs, one barrier per k-tile inside the last slab).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;
constexpr int PLANE = (BM + BN) * ROWB;
constexpr int BUF = 3 * PLANE;

__device__ __forceinline__ void split8(const float4& lo4, const float4& hi4, u32x4& ph, u32x4& pm, u32x4& pl) {
  const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x2 x = {v[2 * q], v[2 * q + 1]};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    x = x - hf;
    const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    x = x - mf;
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
    ph[q] = h, pm[q] = m, pl[q] = l;
  }
}

template <int NPROD>
__global__ __launch_bounds__(256) void gemm_split3_kernel(int M, int N, int K, const float* __restrict__ A,
                                                          const float* __restrict__ B, float* __restrict__ C) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile = (int)(blockIdx.x & 7) * ((gridDim.x + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (tile >= tiles_n * ((M + BM - 1) / BM)) return;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int srow = tid >> 2, soct = tid & 3;
  // octet q of a thread: q = 0, 1 -> A rows srow, srow + 64; q = 2, 3 -> B rows
  const float* gp[4];
  gp[0] = A + (size_t)min(m0 + srow, M - 1) * K + soct * 8;
  gp[1] = A + (size_t)min(m0 + srow + 64, M - 1) * K + soct * 8;
  gp[2] = B + (size_t)min(n0 + srow, N - 1) * K + soct * 8;
  gp[3] = B + (size_t)min(n0 + srow + 64, N - 1) * K + soct * 8;
  const int lrow[4] = {srow, srow + 64, BM + srow, BM + srow + 64};
  const int KT = K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 r0[8], r1[8];                          // two register sets: [octet][half]
  auto gload_all = [&](float4 (&rg)[8], int kt) __attribute__((always_inline)) {
    const int k = min(kt, KT - 1) * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rg[2 * q] = *reinterpret_cast<const float4*>(gp[q] + k);
      rg[2 * q + 1] = *reinterpret_cast<const float4*>(gp[q] + k + 4);
    }
  };
  auto lstore_oct = [&](const float4 (&rg)[8], int q, int buf) __attribute__((always_inline)) {
    u32x4 ph, pm, pl;
    split8(rg[2 * q], rg[2 * q + 1], ph, pm, pl);
    char* d = lds + buf * BUF + lrow[q] * ROWB + soct * 16;
    *reinterpret_cast<u32x4*>(d) = ph;
    *reinterpret_cast<u32x4*>(d + PLANE) = pm;
    *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pl;
  };
  bf16x8 fa[2][3][2], fb[2][3][2];              // [stage][plane][tile]
  auto frag = [&](int st, int buf, int s, int q) __attribute__((always_inline)) {   // q = 0..11: plane-major, A tiles then B tiles
    const int pl = q >> 2, w = q & 3;
    const char* base = lds + buf * BUF + pl * PLANE + s * 32 + 16 * h;
    if (w < 2) fa[st][pl][w] = *reinterpret_cast<const bf16x8*>(base + (wm * 64 + w * 32 + r) * ROWB);
    else fb[st][pl][w - 2] = *reinterpret_cast<const bf16x8*>(base + (BM + wn * 64 + (w - 2) * 32 + r) * ROWB);
  };
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
  int buf = 0;
  auto ktile = [&](float4 (&ld)[8], float4 (&st)[8], int kt) __attribute__((always_inline)) {
    const int kl = min(kt + 2, KT - 1) * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t >= 6 - NPROD) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][PA[t]][i], fb[s][PB[t]][j], acc[i][j], 0, 0, 0);
        }
        if (s == 0) {
          // global loads of slab kt + 2 (8 float4: groups 0..3), fragments of step 1 (12: two per group)
          if (t < 4) {
            ld[2 * t] = *reinterpret_cast<const float4*>(gp[t] + kl);
            ld[2 * t + 1] = *reinterpret_cast<const float4*>(gp[t] + kl + 4);
          }
          frag(1, buf, 1, 2 * t);
          frag(1, buf, 1, 2 * t + 1);
        } else {
          // split + store slab kt + 1 (groups 0..3), barrier, fragments of the next k-tile's step 0 (groups 4, 5)
          if (t < 4) lstore_oct(st, t, buf ^ 1);
          if (t == 3) __syncthreads();
          if (t >= 4) {
#pragma unroll
            for (int q = 0; q < 6; ++q) frag(0, buf ^ 1, 0, (t - 4) * 6 + q);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    buf ^= 1;
  };

  gload_all(r0, 0);
  gload_all(r1, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) lstore_oct(r0, q, 0);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 12; ++q) frag(0, 0, 0, q);
  for (int kt = 0; kt < KT; kt += 2) {
    ktile(r0, r1, kt);          // loads slab kt + 2 into r0, stores r1 (slab kt + 1)
    if (kt + 1 < KT) ktile(r1, r0, kt + 1);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * 64 + j * 32 + r;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][e];
      }
}

extern "C" int lab_gemm_split3(int M, int N, int K, const float* A, const float* B, float* C, int nprod, void* stream) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int grid = ((tiles + 7) / 8) * 8;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
  }
  if (nprod >= 6) hipLaunchKernelGGL(gemm_split3_kernel<6>, dim3(grid), dim3(256), 2 * BUF, (hipStream_t)stream, M, N, K, A, B, C);
  else if (nprod >= 3) hipLaunchKernelGGL(gemm_split3_kernel<3>, dim3(grid), dim3(256), 2 * BUF, (hipStream_t)stream, M, N, K, A, B, C);
  else hipLaunchKernelGGL(gemm_split3_kernel<1>, dim3(grid), dim3(256), 2 * BUF, (hipStream_t)stream, M, N, K, A, B, C);
  return (int)hipGetLastError();
}
