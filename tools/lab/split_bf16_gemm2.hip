// LAB ONLY: C[M,N] = A[M,K] . B[N,K]^T, fp32 operands split INSIDE the kernel into three bf16 planes while they are
// staged into LDS (hi + mid + lo = the fp32 value exactly), six bf16 MFMA products per k-step, fp32 accumulation.
// 128 x 128 tiles, 4 waves (2 x 2) of 64 x 64, BK = 32, two register sets of global loads in flight.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lab/split_bf16_gemm2.hip -o gpurun_out/libsplit_bf16_2.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWB = 80;                        // bytes per LDS row: 32 bf16 + 16 B pad (5 x 16 B: conflict-free b128 reads)
constexpr int PLANE = (BM + BN) * ROWB;         // one plane of one buffer
constexpr int BUF = 3 * PLANE;                  // 61440 B

// eight consecutive floats -> three 16-byte bf16 octets
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
__global__ __launch_bounds__(256) void gemm_split_kernel(int M, int N, int K, const float* __restrict__ A,
                                                         const float* __restrict__ B, float* __restrict__ C) {
  extern __shared__ char lds[];                 // [2][3][256 rows][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile = (int)(blockIdx.x & 7) * ((gridDim.x + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (tile >= tiles_n * ((M + BM - 1) / BM)) return;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  // staging: thread -> (row = tid / 4 + 64 i, octet = tid % 4), i = 0, 1 for A and for B
  const int srow = tid >> 2, soct = tid & 3;
  const float* ap[2];
  const float* bp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ap[i] = A + (size_t)min(m0 + srow + 64 * i, M - 1) * K + soct * 8;
    bp[i] = B + (size_t)min(n0 + srow + 64 * i, N - 1) * K + soct * 8;
  }
  const int KT = K / BK;
  float4 ra[2][4], rb[2][4];                    // [register set][2 rows x 2 float4]
  auto gload = [&](int set, int kt) __attribute__((always_inline)) {
    const int k = min(kt, KT - 1) * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[set][2 * i] = *reinterpret_cast<const float4*>(ap[i] + k);
      ra[set][2 * i + 1] = *reinterpret_cast<const float4*>(ap[i] + k + 4);
      rb[set][2 * i] = *reinterpret_cast<const float4*>(bp[i] + k);
      rb[set][2 * i + 1] = *reinterpret_cast<const float4*>(bp[i] + k + 4);
    }
  };
  auto lstore = [&](int set, int buf) __attribute__((always_inline)) {
    char* base = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 ph, pm, pl;
      split8(ra[set][2 * i], ra[set][2 * i + 1], ph, pm, pl);
      char* d = base + (srow + 64 * i) * ROWB + soct * 16;
      *reinterpret_cast<u32x4*>(d) = ph;
      *reinterpret_cast<u32x4*>(d + PLANE) = pm;
      *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pl;
      split8(rb[set][2 * i], rb[set][2 * i + 1], ph, pm, pl);
      d = base + (BM + srow + 64 * i) * ROWB + soct * 16;
      *reinterpret_cast<u32x4*>(d) = ph;
      *reinterpret_cast<u32x4*>(d + PLANE) = pm;
      *reinterpret_cast<u32x4*>(d + 2 * PLANE) = pl;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* base = lds + buf * BUF;
    const char* pa = base + (wm * 64 + r) * ROWB + 16 * h;
    const char* pb = base + (BM + wn * 64 + r) * ROWB + 16 * h;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[pl][i] = *reinterpret_cast<const bf16x8*>(pa + pl * PLANE + i * 32 * ROWB + s * 32);
          b[pl][i] = *reinterpret_cast<const bf16x8*>(pb + pl * PLANE + i * 32 * ROWB + s * 32);
        }
      // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        if (t < 6 - NPROD) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
      }
    }
  };

  gload(0, 0);
  gload(1, 1);
  lstore(0, 0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < KT; kt += 2) {
    gload(0, kt + 2);
    compute(buf);
    lstore(1, buf ^ 1);
    __syncthreads();
    buf ^= 1;
    if (kt + 1 >= KT) break;
    gload(1, kt + 3);
    compute(buf);
    lstore(0, buf ^ 1);
    __syncthreads();
    buf ^= 1;
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

extern "C" int lab_gemm_split(int M, int N, int K, const float* A, const float* B, float* C, int nprod, void* stream) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int grid = ((tiles + 7) / 8) * 8;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
  }
  if (nprod >= 6) hipLaunchKernelGGL(gemm_split_kernel<6>, dim3(grid), dim3(256), 2 * BUF, (hipStream_t)stream, M, N, K, A, B, C);
  else if (nprod >= 3) hipLaunchKernelGGL(gemm_split_kernel<3>, dim3(grid), dim3(256), 2 * BUF, (hipStream_t)stream, M, N, K, A, B, C);
  else hipLaunchKernelGGL(gemm_split_kernel<1>, dim3(grid), dim3(256), 2 * BUF, (hipStream_t)stream, M, N, K, A, B, C);
  return (int)hipGetLastError();
}
