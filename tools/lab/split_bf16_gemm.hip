// LAB ONLY (not part of libpdae_hip.so): C[M,N] = A[M,K] . B[N,K]^T in "bf16x3" arithmetic -- every fp32 operand split
// into three bf16 terms (hi + mid + lo = the fp32 value to 2^-24), six bf16 MFMA products per k-step (hi hi, hi mid,
// mid hi, hi lo, lo hi, mid mid; the dropped terms are below 2^-24 of the product), fp32 accumulation.  Measures what
// the 16x faster bf16 matrix pipe could buy the step's GEMMs at fp32-class accuracy.  Operands arrive PRE-SPLIT as three
// bf16 planes each (a producer epilogue / a once-per-step weight pass would write them).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lab/split_bf16_gemm.hip -o gpurun_out/libsplit_bf16.so
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float x) {       // round to nearest even (finite inputs)
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// x (rows, cols) fp32 -> planes [3][rows][cols] bf16
__global__ void split3_kernel(long long n, const float* __restrict__ x, unsigned short* __restrict__ p) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const unsigned short h = f2bf(v);
  const float r1 = v - bf2f(h);
  const unsigned short m = f2bf(r1);
  const float r2 = r1 - bf2f(m);
  p[i] = h, p[n + i] = m, p[2 * n + i] = f2bf(r2);
}

constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 8;   // LDS rows padded to 40 bf16 (80 B)

// 256 threads = 4 waves (2 x 2), every wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
template <int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(int M, int N, int K, const unsigned short* __restrict__ A3,
                                                          const unsigned short* __restrict__ B3, float* __restrict__ C) {
  __shared__ unsigned short As[3][BM][LDK], Bs[3][BN][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int tiles_n = (N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const size_t pa = (size_t)M * K, pb = (size_t)N * K;
  f32x16 acc[2][2], cor[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f, cor[i][j][e] = 0.f;
  // staging: a k-tile of one plane is 128 rows x 32 bf16 = 128 x 64 B: 4 x 16-B pieces per row, 512 pieces, 2 per thread
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int piece = tid + q * 256, row = piece >> 2, c8 = (piece & 3) * 8;
        const int ga = min(m0 + row, M - 1), gb = min(n0 + row, N - 1);
        *reinterpret_cast<uint4*>(&As[pl][row][c8]) = *reinterpret_cast<const uint4*>(A3 + pl * pa + (size_t)ga * K + k0 + c8);
        *reinterpret_cast<uint4*>(&Bs[pl][row][c8]) = *reinterpret_cast<const uint4*>(B3 + pl * pb + (size_t)gb * K + k0 + c8);
      }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[pl][i] = *reinterpret_cast<const bf16x8*>(&As[pl][wm * 64 + i * 32 + r][s * 16 + 8 * h]);
          b[pl][i] = *reinterpret_cast<const bf16x8*>(&Bs[pl][wn * 64 + i * 32 + r][s * 16 + 8 * h]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms into their own accumulator, the leading term into acc
          if (NPROD >= 6) {
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], cor[i][j], 0, 0, 0);   // hi lo
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], cor[i][j], 0, 0, 0);   // lo hi
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], cor[i][j], 0, 0, 0);   // mid mid
          }
          if (NPROD >= 3) {
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], cor[i][j], 0, 0, 0);   // hi mid
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], cor[i][j], 0, 0, 0);   // mid hi
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);     // hi hi
        }
    }
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * 64 + j * 32 + r;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][e] + cor[i][j][e];
      }
}

extern "C" int lab_split3(long long n, const float* x, void* planes, void* stream) {
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, x,
                     (unsigned short*)planes);
  return (int)hipGetLastError();
}
extern "C" int lab_gemm_bf16x3(int M, int N, int K, const void* A3, const void* B3, float* C, int nprod, void* stream) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (nprod >= 6) hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, M, N, K, (const unsigned short*)A3, (const unsigned short*)B3, C);
  else if (nprod >= 3) hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, M, N, K, (const unsigned short*)A3, (const unsigned short*)B3, C);
  else hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, M, N, K, (const unsigned short*)A3, (const unsigned short*)B3, C);
  return (int)hipGetLastError();
}
