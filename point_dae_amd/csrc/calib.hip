// calib.hip -- box calibration for bench.py (include/pdae.h "Measurement aids"): two kernels of known work whose rates
// differ from one MI355X box to the next by as much as the step does (+-4 %, up to 12 % on matrix loops: the clock a
// device holds under load, MI355X_MICROARCH.md "DVFS give-back" item 5), so that a round's delta can be read against the
// box it was measured on.  Nothing of the training step calls them.
//   pdae_calib_mfma_bf16   register-only v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands, one wave per SIMD on
//                          every CU; reports the shader clock the loop held (s_memtime / s_memrealtime)
//   pdae_calib_copy        16 B-per-lane streaming copy (the microarchitecture guide's HBM yardstick: 6.29 TB/s)
#include "common.h"

namespace pdae {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaPerIter = 32;

__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, float* sink, long long* clk) {
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  // operands: bf16 values in [0.5, 2) with fresh mantissa bits every iteration (the switching activity of real data:
  // all-zero operands hold a higher clock and would flatter the box)
  unsigned sx = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x, sy = 0x85ebca6bu * (threadIdx.x + 7) + 3u * blockIdx.x;
  for (int i = 0; i < iters; ++i) {
    u32x4 ux, uy;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sx = sx * 1664525u + 1013904223u, sy = sy * 22695477u + 1u;
      ux[q] = (sx & 0x007f007fu) | 0x3f003f00u, uy[q] = (sy & 0x007f007fu) | 0x3f003f00u;
    }
    const bf16x8 x = __builtin_bit_cast(bf16x8, ux), y = __builtin_bit_cast(bf16x8, uy);
    if ((i & 15) == 0) {                       // keep the sums finite
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] *= 1e-30f;
    }
#pragma unroll
    for (int u = 0; u < kMfmaPerIter / 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  if (s == 12345.678f) sink[0] = s;           // (keeps the loop; never true)
  if (threadIdx.x == 0) {                      // shader-clock cycles and 100 MHz ticks wave 0 of this block lived
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace
}  // namespace pdae

using namespace pdae;

extern "C" int pdae_calib_mfma_bf16(int blocks, int iters, float* sink, long long* clk /*[blocks][2]*/, double* flops,
                                    pdae_stream_t stream) {
  if (blocks <= 0 || iters <= 0 || !sink || !clk) return bad_arg("calib_mfma_bf16: blocks, iters > 0 and both buffers");
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), iters, sink, clk);
  if (flops) *flops = (double)blocks * 4.0 * iters * kMfmaPerIter * (32.0 * 32.0 * 16.0 * 2.0);
  return check_launch("calib_mfma_bf16");
}

extern "C" int pdae_calib_copy(long long bytes, const void* src, void* dst, pdae_stream_t stream) {
  if (bytes <= 0 || bytes % 16 || !src || !dst) return bad_arg("calib_copy: bytes > 0, a multiple of 16");
  hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 8), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), bytes / 16);
  return check_launch("calib_copy");
}
