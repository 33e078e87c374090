// LAB: what does the fp32-input matrix pipe SUSTAIN?  Register-only v_mfma_f32_32x32x2_f32 chains (no memory traffic), four
// independent accumulators per wave, `waves` waves per SIMD, for launches of ~50 us to ~20 ms.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out, long long* clk) {
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 a0, a1, a2, a3;
  for (int e = 0; e < 16; ++e) a0[e] = a1[e] = a2[e] = a3[e] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#ifdef RANDOM_OPERANDS
    // operands with fresh mantissa bits every iteration (switching activity of real data), values in [0.5, 2)
    unsigned ux = __float_as_uint(x) * 1664525u + 1013904223u + threadIdx.x, uy = __float_as_uint(y) * 22695477u + 1u + blockIdx.x;
    x = __uint_as_float((ux & 0x00ffffffu) | 0x3f000000u), y = __uint_as_float((uy & 0x00ffffffu) | 0x3f000000u);
    if ((i & 63) == 0) { for (int e = 0; e < 16; ++e) a0[e] *= 1e-30f, a1[e] *= 1e-30f, a2[e] *= 1e-30f, a3[e] *= 1e-30f; }
#endif
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {      // shader-clock cycles and 100 MHz ticks this wave lived
    clk[0] = __builtin_amdgcn_s_memtime() - c0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  long long* clk;
  hipMalloc(&clk, 16);
  long long hclk[2];
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int blocks_per_cu : {1, 2}) {
    for (int iters : {50, 1000, 20000, 100000}) {
      const int grid = 256 * blocks_per_cu;
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, iters, out, clk);
      hipDeviceSynchronize();
      float best = 1e30f, sum = 0.f;
      const int reps = iters >= 20000 ? 5 : 20;
      for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, iters, out, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best, sum += ms;
      }
      const double flop = (double)grid * 4 /*waves*/ * iters * 32 /*mfma per iter*/ * (32.0 * 32 * 2 * 2);
      hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
      printf("%d block(s)/CU x 4 waves, %6d iters: best %8.3f ms = %6.1f TFLOP/s, mean %8.3f ms = %6.1f TFLOP/s, shader clock %.2f GHz\n",
             blocks_per_cu, iters, best, flop / best / 1e9, sum / reps, flop / (sum / reps) / 1e9, hclk[0] / (hclk[1] * 10.0));
    }
  }
  return 0;
}
