// LAB: the shader clock the chip is holding RIGHT NOW: one wave spins ~10 us and reports s_memtime / s_memrealtime.
// Launched on the stream right behind the kernels of interest (tools/lab/clock_after.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lab/clock_probe.hip -o tools/lab/libclock_probe.so
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(long long* out) {
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  long long r = r0;
  while (r - r0 < 1000) r = __builtin_amdgcn_s_memrealtime();      // 1000 ticks of 100 MHz = 10 us
  out[0] = __builtin_amdgcn_s_memtime() - c0;
  out[1] = r - r0;
}
extern "C" int clock_probe(long long* out, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  return (int)hipGetLastError();
}
