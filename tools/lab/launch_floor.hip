// Per-kernel cost of a dependent chain of small kernels: plain stream launches from a C++ loop vs a captured hipGraph.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/launch_floor.hip -o gpurun_out/launch_floor && gpurun_out/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
// ~20 us of streaming work: 64 MB read+write
__global__ void medium(float4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}

int main() {
  float* d; size_t big = 16u << 20;               // 16 M floats = 64 MB
  CK(hipMalloc(&d, big * 4)); CK(hipMemset(d, 0, big * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 400;
  for (int mode = 0; mode < 3; ++mode) {          // 0: tiny 1 block, 1: tiny 1024 blocks, 2: medium
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, d, 256);
      else if (mode == 1) hipLaunchKernelGGL(tiny, dim3(1024), dim3(256), 0, s, d, 1024 * 256);
      else hipLaunchKernelGGL(medium, dim3(2048), dim3(256), 0, s, (float4*)d, big / 4);
    };
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(s));
    // (a) plain launches
    float best_a = 1e9, best_host = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::high_resolution_clock::now();
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) launch();
      CK(hipEventRecord(e1, s));
      auto t1 = std::chrono::high_resolution_clock::now();
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float host = std::chrono::duration<float, std::micro>(t1 - t0).count() / N;
      if (ms * 1000 / N < best_a) best_a = ms * 1000 / N;
      if (host < best_host) best_host = host;
    }
    // (b) captured graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best_b = 1e9, best_hostb = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::high_resolution_clock::now();
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      auto t1 = std::chrono::high_resolution_clock::now();
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float host = std::chrono::duration<float, std::micro>(t1 - t0).count() / N;
      if (ms * 1000 / N < best_b) best_b = ms * 1000 / N;
      if (host < best_hostb) best_hostb = host;
    }
    printf("mode %d: stream launches %.2f us/kernel (host %.2f us/launch) | hipGraph replay %.2f us/kernel (host %.2f us/node)\n",
           mode, best_a, best_host, best_b, best_hostb);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
