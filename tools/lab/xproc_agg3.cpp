// LAB: loop one variant of librows3_lab.so's gemm3_kernel from a bare HIP process (no torch): is the aggressor's
// effect on another process a property of the kernel or of the process that launches it?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
extern "C" int lab_gemm3(int variant, int M, int N, int K, const float* A, const float* B, int bkn, float* C, void* stream);
int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 7.0;
  const int M = 2944, N = 1152, K = 384;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)N * K * 4);
  hipStream_t s; hipStreamCreate(&s);
  const auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int i = 0; i < 50; ++i) if (lab_gemm3(v, M, N, K, A, B, 0, C, s)) { printf("launch failed\n"); return 2; }
    hipStreamSynchronize(s); n += 50;
  }
  printf("agg3 (bare process): variant %d, %ld launches\n", v, n);
  return 0;
}
