// LAB: the same pairing inside ONE process: gemm3_kernel (librows3_lab.so, variant argv[1]) looping on stream 1 from a
// second host thread, the victim reduction of tools/xproc_repro.hip looping on stream 2.
#define XPROC_NO_MAIN
#include "../xproc_repro.hip"
#include <atomic>
#include <thread>
extern "C" int lab_gemm3(int variant, int M, int N, int K, const float* A, const float* B, int bkn, float* C, void* stream);
int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 4.0;
  const int M = 2944, N = 1152, K = 384;
  float *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMemset(B, 0, (size_t)N * K * 4));
  hipStream_t s1, s; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s));
  std::atomic<bool> stop{false};
  long nagg = 0;
  std::thread agg([&] {
    while (!stop) { for (int i = 0; i < 50; ++i) lab_gemm3(v, M, N, K, A, B, 0, C, s1); hipStreamSynchronize(s1); nagg += 50; }
  });
  const int R = 65536, C4 = 32, blocks = R / 1024;
  std::vector<float> hd((size_t)R * C4 * 4), hx((size_t)R * 3);
  unsigned st = 12345u;
  auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& t : hd) t = rnd();
  for (auto& t : hx) t = rnd();
  float4 *d, *part; float* x;
  const size_t pbytes = (size_t)blocks * 3 * C4 * sizeof(float4);
  CK(hipMalloc(&d, hd.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&part, pbytes));
  CK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  // reference on the host, the kernel's own summation order is not reproduced: compare iterations with each other
  std::vector<char> ref(pbytes), cur(pbytes);
  long iters = 0, bad = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), sizeof(float4) * 3 * 256, s, R, C4, d, x, part);
    CK(hipMemcpyAsync(cur.data(), part, pbytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if (iters == 0) ref = cur; else bad += memcmp(ref.data(), cur.data(), pbytes) != 0;
    ++iters;
  }
  stop = true; agg.join();
  printf("in one process: gemm3 variant %d x %ld launches beside %ld victim iterations, %ld differing from the first\n", v, nagg, iters, bad);
  return 0;
}
