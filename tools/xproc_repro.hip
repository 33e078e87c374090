// xproc_repro.hip -- library-free repro of a gfx950 co-execution hazard (MI355X, ROCm 7.2): packed fp32 math with an
// operand select (v_pk_fma_f32 / v_pk_mul_f32 ... op_sel:[0,1,0], what hipcc's SLP vectoriser makes of `a.x += g.x * s`
// pairs) returns WRONG low results in lanes 16-31 while another wave on the CU runs bf16 MFMAs interleaved with
// ds_read_b128 fragment reads -- another stream of the same process or another process alike.  Round 4 met it as wrong
// partial sums out of embed.hip's conv1_backward_weight_kernel whenever a second rank ran the exact-split GEMMs.
//   hipcc -O3 --offload-arch=gfx950 tools/xproc_repro.hip -o tools/lab/lab_xproc        (default: SLP packs the victim)
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize ... -o tools/lab/lab_xproc_noslp  (no v_pk_* in the victim)
//   lab_xproc B 4                      victim alone: 0 bad iterations
//   lab_xproc G 6 & lab_xproc B 4      victim beside the reduced GEMM loop: EVERY iteration differs (measured 18674 of 18675)
//   lab_xproc G 6 & lab_xproc_noslp B 4   the same victim without packed math: 0 bad iterations
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// aggressor: the k-loop of an LDS-tiled bf16 GEMM reduced to what a bisection of the library's kernel left (MFMAs + fragment
// reads alone do it; MFMAs alone, global loads, barriers do not): per 32-deep tile 24 MFMAs, a ds_read_b128 pinned behind
// most of them, fragments of 80-byte LDS rows [plane][row] at large immediate offsets, 8 waves as 4 x 2, 120 KB of LDS
__global__ __launch_bounds__(512) void gemm_loop_kernel(float* out, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) char raw[];
  constexpr int ROWB = 80, PLANE = 256 * ROWB, BUF = 3 * PLANE;
  for (int i = threadIdx.x * 16; i < 2 * BUF; i += 512 * 16) *reinterpret_cast<float4*>(raw + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int fa_off = (wm * 32 + r) * ROWB + 16 * h, fb_off = (128 + wn * 64 + r) * ROWB + 16 * h;
  f32x16 hi[2], lo[2];
  for (int e = 0; e < 16; ++e) hi[0][e] = hi[1][e] = lo[0][e] = lo[1][e] = 0.f;
  bf16x8 fa[2][3], fb[2][2][3];
  auto rd = [&](int buf, int s16, int pl, int off) { return *reinterpret_cast<const bf16x8*>(raw + buf * BUF + s16 * 32 + pl * PLANE + off); };
  for (int st = 0; st < 2; ++st)
    for (int pl = 0; pl < 3; ++pl) fa[st][pl] = rd(0, st, pl, fa_off), fb[st][0][pl] = rd(0, st, pl, fb_off), fb[st][1][pl] = rd(0, st, pl, fb_off + 32 * ROWB);
  for (int kt = 0; kt < ktiles; ++kt) {
    const int buf = kt & 1;
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      const int step = s / 12, g = (s % 12) / 6, q = s % 6;
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      if (q < 5) lo[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[step][PA[q]], fb[step][g][PB[q]], lo[g], 0, 0, 0);
      else hi[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[step][PA[q]], fb[step][g][PB[q]], hi[g], 0, 0, 0);
      if (s < 9) {                 // step 1's fragments beside step 0's MFMAs
        const int pl = s % 3, w = s / 3;
        if (w == 0) fa[1][pl] = rd(buf, 1, pl, fa_off); else fb[1][w - 1][pl] = rd(buf, 1, pl, fb_off + (w - 1) * 32 * ROWB);
      }
      if (s >= 15) {               // the next tile's step-0 fragments behind the last MFMAs
        const int f = s - 15, pl = f % 3, w = f / 3;
        if (w == 0) fa[0][pl] = rd(buf ^ 1, 0, pl, fa_off); else fb[0][w - 1][pl] = rd(buf ^ 1, 0, pl, fb_off + (w - 1) * 32 * ROWB);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = 0.f;
  for (int e = 0; e < 16; ++e) t += (hi[0][e] + lo[0][e]) + (hi[1][e] + lo[1][e]);
  out[blockIdx.x * 512 + threadIdx.x] = t;
}

// victim: a deterministic block reduction (256 threads, 12 KB of LDS): a thread accumulates four channels x three inputs over its
// row phase, the phases meet in LDS, one partial [3][C4] float4 per block.  hipcc -O3 turns the six accumulations into
// v_pk_fma_f32 with op_sel / op_sel_hi broadcasts of x0 / x1 / x2; the damaged outputs are exactly the LOW halves of the two
// `op_sel:[0,1,0]` instructions (k = 1, components x and z), lanes 16-31 (q >= 16)
__global__ __launch_bounds__(256) void victim_kernel(int R, int C4, const float4* d, const float* x, float4* part) {
  extern __shared__ float4 red[];
  const int PH = 256 / C4, q = threadIdx.x % C4, ph = threadIdx.x / C4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
  const int r0 = blockIdx.x * 1024, r1 = min(R, r0 + 1024);
  for (int r = r0 + ph; r < r1; r += PH) {
    const float4 g = d[(size_t)r * C4 + q];
    const float x0 = x[(size_t)r * 3], x1 = x[(size_t)r * 3 + 1], x2 = x[(size_t)r * 3 + 2];
    a0.x += g.x * x0, a0.y += g.y * x0, a0.z += g.z * x0, a0.w += g.w * x0;
    a1.x += g.x * x1, a1.y += g.y * x1, a1.z += g.z * x1, a1.w += g.w * x1;
    a2.x += g.x * x2, a2.y += g.y * x2, a2.z += g.z * x2, a2.w += g.w * x2;
  }
  red[(ph * 3 + 0) * C4 + q] = a0, red[(ph * 3 + 1) * C4 + q] = a1, red[(ph * 3 + 2) * C4 + q] = a2;
  __syncthreads();
  if (ph == 0)
    for (int k = 0; k < 3; ++k) {
      float4 t = red[k * C4 + q];
      for (int p = 1; p < PH; ++p) { const float4 u = red[(p * 3 + k) * C4 + q]; t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w; }
      part[((size_t)blockIdx.x * 3 + k) * C4 + q] = t;
    }
}

#ifndef XPROC_NO_MAIN
int main(int argc, char** argv) {
  const char mode = argc > 1 ? argv[1][0] : 'B';
  const double secs = argc > 2 ? atof(argv[2]) : 4.0;
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  hipStream_t s;
  CK(hipStreamCreate(&s));
  if (mode == 'G') {                                    // the aggressor, looped
    const int lds = 2 * 3 * 256 * 80;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_loop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float* out; CK(hipMalloc(&out, 1024 * 512 * 4));
    long n = 0;
    while (elapsed() < secs) {
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gemm_loop_kernel, dim3(208), dim3(512), lds, s, out, 12);
      CK(hipStreamSynchronize(s)); n += 20;
    }
    printf("G: %ld launches of the reduced GEMM loop (%d B of LDS) in %.1f s\n", n, lds, elapsed());
    return 0;
  }
  // the victim, looped: every iteration's bytes against a host-computed truth is not needed -- the kernel is deterministic, so
  // iterations are compared with each other (the first one is the reference; start the aggressor AFTER it for a clean one)
  const int R = 65536, C4 = 32, blocks = R / 1024;
  std::vector<float> hd((size_t)R * C4 * 4), hx((size_t)R * 3);
  unsigned st = 12345u;
  auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hd) v = rnd();
  for (auto& v : hx) v = rnd();
  float4 *d, *part; float* x;
  const size_t pbytes = (size_t)blocks * 3 * C4 * sizeof(float4);
  CK(hipMalloc(&d, hd.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&part, pbytes));
  CK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  std::vector<char> ref(pbytes), cur(pbytes);
  long iters = 0, bad = 0, bad_elems = 0;
  while (elapsed() < secs) {
    CK(hipMemsetAsync(part, 0xff, pbytes, s));
    hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), sizeof(float4) * 3 * 256, s, R, C4, d, x, part);
    CK(hipMemcpyAsync(cur.data(), part, pbytes, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if (iters == 0) ref = cur;
    else if (memcmp(ref.data(), cur.data(), pbytes)) {
      ++bad;
      int shown = 0;
      for (size_t i = 0; i < pbytes / 4; ++i) {
        const bool df = ((const unsigned*)ref.data())[i] != ((const unsigned*)cur.data())[i];
        bad_elems += df;
        if (df && bad <= 1 && shown++ < 8)     // element i = ((block * 3 + k) * C4 + q) * 4 + component
          printf("  it %ld: block %zu k %zu q %zu comp %zu: %.9g (first iteration %.9g)\n", iters, i / (12 * C4), (i / (4 * C4)) % 3,
                 (i / 4) % C4, i % 4, ((const float*)cur.data())[i], ((const float*)ref.data())[i]);
      }
    }
    ++iters;
  }
  printf("B: %ld iterations of the block reduction in %.1f s, %ld with bytes differing from the first (%ld elements)\n", iters, elapsed(), bad, bad_elems);
  return bad ? 1 : 0;
}
#endif
