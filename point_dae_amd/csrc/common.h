// common.h -- shared helpers of the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdae.h"

namespace pdae {

constexpr int kWave = 64;

// Records the reason of the last failing status (thread local, host side).
void set_error(const char* msg);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(hipGetErrorString(e));
    (void)what;
    return PDAE_ERR_LAUNCH;
  }
  return PDAE_OK;
}

inline int bad_arg(const char* msg) {
  set_error(msg);
  return PDAE_ERR_BAD_ARG;
}

inline int unsupported(const char* msg) {
  set_error(msg);
  return PDAE_ERR_UNSUPPORTED;
}

inline hipStream_t as_stream(pdae_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// The squared distance every geometry kernel uses.  Written exactly like the
// reference kernels write it and compiled with -ffp-contract=off:
// ((dx*dx + dy*dy) + dz*dz), each operation rounded.
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by,
                                        float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

// ---- deterministic mode (pdae_set_deterministic, det.hip) --------------------
// Column reductions that normally end in device-scope float atomics (order of arrival) write
// one partial row per block, [partition][width], into the registered workspace instead;
// det_reduce then adds the rows in partition order.  Same arithmetic per block either way.
int& ctx_gemm_arith();   // the current context's GEMM arithmetic (det.hip; -1 = not decided yet)
bool det_on();
// the workspace when the mode is on (nullptr when it is off); *rc = PDAE_ERR_UNSUPPORTED and
// nullptr when the registered buffer is smaller than `bytes`
void* det_workspace(size_t bytes, int* rc);
// outs[k][c] += sum_p part[p][off_k + c], p ascending; up to three output segments of
// lens[k] columns laid side by side in a partial row (null outs are skipped)
int det_reduce(hipStream_t s, int P, int width, const float* part, float* o0, int n0, float* o1 = nullptr,
               int n1 = 0, float* o2 = nullptr, int n2 = 0);
int det_reduce_f64(hipStream_t s, int P, int width, const double* part, double* out);

// ---- deferred column reductions (pdae_deferred_begin / pdae_deferred_flush, det.hip) ----------
// Between begin and flush a reduction site may park its per-block partial rows in the registered buffer
// instead of finishing with atomics (or with its own ordered pass): deferred_take returns the rows' place
// (nullptr when the mode is off or the buffer is full) and records the job; ONE launch at flush adds the
// partials of all parked jobs in block order into their outputs.  Outputs are complete only after the
// flush -- the caller (the graphed step) guarantees nothing reads them earlier.
float* deferred_take(int P, int width, float* o0, int n0, float* o1, int n1, float* o2, int n2);
bool deferred_on();
int rows_wgrad_flush(hipStream_t s);   // rows_gemm.hip: the parked weight-gradient reductions
// embed.hip: S[0][c] = sum_r t, S[1][c] = sum_r t xhat with t = relu'(bn(X)) ? dA : 0 over R rows (the reduction sweep of
// pdae_bnrelu_backward; X rows through `groups` when given); S is overwritten
int bnrelu_backward_sums(hipStream_t s, int R, int C, const float* dA, const float* X, const float* scale, const float* shift,
                         const float* mean, const float* invstd, float* S, const int* groups);

__device__ __forceinline__ void col_add(float* out, float* part, int partition, int width, int c, float t) {
  if (part) part[(size_t)partition * width + c] = t;
  else atomicAdd(out, t);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// ---- write-through result stores (round 6) ------------------------------------------------------------------------
// A kernel's results leave through `sc1` stores: the bytes go to memory while the kernel still runs instead of staying
// dirty in the XCD's L2 until the write-back at the kernel's end, which the NEXT dependent kernel waits for (a boundary
// costs ~1.5 us + dirty bytes / 6 TB/s: MI355X_MICROARCH.md price list, "boundary"; the step is a chain of 330 dependent
// kernels).  The consumer is always another kernel, whose L2 does not keep this one's lines anyway.  Measured on the
// step with the row GEMMs' epilogues alone: -0.11 ms of 10.4 (tools/lab/NOTES.md).  -DPDAE_PLAIN_STORES: plain stores (A/B).
typedef float f32x4_wt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float* p, float v) {
#ifdef PDAE_PLAIN_STORES
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store_dword ... sc1
#endif
}
__device__ __forceinline__ void store_wt4(float* p, const float4& v) {
#ifdef PDAE_PLAIN_STORES
  *reinterpret_cast<float4*>(p) = v;
#else
  const f32x4_wt t = {v.x, v.y, v.z, v.w};
  // (no builtin carries the sc1 bit for a 16-byte store; the trailing s_nop keeps the data registers until the store
  // has read them: cdna_hip_programming.md 5.7)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#endif
}

// exact (erf) GELU of nn.GELU (models/PointCAE_transformer.py:94-110) and its derivative
__device__ __forceinline__ float gelu_f(float v) {
  return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float v) {
  // d/dv [v Phi(v)] = Phi(v) + v phi(v)
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// GELU(v) and GELU'(v) together, for the GEMM epilogues (fc1 of every Transformer block: 48 elements per thread behind the
// k-loop, where erff + expf -- ~55 vector instructions per element, both branches of erff under divergence -- made the
// epilogue a quarter of the launch).  One exponential serves both: with E = exp(-v^2 / 2),
//   Phi(-|v|) = erfc(|v| / sqrt 2) / 2 = E (1 + u R(u)) / 2,  u = p x / (1 + p x),  x = |v| / sqrt 2,  p = 0.6
// R = degree-6 minimax fit of (erfcx(x) - 1) / u on x in [0, 6.6] (error 9e-10 before rounding; in u rather than in
// 1 / (1 + p x) so that the coefficients stay small and v ~ 0 keeps its relative accuracy), phi(v) = E / sqrt(2 pi).
// Measured against fp64 over v in [-9, 9], 4 M points (fp32 evaluation, fused multiply-adds as written): |Phi| 8.0e-8,
// |GELU| 3.8e-7, |GELU'| 1.3e-7 -- the erff / expf formulation above: 6.1e-8, 4.5e-7.  ~22 instructions, two of them
// transcendental (v_rcp_f32, v_exp_f32).
__device__ __forceinline__ void gelu_pair_f(float v, float& gelu, float& grad) {
  // (|v| clamped at 1e18: v = +inf must give GELU = inf, GELU' = 1 as the erff / expf form does, not inf * rcp(inf) = NaN;
  // finite activations are untouched by the clamp)
  const float ax = fminf(fabsf(v) * 0.424264069f, 1e18f);                    // p |v| / sqrt 2
  const float u = ax * __builtin_amdgcn_rcpf(1.0f + ax);
  float r = 0.0685024065f;
  r = __builtin_fmaf(r, u, -0.0438022078f);
  r = __builtin_fmaf(r, u, -0.0923887339f);
  r = __builtin_fmaf(r, u, -0.13927262f);
  r = __builtin_fmaf(r, u, 0.192483393f);
  r = __builtin_fmaf(r, u, 0.897135465f);
  r = __builtin_fmaf(r, u, -1.88063177f);
  const float e = __builtin_amdgcn_exp2f((v * v) * -0.72134752044448170368f);   // exp(-v^2 / 2)
  const float q = (0.5f * __builtin_fmaf(r, u, 1.0f)) * e;                       // Phi(-|v|)
  const float cdf = v >= 0.f ? 1.0f - q : q;
  gelu = v * cdf;
  grad = __builtin_fmaf(fminf(v, 1e18f), 0.39894228040143267794f * e, cdf);
}

// ---- 64-lane shuffles on 64-bit keys (two ds_bpermute each) ----------------
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src_lane) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)(v & 0xffffffffull), src_lane, kWave);
  const unsigned hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src_lane, kWave);
  return ((unsigned long long)hi << 32) | lo;
}

// ---- DPP helpers ----------------------------------------------------------
// dpp_ctrl encodings (gfx9): quad_perm 0x00-0xff, row_shr:n 0x110+n,
// row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = dpp_u32<CTRL>((unsigned)(v & 0xffffffffull));
  const unsigned hi = dpp_u32<CTRL>((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long max_u64(unsigned long long a,
                                                      unsigned long long b) {
  return a > b ? a : b;
}
__device__ __forceinline__ unsigned long long min_u64(unsigned long long a,
                                                      unsigned long long b) {
  return a < b ? a : b;
}

// Max of a 64-bit key over the wave; result is wave-uniform (held in SGPRs).
// In-row butterfly with DPP (xor1, xor2, half-mirror, mirror), then the four
// row values are read with v_readlane and combined on the scalar unit.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = max_u64(v, dpp_u64<0xB1>(v));   // quad_perm [1,0,3,2]
  v = max_u64(v, dpp_u64<0x4E>(v));   // quad_perm [2,3,0,1]
  v = max_u64(v, dpp_u64<0x141>(v));  // row_half_mirror
  v = max_u64(v, dpp_u64<0x140>(v));  // row_mirror
  const unsigned lo = (unsigned)(v & 0xffffffffull), hi = (unsigned)(v >> 32);
  unsigned long long r = 0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const unsigned l = (unsigned)__builtin_amdgcn_readlane((int)lo, row * 16);
    const unsigned h = (unsigned)__builtin_amdgcn_readlane((int)hi, row * 16);
    r = max_u64(r, ((unsigned long long)h << 32) | l);
  }
  return r;
}

__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
  return v;
}

}  // namespace pdae
