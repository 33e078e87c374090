// block.hip -- the row-wise (memory-bound) pieces of a pre-LN Transformer block,
// each as one fused sweep.  Block.forward of the reference
// (models/PointCAE_transformer.py:155-158 with :174-177):
//     x = x + pos;  x = x + dp(attn(ln1(x)));  x = x + dp(mlp(ln2(x)))
// PyTorch runs add, LayerNorm, bias, GELU, DropPath's div/mul, the residual adds
// and every backward twin as separate kernels (~80 launches per block and step).
//   add_layernorm_fwd   s = x (+ pos);  y = LN(s);  saves mean / rstd
//   layernorm_bwd       dx = LN'(dy) (+ skip-connection gradient), dgamma/dbeta
//   gelu_fwd / gelu_bwd exact (erf) GELU and its derivative
//   scale_residual      y = res + keep[b] * (a + bias)   (DropPath + bias + skip)
//   rowscale            da = keep[b] * dy                (its backward)
//   colsum              bias gradients
// One wave per row for the LayerNorm kernels (C <= 2048), float4 everywhere.
#include "common.h"

namespace pdae {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

constexpr int LN_MAX4 = 8;  // float4 per lane: C <= 2048

// s = x (+ keep[row/T] * (br + bias): the previous sub-layer's branch, i.e. scale_residual)
//       (+ pos);  y = LN(s).  s is written to xsum whenever it differs from x.
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(
    int M, int C, const float* __restrict__ x, const float* __restrict__ pos,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float* __restrict__ xsum, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, const float* __restrict__ br, const float* __restrict__ bias,
    const float* __restrict__ keep, int T, int br_slabs) {
  // br_slabs > 1: the branch arrives as split-K slabs [br_slabs][M][C] of partial products
  // (rows_gemm.hip), added up here in slab order
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = lane_id();
  const int n4 = C >> 2;
  const float k = keep ? keep[row / T] : 1.f;
  float4 v[LN_MAX4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX4; ++i) {
    const int c4 = lane + i * kWave;
    if (c4 < n4) {
      float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * C + c4 * 4);
      if (br) {      // same operation order as scale_residual: res + keep * (a + bias)
        float4 t = *reinterpret_cast<const float4*>(br + (size_t)row * C + c4 * 4);
        for (int q = 1; q < br_slabs; ++q) {
          const float4 u = *reinterpret_cast<const float4*>(br + ((size_t)q * M + row) * C + c4 * 4);
          t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
        }
        if (bias) {
          const float4 b = *reinterpret_cast<const float4*>(bias + c4 * 4);
          t.x += b.x, t.y += b.y, t.z += b.z, t.w += b.w;
        }
        if (keep) t.x *= k, t.y *= k, t.z *= k, t.w *= k;
        a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
      }
      if (pos) {
        const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)row * C + c4 * 4);
        a.x += p.x, a.y += p.y, a.z += p.z, a.w += p.w;
      }
      if (pos || br) store_wt4(xsum + (size_t)row * C + c4 * 4, a);
      v[i] = a;
      s += (a.x + a.y) + (a.z + a.w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX4; ++i) {
    const int c4 = lane + i * kWave;
    if (c4 < n4) {
      const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < LN_MAX4; ++i) {
    const int c4 = lane + i * kWave;
    if (c4 < n4) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
      const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      store_wt4(y + (size_t)row * C + c4 * 4, o);
    }
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dres);
// dgamma += sum dy*xhat, dbeta += sum dy over this block's rows (LDS, then atomics).
// One wave per row, NS float4 slots per lane (C <= 256*NS), NW waves per block, `rows`
// rows per block (a multiple of NW).  The pass is latency-bound, not bandwidth-bound
// (12-50 MB per launch): every load of a row -- dy, x, the skip gradient -- is issued
// before the first reduction and gamma stays in registers across rows.  The 2*C global
// atomics per BLOCK are what the launch pays for (device-scope float atomics run at
// ~28 G/s on 24 cache lines: 2048 blocks cost 55 us, 128 blocks 3 us), hence 16-wave
// blocks: few blocks, many rows in flight per CU.
template <int NS, int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_kernel(
    int M, int C, int rows, const float* __restrict__ dy, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ dres, float* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, const float* __restrict__ keep, int T,
    float* __restrict__ da, float* __restrict__ dbias, int dy_slabs, float* __restrict__ dacc, int dacc_mode,
    float* __restrict__ part) {
  // (part: deterministic mode -- this block's 2-3 column partials go to row blockIdx.x of it)
  // (dy_slabs > 1: dy arrives as split-K slabs [dy_slabs][M][C], added up in slab order;
  //  dacc: a second copy of dx that is written (mode 1) or added to (mode 2): the gradient of the
  //  position embedding, which every block of a stack re-adds, sums over the blocks right here)
  // (keep / da / dbias: the backward of the branch folded into the forward -- da =
  //  keep[row/T] * dx, dbias += column sums of da; see residual_layernorm_backward)
  extern __shared__ float red[];  // [NW][3][C]: one plain-store slot per wave (LDS float
                                  // atomics serialise: 256 of them cost 12 us per block)
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int n4 = C >> 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ag[NS], ab[NS], g[NS], ac[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    ag[i] = ab[i] = ac[i] = zero4;
    const int c4 = lane + i * kWave;
    g[i] = c4 < n4 ? *reinterpret_cast<const float4*>(gamma + c4 * 4) : zero4;
  }
  float4 d[NS], xv[NS], e[NS];
  float mu = 0.f, rs = 0.f, kp = 1.f;
  auto load_row = [&](int row) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int c4 = lane + i * kWave;
      const bool on = c4 < n4 && row < M;
      const size_t off = (size_t)row * C + c4 * 4;
      d[i] = on ? *reinterpret_cast<const float4*>(dy + off) : zero4;
      for (int q = 1; q < dy_slabs; ++q) {
        const float4 u = on ? *reinterpret_cast<const float4*>(dy + (size_t)q * M * C + off) : zero4;
        d[i].x += u.x, d[i].y += u.y, d[i].z += u.z, d[i].w += u.w;
      }
      xv[i] = on ? *reinterpret_cast<const float4*>(x + off) : zero4;
      e[i] = (on && dres) ? *reinterpret_cast<const float4*>(dres + off) : zero4;
    }
    mu = row < M ? mean[row] : 0.f;
    rs = row < M ? rstd[row] : 0.f;
    kp = (keep && row < M) ? keep[row / T] : 1.f;
  };
  const int row0 = blockIdx.x * rows + w;
  load_row(row0);
  for (int rr = 0; rr < rows / NW; ++rr) {
    const int row = row0 + rr * NW;
    if (row >= M) break;
    float4 gd[NS], xh[NS];
    float s1 = 0.f, s2 = 0.f;
    const float rs_ = rs, kp_ = kp;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      // (slots past the row hold zeros: xh = -mu*rs there, but d = gd = 0 keeps every sum exact)
      xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
      gd[i] = make_float4(d[i].x * g[i].x, d[i].y * g[i].y, d[i].z * g[i].z, d[i].w * g[i].w);
      s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
      s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
      ag[i].x += d[i].x * xh[i].x, ag[i].y += d[i].y * xh[i].y, ag[i].z += d[i].z * xh[i].z, ag[i].w += d[i].w * xh[i].w;
      ab[i].x += d[i].x, ab[i].y += d[i].y, ab[i].z += d[i].z, ab[i].w += d[i].w;
    }
    float4 sk[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) sk[i] = e[i];
    if (rr + 1 < rows / NW) load_row(row + NW);      // next row's loads fly during the reductions
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int c4 = lane + i * kWave;
      if (c4 < n4) {
        float4 o;
        o.x = rs_ * (gd[i].x - m1 - xh[i].x * m2) + sk[i].x;
        o.y = rs_ * (gd[i].y - m1 - xh[i].y * m2) + sk[i].y;
        o.z = rs_ * (gd[i].z - m1 - xh[i].z * m2) + sk[i].z;
        o.w = rs_ * (gd[i].w - m1 - xh[i].w * m2) + sk[i].w;
        store_wt4(dx + (size_t)row * C + c4 * 4, o);
        if (dacc_mode == 1) {
          store_wt4(dacc + (size_t)row * C + c4 * 4, o);
        } else if (dacc_mode == 2) {
          float4 t = *reinterpret_cast<const float4*>(dacc + (size_t)row * C + c4 * 4);
          t.x += o.x, t.y += o.y, t.z += o.z, t.w += o.w;
          *reinterpret_cast<float4*>(dacc + (size_t)row * C + c4 * 4) = t;
        }
        if (dbias) {
          if (keep) o.x *= kp_, o.y *= kp_, o.z *= kp_, o.w *= kp_;
          if (da) store_wt4(da + (size_t)row * C + c4 * 4, o);
          ac[i].x += o.x, ac[i].y += o.y, ac[i].z += o.z, ac[i].w += o.w;
        }
      }
    }
  }
  const int nred = dbias ? 3 : 2;
  float* mine = red + (size_t)w * nred * C;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int c4 = lane + i * kWave;
    if (c4 < n4) {
      *reinterpret_cast<float4*>(mine + c4 * 4) = ag[i];
      *reinterpret_cast<float4*>(mine + C + c4 * 4) = ab[i];
      if (dbias) *reinterpret_cast<float4*>(mine + 2 * C + c4 * 4) = ac[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nred * C; c += NW * 64) {
    float t = 0.f;
#pragma unroll 4
    for (int k = 0; k < NW; ++k) t += red[(size_t)k * nred * C + c];
    col_add(c < C ? dgamma + c : (c < 2 * C ? dbeta + (c - C) : dbias + (c - 2 * C)), part, blockIdx.x,
            nred * C, c, t);
  }
}

// h = gelu(z + bias[col]); backward: dz = dh * gelu'(z + bias), dbias += colsum(dz)
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(long long n4, int C4,
                                                            const float4* __restrict__ z,
                                                            const float4* __restrict__ bias,
                                                            float4* __restrict__ h) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = z[i], b = bias[i % C4];
  h[i] = make_float4(gelu_f(v.x + b.x), gelu_f(v.y + b.y), gelu_f(v.z + b.z), gelu_f(v.w + b.w));
}

// Column-reducing sweeps (bias_gelu_bwd, scale_colsum, colsum2): one block = 64 column
// quads x CS_NW row phases over a slab of CS_ROWS rows.  16 waves with 4 independent rows
// each keep enough loads in flight (these passes are latency-bound at M = 3-8 k rows) while
// the block count -- the number of global atomics -- stays small.
constexpr int CS_NW = 16, CS_ROWS = 4 * CS_NW;
__device__ __forceinline__ void colsum_finish(float4 s, float4 (*part)[64], float* __restrict__ out, int c,
                                              int N, float* __restrict__ det) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < N && out) {
    float4 t = part[0][lane];
#pragma unroll
    for (int k = 1; k < CS_NW; ++k) {
      const float4 u = part[k][lane];
      t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
    }
    if (det) {   // deterministic mode: row blockIdx.y of the partial matrix
      *reinterpret_cast<float4*>(det + (size_t)blockIdx.y * N + c) = t;
    } else {
      atomicAdd(out + c + 0, t.x), atomicAdd(out + c + 1, t.y);
      atomicAdd(out + c + 2, t.z), atomicAdd(out + c + 3, t.w);
    }
  }
}

__global__ __launch_bounds__(CS_NW * 64) void bias_gelu_bwd_kernel(int M, int C, const float* __restrict__ z,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ dh,
                                                            float* __restrict__ dz,
                                                            float* __restrict__ dbias,
                                                            int rows_per_split, float* __restrict__ det) {
  __shared__ float4 part[CS_NW][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int mbeg = blockIdx.y * rows_per_split, mend = min(M, mbeg + rows_per_split);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll 4
    for (int m = mbeg + w; m < mend; m += CS_NW) {
      const float4 v = *reinterpret_cast<const float4*>(z + (size_t)m * C + c);
      const float4 d = *reinterpret_cast<const float4*>(dh + (size_t)m * C + c);
      float4 o;
      o.x = d.x * gelu_grad_f(v.x + b.x);
      o.y = d.y * gelu_grad_f(v.y + b.y);
      o.z = d.z * gelu_grad_f(v.z + b.z);
      o.w = d.w * gelu_grad_f(v.w + b.w);
      *reinterpret_cast<float4*>(dz + (size_t)m * C + c) = o;
      s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
    }
  }
  colsum_finish(s, part, dbias, c, C, det);
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(long long n4, const float4* __restrict__ z,
                                                       float4* __restrict__ h) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = z[i];
  h[i] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(long long n4, const float4* __restrict__ z,
                                                       const float4* __restrict__ dh,
                                                       float4* __restrict__ dz) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = z[i], d = dh[i];
  dz[i] = make_float4(d.x * gelu_grad_f(v.x), d.y * gelu_grad_f(v.y), d.z * gelu_grad_f(v.z),
                      d.w * gelu_grad_f(v.w));
}

// y[row] = res[row] + keep[row / T] * (a[row] + bias)     (keep, bias, res nullable)
__global__ __launch_bounds__(256) void scale_residual_kernel(long long n4, int C4, int T,
                                                             const float4* __restrict__ a,
                                                             const float4* __restrict__ bias,
                                                             const float* __restrict__ keep,
                                                             const float4* __restrict__ res,
                                                             float4* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long row = i / C4;
  float4 v = a[i];
  if (bias) {
    const float4 b = bias[i - row * C4];
    v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
  }
  if (keep) {
    const float k = keep[row / T];
    v.x *= k, v.y *= k, v.z *= k, v.w *= k;
  }
  if (res) {
    const float4 r = res[i];
    v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
  }
  y[i] = v;
}

// out[c] += sum over a slab of rows; lane owns 4 adjacent columns, the waves stride the rows.
// Narrow matrices (N <= 128): LPR = N/4 lanes span a row and a wave reads 64/LPR rows at once.
template <int LPR>
__global__ __launch_bounds__(CS_NW * 64) void colsum2_kernel(int M, int N, const float* __restrict__ X,
                                                             float* __restrict__ out, int rows_per_split,
                                                             float* __restrict__ det) {
  __shared__ float4 part[CS_NW][64];
  constexpr int RPW = 64 / LPR;                       // rows per wave and pass
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + (lane % LPR)) * 4;
  const int mbeg = blockIdx.y * rows_per_split, mend = min(M, mbeg + rows_per_split);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
#pragma unroll 4
    for (int m = mbeg + w * RPW + lane / LPR; m < mend; m += CS_NW * RPW) {
      const float4 v = *reinterpret_cast<const float4*>(X + (size_t)m * N + c);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  if (LPR < 64) {                                     // fold the lanes that share a column
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) {
      s.x += __shfl_xor(s.x, o, kWave), s.y += __shfl_xor(s.y, o, kWave);
      s.z += __shfl_xor(s.z, o, kWave), s.w += __shfl_xor(s.w, o, kWave);
    }
    if (lane >= LPR) s = make_float4(0.f, 0.f, 0.f, 0.f), part[w][lane] = s;
  }
  colsum_finish(s, part, (LPR < 64 && lane >= LPR) ? nullptr : out, c, N, det);
}

// Y[m][c] = keep[m/T] * X[m][c] and out[c] += column sums of Y: the backward of
// scale_residual (DropPath scaling of the branch gradient + the Linear bias gradient) in
// one pass over the gradient instead of a scaling pass and a column-sum pass.
__global__ __launch_bounds__(CS_NW * 64) void scale_colsum_kernel(int M, int N, int T,
                                                                  const float* __restrict__ X,
                                                                  const float* __restrict__ keep,
                                                                  float* __restrict__ Y, float* __restrict__ out,
                                                                  int rows_per_split, float* __restrict__ det) {
  __shared__ float4 part[CS_NW][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int mbeg = blockIdx.y * rows_per_split, mend = min(M, mbeg + rows_per_split);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
#pragma unroll 4
    for (int m = mbeg + w; m < mend; m += CS_NW) {
      float4 v = *reinterpret_cast<const float4*>(X + (size_t)m * N + c);
      const float k = keep[m / T];
      v.x *= k, v.y *= k, v.z *= k, v.w *= k;
      *reinterpret_cast<float4*>(Y + (size_t)m * N + c) = v;
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  colsum_finish(s, part, out, c, N, det);
}

}  // namespace pdae

using namespace pdae;

// rows per block of the column-reducing sweeps: CS_ROWS, more once that would exceed 256 row
// slabs -- every slab ends in atomics on the same few cache lines (2048 slabs of a 128-column
// matrix: 200 us of atomics for a 30 us read)
static int cs_rows(int M) {
  int rows = CS_ROWS;
  if ((M + rows - 1) / rows > 256) rows = CS_NW * ((M + 256 * CS_NW - 1) / (256 * CS_NW));
  return rows;
}

static int ln_forward(const char* what, int M, int C, int T, const float* x, const float* br,
                      const float* bias, const float* keep, const float* pos, const float* gamma,
                      const float* beta, float eps, float* xsum, float* y, float* mean, float* rstd,
                      int br_slabs, pdae_stream_t stream) {
  if (M < 0 || C <= 0 || C % 4 != 0 || C > 4 * kWave * LN_MAX4 || T <= 0 || br_slabs < 1 || br_slabs > 8)
    return bad_arg("layernorm forward: C must be a multiple of 4, at most 2048; T > 0; 1..8 slabs");
  if (M == 0) return PDAE_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || ((pos || br) && !xsum) || ((bias || keep) && !br))
    return bad_arg("layernorm forward: null pointer");
  hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, as_stream(stream), M,
                     C, x, pos, gamma, beta, eps, xsum, y, mean, rstd, br, bias, keep, T, br_slabs);
  return check_launch(what);
}

extern "C" int pdae_add_layernorm_forward(int M, int C, const float* x, const float* pos,
                                          const float* gamma, const float* beta, float eps,
                                          float* xsum, float* y, float* mean, float* rstd,
                                          pdae_stream_t stream) {
  return ln_forward("add_layernorm_forward", M, C, 1, x, nullptr, nullptr, nullptr, pos, gamma, beta, eps,
                    xsum, y, mean, rstd, 1, stream);
}

extern "C" int pdae_residual_layernorm_forward(int M, int C, int T, const float* a, int a_slabs, const float* bias,
                                               const float* keep, const float* res, const float* pos,
                                               const float* gamma, const float* beta, float eps,
                                               float* xsum, float* y, float* mean, float* rstd,
                                               pdae_stream_t stream) {
  if (!a) return bad_arg("residual_layernorm_forward: null pointer");
  return ln_forward("residual_layernorm_forward", M, C, T, res, a, bias, keep, pos, gamma, beta, eps, xsum,
                    y, mean, rstd, a_slabs, stream);
}

static int ln_backward(const char* what, int M, int C, int T, const float* dy, int dy_slabs, const float* x,
                       const float* mean, const float* rstd, const float* gamma, const float* dres,
                       const float* keep, float* dx, float* da, float* dgamma, float* dbeta, float* dbias,
                       int accumulate, float* dacc, int dacc_mode, pdae_stream_t stream) {
  if (dacc_mode < 0 || dacc_mode > 2 || (dacc_mode && !dacc)) return bad_arg("layernorm backward: bad dacc");
  if (M < 0 || C <= 0 || C % 4 != 0 || C > 4 * kWave * LN_MAX4 || T <= 0 || dy_slabs < 1 || dy_slabs > 8)
    return bad_arg("layernorm backward: C must be a multiple of 4, at most 2048; T > 0; 1..8 slabs");
  if (!dgamma || !dbeta) return bad_arg("layernorm backward: null pointer");
  hipStream_t s = as_stream(stream);
  if (!accumulate) {
    (void)hipMemsetAsync(dgamma, 0, sizeof(float) * C, s);
    (void)hipMemsetAsync(dbeta, 0, sizeof(float) * C, s);
    if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * C, s);
  }
  if (M == 0) return check_launch(what);
  if (!dy || !x || !mean || !rstd || !gamma || !dx || ((keep || da) && !dbias) || (keep && !da))
    return bad_arg("layernorm backward: null pointer");
  const int nred = dbias ? 3 : 2;
  int rows, blocks;
  if (C <= 512) {
    // 16-wave blocks, one row per wave (two from 6144 rows on: fewer blocks = fewer atomics);
    // measured 13 us at M = 2944 and 20 us at M = 8192 (was 24 / 29 us with LDS atomics)
    rows = 16 * (M >= 6144 ? 2 : 1);
  } else {
    rows = 4 * ((M + 4 * 256 - 1) / (4 * 256));
    if (rows > 32) rows = 32;
  }
  blocks = (M + rows - 1) / rows;
  int rc = PDAE_OK;
  float* part = static_cast<float*>(det_workspace(sizeof(float) * (size_t)blocks * nred * C, &rc));
  if (rc) return rc;
  // between pdae_deferred_begin and _flush the partial rows are parked and added by the flush's one launch
  // (the parameter gradients are then complete only after it): no atomics, no reduction launch per LayerNorm
  bool parked = false;
  if (!part && (part = deferred_take(blocks, nred * C, dgamma, C, dbeta, C, dbias, dbias ? C : 0))) parked = true;
  if (C <= 512)
    hipLaunchKernelGGL((layernorm_bwd_kernel<2, 16>), dim3(blocks), dim3(1024), 16 * nred * C * sizeof(float), s, M,
                       C, rows, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, keep, T, da, dbias, dy_slabs,
                       dacc, dacc_mode, part);
  else
    hipLaunchKernelGGL((layernorm_bwd_kernel<LN_MAX4, 4>), dim3(blocks), dim3(256), 4 * nred * C * sizeof(float), s,
                       M, C, rows, dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, keep, T, da, dbias, dy_slabs,
                       dacc, dacc_mode, part);
  if (part && !parked) return det_reduce(s, blocks, nred * C, part, dgamma, C, dbeta, C, dbias, dbias ? C : 0);
  return check_launch(what);
}

extern "C" int pdae_layernorm_backward(int M, int C, const float* dy, int dy_slabs, const float* x,
                                       const float* mean, const float* rstd, const float* gamma,
                                       const float* dres, float* dx, float* dgamma, float* dbeta,
                                       int accumulate, float* dacc, int dacc_mode, pdae_stream_t stream) {
  return ln_backward("layernorm_backward", M, C, 1, dy, dy_slabs, x, mean, rstd, gamma, dres, nullptr, dx, nullptr,
                     dgamma, dbeta, nullptr, accumulate, dacc, dacc_mode, stream);
}

extern "C" int pdae_residual_layernorm_backward(int M, int C, int T, const float* dy, int dy_slabs, const float* x,
                                                const float* mean, const float* rstd,
                                                const float* gamma, const float* dres,
                                                const float* keep, float* dx, float* da, float* dgamma,
                                                float* dbeta, float* dbias, int accumulate,
                                                float* dacc, int dacc_mode, pdae_stream_t stream) {
  if (!dbias) return bad_arg("residual_layernorm_backward: null pointer");
  return ln_backward("residual_layernorm_backward", M, C, T, dy, dy_slabs, x, mean, rstd, gamma, dres, keep, dx, da,
                     dgamma, dbeta, dbias, accumulate, dacc, dacc_mode, stream);
}

extern "C" int pdae_gelu_forward(long long n, const float* z, float* h, pdae_stream_t stream) {
  if (n < 0 || n % 4 != 0) return bad_arg("gelu_forward: n must be a multiple of 4");
  if (n == 0) return PDAE_OK;
  if (!z || !h) return bad_arg("gelu_forward: null pointer");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), n / 4, reinterpret_cast<const float4*>(z),
                     reinterpret_cast<float4*>(h));
  return check_launch("gelu_forward");
}

extern "C" int pdae_gelu_backward(long long n, const float* z, const float* dh, float* dz,
                                  pdae_stream_t stream) {
  if (n < 0 || n % 4 != 0) return bad_arg("gelu_backward: n must be a multiple of 4");
  if (n == 0) return PDAE_OK;
  if (!z || !dh || !dz) return bad_arg("gelu_backward: null pointer");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), n / 4, reinterpret_cast<const float4*>(z),
                     reinterpret_cast<const float4*>(dh), reinterpret_cast<float4*>(dz));
  return check_launch("gelu_backward");
}

extern "C" int pdae_bias_gelu_forward(int M, int C, const float* z, const float* bias, float* h,
                                      pdae_stream_t stream) {
  if (M < 0 || C <= 0 || C % 4 != 0) return bad_arg("bias_gelu_forward: C must be a positive multiple of 4");
  if (M == 0) return PDAE_OK;
  if (!z || !bias || !h) return bad_arg("bias_gelu_forward: null pointer");
  const long long n4 = (long long)M * (C / 4);
  hipLaunchKernelGGL(bias_gelu_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), n4, C / 4, reinterpret_cast<const float4*>(z),
                     reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(h));
  return check_launch("bias_gelu_forward");
}

extern "C" int pdae_bias_gelu_backward(int M, int C, const float* z, const float* bias,
                                       const float* dh, float* dz, float* dbias, int accumulate,
                                       pdae_stream_t stream) {
  if (M < 0 || C <= 0 || C % 4 != 0) return bad_arg("bias_gelu_backward: C must be a positive multiple of 4");
  if (!dbias) return bad_arg("bias_gelu_backward: null pointer");
  hipStream_t s = as_stream(stream);
  if (!accumulate) (void)hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)C, s);
  if (M == 0) return check_launch("bias_gelu_backward");
  if (!z || !bias || !dh || !dz) return bad_arg("bias_gelu_backward: null pointer");
  const int rows = cs_rows(M), P = (M + rows - 1) / rows;
  int rc = PDAE_OK;
  float* det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)P * C, &rc));
  if (rc) return rc;
  hipLaunchKernelGGL(bias_gelu_bwd_kernel, dim3((C + 255) / 256, P), dim3(CS_NW * 64), 0, s, M, C, z, bias, dh, dz,
                     dbias, rows, det);
  if (det) return det_reduce(s, P, C, det, dbias, C);
  return check_launch("bias_gelu_backward");
}

extern "C" int pdae_scale_residual(int M, int C, int T, const float* a, const float* bias,
                                   const float* keep, const float* res, float* y,
                                   pdae_stream_t stream) {
  if (M < 0 || C <= 0 || C % 4 != 0 || T <= 0) return bad_arg("scale_residual: bad size");
  if (M == 0) return PDAE_OK;
  if (!a || !y) return bad_arg("scale_residual: null pointer");
  const long long n4 = (long long)M * (C / 4);
  hipLaunchKernelGGL(scale_residual_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     as_stream(stream), n4, C / 4, T, reinterpret_cast<const float4*>(a),
                     reinterpret_cast<const float4*>(bias), keep, reinterpret_cast<const float4*>(res),
                     reinterpret_cast<float4*>(y));
  return check_launch("scale_residual");
}

extern "C" int pdae_colsum(int M, int N, const float* X, float* out, int accumulate,
                           pdae_stream_t stream) {
  if (M < 0 || N <= 0 || N % 4 != 0) return bad_arg("colsum: N must be a positive multiple of 4");
  if (!out) return bad_arg("colsum: null pointer");
  hipStream_t s = as_stream(stream);
  if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);
  if (M == 0) return check_launch("colsum");
  if (!X) return bad_arg("colsum: null pointer");
  const int rows = cs_rows(M), P = (M + rows - 1) / rows;
  int rc = PDAE_OK;
  float* det = static_cast<float*>(det_workspace(sizeof(float) * (size_t)P * N, &rc));
  if (rc) return rc;
  if (N == 128)
    hipLaunchKernelGGL(colsum2_kernel<32>, dim3(1, P), dim3(CS_NW * 64), 0, s, M, N, X, out, rows, det);
  else if (N == 64)
    hipLaunchKernelGGL(colsum2_kernel<16>, dim3(1, P), dim3(CS_NW * 64), 0, s, M, N, X, out, rows, det);
  else
    hipLaunchKernelGGL(colsum2_kernel<64>, dim3((N + 255) / 256, P), dim3(CS_NW * 64), 0, s, M, N, X, out, rows, det);
  if (det) return det_reduce(s, P, N, det, out, N);
  return check_launch("colsum");
}

extern "C" int pdae_scale_colsum(int M, int N, int T, const float* X, const float* keep, float* Y,
                                 float* out, int accumulate, pdae_stream_t stream) {
  if (M < 0 || N <= 0 || N % 4 != 0 || T <= 0) return bad_arg("scale_colsum: N must be a positive multiple of 4, T > 0");
  hipStream_t s = as_stream(stream);
  if (out && !accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);
  if (M == 0) return check_launch("scale_colsum");
  if (!X || !keep || !Y) return bad_arg("scale_colsum: null pointer");
  const int rows = cs_rows(M), P = (M + rows - 1) / rows;
  int rc = PDAE_OK;
  float* det = out ? static_cast<float*>(det_workspace(sizeof(float) * (size_t)P * N, &rc)) : nullptr;
  if (rc) return rc;
  hipLaunchKernelGGL(scale_colsum_kernel, dim3((N + 255) / 256, P), dim3(CS_NW * 64), 0, s, M, N, T, X, keep, Y,
                     out, rows, det);
  if (det) return det_reduce(s, P, N, det, out, N);
  return check_launch("scale_colsum");
}


// ---- small fused launches that replace chains of framework elementwise kernels inside the step -----------------
namespace pdae {

// timm 0.4.5 DropPath for all stochastic-depth sites of a stack at once: r (S, B) uniforms in, per-sample factors
// floor(r + keep_s) / keep_s out (in place allowed) -- the reference's `x.div(keep) * (keep + rand).floor_()`.
__global__ __launch_bounds__(256) void drop_path_keep_kernel(int n, int B, const float* __restrict__ r,
                                                             const float* __restrict__ keep, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float k = keep[i / B];
  out[i] = floorf(r[i] + k) / k;
}

// First layer of pos_embed (models/PointCAE_transformer.py:329-333: Linear(3,128) -> GELU) on gathered centre rows:
// z = xyz[rows[m]] . W1^T + b1 with K = 3 as an fma chain in k order (what the row GEMM computes on the zero-padded
// operands), h = GELU(z), gp = GELU'(z); also leaves the gathered rows zero-padded to 4 columns for the weight-gradient
// GEMM of the backward.  One thread per (row, 4 channels).
__global__ __launch_bounds__(256) void pos_embed_fc1_kernel(int M, int H, const float* __restrict__ xyz,
                                                            const long long* __restrict__ rows,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            float* __restrict__ h, float* __restrict__ gp,
                                                            float* __restrict__ xp) {
  const int q = H / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * q) return;
  const int m = (int)(i / q), c = (int)(i % q) * 4;
  const long long src = rows ? rows[m] : m;
  const float x0 = xyz[src * 3 + 0], x1 = xyz[src * 3 + 1], x2 = xyz[src * 3 + 2];
  if (c == 0) *reinterpret_cast<float4*>(xp + (size_t)m * 4) = make_float4(x0, x1, x2, 0.f);
  float hv[4], gv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* w = w1 + (size_t)(c + e) * 3;
    const float z = fmaf(x2, w[2], fmaf(x1, w[1], x0 * w[0])) + b1[c + e];
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
    hv[e] = z * cdf, gv[e] = cdf + z * pdf;
  }
  *reinterpret_cast<float4*>(h + (size_t)m * H + c) = make_float4(hv[0], hv[1], hv[2], hv[3]);
  *reinterpret_cast<float4*>(gp + (size_t)m * H + c) = make_float4(gv[0], gv[1], gv[2], gv[3]);
}

// The decoder's last block returns only the last `tail` tokens of every sample (TransformerDecoder.forward,
// models/PointCAE_transformer.py:225-232: x[:, -return_token_num:]): its row-wise second half runs on those rows alone.
// gather: out[b tail + t] = in[b T + (T - tail) + t] for up to two tensors in one launch; scatter: the reverse into
// full-size gradients, zero outside the tail (the framework ran two strided copies forward, a fill and four copies back).
__global__ __launch_bounds__(256) void tail_rows_gather_kernel(long long n4, int C4, int T, int tail, const float4* __restrict__ a,
                                                               const float4* __restrict__ b, float4* __restrict__ a_t,
                                                               float4* __restrict__ b_t) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long row = i / C4;
  const int c = (int)(i - row * C4);
  const long long smp = row / tail;
  const long long src = (smp * T + (T - tail) + (row - smp * tail)) * C4 + c;
  a_t[i] = a[src];
  if (b) b_t[i] = b[src];
}
__global__ __launch_bounds__(256) void tail_rows_scatter_kernel(long long n4, int C4, int T, int tail, const float4* __restrict__ da_t,
                                                                const float4* __restrict__ db_t, float4* __restrict__ da,
                                                                float4* __restrict__ db) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over the FULL rows
  if (i >= n4) return;
  const long long row = i / C4;
  const int c = (int)(i - row * C4);
  const long long smp = row / T;
  const int t = (int)(row - smp * T) - (T - tail);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long src = (smp * tail + t) * C4 + c;
  da[i] = t >= 0 ? da_t[src] : zero;
  if (db) db[i] = t >= 0 ? db_t[src] : zero;
}

// Y = epi(sum_s slabs[s] + bias): the consumer of split-K slabs where no LayerNorm follows (the coarse heads' Linear
// layers on a handful of rows: 32 x 1024 x 1024 as ONE 64 x 128 tile per 128 columns keeps 8 CUs busy for 32 k-tiles;
// in 8 slabs it is 64 blocks of 4 k-tiles + this pass).  Slabs are added in slab order.  epi 0 | 1 ReLU | 4 Z > 0 ? . : 0.
__global__ __launch_bounds__(256) void slab_sum_epi_kernel(long long n4, int N4, int S, const float4* __restrict__ slabs,
                                                           const float4* __restrict__ bias, int epi,
                                                           const float4* __restrict__ z, float4* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = slabs[i];
  for (int q = 1; q < S; ++q) {
    const float4 v = slabs[(size_t)q * n4 + i];
    a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
  }
  if (bias) {
    const float4 b = bias[i % N4];
    a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
  }
  if (epi == 1) a.x = a.x > 0.f ? a.x : 0.f, a.y = a.y > 0.f ? a.y : 0.f, a.z = a.z > 0.f ? a.z : 0.f, a.w = a.w > 0.f ? a.w : 0.f;
  if (epi == 4) {
    const float4 m = z[i];
    a.x = m.x > 0.f ? a.x : 0.f, a.y = m.y > 0.f ? a.y : 0.f, a.z = m.z > 0.f ? a.z : 0.f, a.w = m.w > 0.f ? a.w : 0.f;
  }
  y[i] = a;
}

}  // namespace pdae

static int tail_check(int B, int T, int tail, int C) {
  if (B < 0 || T <= 0 || tail <= 0 || tail > T || C <= 0 || C % 4 != 0) return bad_arg("tail_rows: B >= 0, 0 < tail <= T, C a positive multiple of 4");
  return PDAE_OK;
}
extern "C" int pdae_tail_rows_gather(int B, int T, int tail, int C, const float* a, const float* b, float* a_t, float* b_t,
                                     pdae_stream_t stream) {
  using namespace pdae;
  int rc = tail_check(B, T, tail, C);
  if (rc) return rc;
  if (B == 0) return PDAE_OK;
  if (!a || !a_t || (b && !b_t)) return bad_arg("tail_rows_gather: null pointer");
  const long long n4 = (long long)B * tail * (C / 4);
  hipLaunchKernelGGL(tail_rows_gather_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4, C / 4, T, tail,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(a_t),
                     reinterpret_cast<float4*>(b_t));
  return check_launch("tail_rows_gather");
}
extern "C" int pdae_tail_rows_scatter(int B, int T, int tail, int C, const float* da_t, const float* db_t, float* da, float* db,
                                      pdae_stream_t stream) {
  using namespace pdae;
  int rc = tail_check(B, T, tail, C);
  if (rc) return rc;
  if (B == 0) return PDAE_OK;
  if (!da_t || !da || (db && !db_t)) return bad_arg("tail_rows_scatter: null pointer");
  const long long n4 = (long long)B * T * (C / 4);
  hipLaunchKernelGGL(tail_rows_scatter_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4, C / 4, T, tail,
                     reinterpret_cast<const float4*>(da_t), reinterpret_cast<const float4*>(db_t), reinterpret_cast<float4*>(da),
                     reinterpret_cast<float4*>(db));
  return check_launch("tail_rows_scatter");
}

extern "C" int pdae_slab_sum_epi(int S, int M, int N, const float* slabs, const float* bias, int epi, const float* Z, float* Y,
                                 pdae_stream_t stream) {
  using namespace pdae;
  if (S <= 0 || S > 8 || M < 0 || N <= 0 || N % 4 != 0) return bad_arg("slab_sum_epi: 1..8 slabs, M >= 0, N a positive multiple of 4");
  if (epi != 0 && epi != 1 && epi != 4) return bad_arg("slab_sum_epi: epi 0 (store), 1 (ReLU) or 4 (mask by Z > 0)");
  if (M == 0) return PDAE_OK;
  if (!slabs || !Y || (epi == 4 && !Z)) return bad_arg("slab_sum_epi: null pointer");
  const long long n4 = (long long)M * (N / 4);
  hipLaunchKernelGGL(slab_sum_epi_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), n4, N / 4, S,
                     reinterpret_cast<const float4*>(slabs), reinterpret_cast<const float4*>(bias), epi,
                     reinterpret_cast<const float4*>(Z), reinterpret_cast<float4*>(Y));
  return check_launch("slab_sum_epi");
}

extern "C" int pdae_drop_path_keep(int sites, int B, const float* r, const float* keep, float* out, pdae_stream_t stream) {
  using namespace pdae;
  if (sites < 0 || B < 0) return bad_arg("drop_path_keep: negative size");
  if (sites == 0 || B == 0) return PDAE_OK;
  if (!r || !keep || !out) return bad_arg("drop_path_keep: null pointer");
  const int n = sites * B;
  hipLaunchKernelGGL(drop_path_keep_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), n, B, r, keep, out);
  return check_launch("drop_path_keep");
}

extern "C" int pdae_pos_embed_fc1(int M, int H, const float* xyz, const int64_t* rows, const float* w1, const float* b1,
                                  float* h, float* gp, float* xp, pdae_stream_t stream) {
  using namespace pdae;
  if (M < 0 || H <= 0 || H % 4 != 0) return bad_arg("pos_embed_fc1: M >= 0, H a positive multiple of 4 required");
  if (M == 0) return PDAE_OK;
  if (!xyz || !w1 || !b1 || !h || !gp || !xp) return bad_arg("pos_embed_fc1: null pointer");
  const long long n = (long long)M * (H / 4);
  hipLaunchKernelGGL(pos_embed_fc1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), M, H, xyz,
                     reinterpret_cast<const long long*>(rows), w1, b1, h, gp, xp);
  return check_launch("pos_embed_fc1");
}
