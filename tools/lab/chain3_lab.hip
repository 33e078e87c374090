// LAB (round 6): a GEMM -> GEMM seam kept INSIDE one persistent launch with band-local dependencies,
// against the same two products as two launches.  The chain is the MLP pair of an encoder block at the step's size --
//   fc1: H[M, N1] = X[M, K1] . W1[N1, K1]^T        (2944 x 1536 x 384)
//   fc2: Y[s][M, N2] = H[M, slab s of N1] . W2[N2, slab s of N1]^T   (2944 x 384 x 1536 in S = 3 split-K slabs)
// on the library's own tile body (rows3::gemm3_body, 128 x 128 tiles, exact-split bf16, write-through result stores).
// One block per CU walks a static unit list: first fc1 tiles (band-major), then fc2 units (band, slab, column tile).  A
// fc2 unit of (band b, slab s) waits for the four fc1 tiles that wrote columns [512 s, 512 s + 512) of band b: a counter
// per (band, slab) in global memory -- no grid-wide barrier.  Hand-off, the guide's valid form (MI355X_MICROARCH.md,
// "Valid forms"): producer = write-through (sc1) stores, every wave's s_waitcnt vmcnt(0), workgroup barrier, ONE relaxed
// agent-scope atomic add; consumer = ONE lane polls the counter with relaxed agent-scope loads + s_sleep, agent acquire
// fence, s_waitcnt vmcnt(0), workgroup barrier, plain loads.  Counters only grow: launch e waits for 4 (e + 1); the epoch
// lives in global memory and the block that finishes last advances it (graph replays need no memset node).
// Every spin is bounded (give-up flag in sync[2]); stamps[block][unit][0..2] = s_memrealtime at unit start / dependency met /
// published, for the in-kernel timeline (tools/lab/chain3_lab.py).
#include "../../point_dae_amd/csrc/rows3_kernel.h"

using namespace pdae;
using namespace pdae::rows3;

struct ChainArgs {
  rows::Args p1, p2;
  int n1, n2;                    // units of the two phases
  int per_slab;                  // fc1 column tiles per fc2 slab (kchunk2 / 128)
  int waits;                     // 1: the chain; 0: no waits (an upper bound on what the seam could ever cost; wrong results)
  int* cnt;                      // [bands][slabs] arrival counters (monotone)
  int* sync;                     // [0] epoch, [1] blocks done, [2] give-ups
  unsigned long long* stamps;    // [blocks][MAXU][4]
};
constexpr int MAXU = 4;

__device__ __forceinline__ int bx_of_tile(const rows::Args& p, int t) {
  const int chunk = (p.tiles + 7) >> 3;
  return (t % chunk) * 8 + t / chunk;     // gemm3_body: tile = (bx & 7) * chunk + (bx >> 3)
}

__global__ __launch_bounds__(512) void chain3_kernel(const ChainArgs c) {
  constexpr auto E = rows::EPI_STORE;
  __shared__ int s_epoch;
  const int tid = threadIdx.x;
  if (tid == 0) s_epoch = __hip_atomic_load(&c.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int target = c.per_slab * (s_epoch + 1);
  const int gx1 = 8 * ((c.p1.tiles + 7) >> 3), gx2 = 8 * ((c.p2.tiles + 7) >> 3);
  // consecutive units on one XCD (blocks b, b + 8, ... share one): the tiles of a row band re-read it from that XCD's L2,
  // as the library's tile order does
  const int pos = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  int k = 0;
  for (int u = pos; u < c.n1 + c.n2; u += gridDim.x, ++k) {
    unsigned long long* st = c.stamps + ((size_t)blockIdx.x * MAXU + (k < MAXU ? k : MAXU - 1)) * 4;
    if (tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
    if (u < c.n1) {
      if (tid == 0) st[1] = st[0];
      gemm3_body<1, 2, 4, 2, 2, false, E, true, 0, false>(c.p1, bx_of_tile(c.p1, u), 0, 0, gx1);
      // publish: this block's stores (write-through) have left, then ONE arrival for the (band, slab) they belong to
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int band = u / c.p1.tiles_n, j = u % c.p1.tiles_n;
        __hip_atomic_fetch_add(&c.cnt[band * c.p2.slabs + j / c.per_slab], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st[2] = __builtin_amdgcn_s_memrealtime();
        st[3] = (unsigned long long)u;
      }
    } else {
      const int v = u - c.n1;
      const int j = v % c.p2.tiles_n, bs = v / c.p2.tiles_n, s = bs % c.p2.slabs, band = bs / c.p2.slabs;
      if (c.waits) {
        if (tid == 0) {
          int spins = 0;
          while (__hip_atomic_load(&c.cnt[band * c.p2.slabs + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 400000) {             // ~25 ms: give up loudly instead of hanging the box
              __hip_atomic_fetch_add(&c.sync[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      }
      if (tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
      gemm3_body<1, 2, 4, 2, 2, false, E, true, 0, false>(c.p2, bx_of_tile(c.p2, band * c.p2.tiles_n + j), s, 0, gx2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st[2] = __builtin_amdgcn_s_memrealtime(), st[3] = (unsigned long long)u;
    }
  }
  // the last block out advances the epoch (every block has read it by then) and clears the arrival count
  __syncthreads();
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(&c.sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(&c.sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&c.sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static void fill(rows::Args& a, int M, int N, int K, const float* A, const float* B, float* C, int splits) {
  a = {};
  a.M = M, a.N = N, a.K = K, a.A = A, a.lda = K, a.B = B, a.ldb = K, a.C = C, a.ldc = N, a.slab = (long long)M * N;
  a.tiles_n = (N + 127) / 128;
  a.tiles = ((M + 127) / 128) * a.tiles_n;
  a.kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
  a.slabs = splits;
}

// the chain in ONE launch (blocks = 256: one per CU)
extern "C" int lab_chain3(int M, int N1, int K1, int N2, int slabs, const float* X, const float* W1, float* H, const float* W2,
                          float* Y, int* cnt, int* sync, unsigned long long* stamps, int waits, int blocks, void* stream) {
  ChainArgs c = {};
  fill(c.p1, M, N1, K1, X, W1, H, 1);
  fill(c.p2, M, N2, N1, H, W2, Y, slabs);
  if (c.p2.kchunk % 128 || N1 % 128 || N1 != slabs * c.p2.kchunk) return -2;
  c.per_slab = c.p2.kchunk / 128;
  c.n1 = c.p1.tiles, c.n2 = c.p2.tiles * slabs;
  if ((c.n1 + c.n2 + blocks - 1) / blocks > MAXU) return -3;
  c.waits = waits, c.cnt = cnt, c.sync = sync, c.stamps = stamps;
  const size_t lds = 2 * 3 * (size_t)(128 + 128) * 80;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(chain3_kernel, dim3(blocks), dim3(512), lds, (hipStream_t)stream, c);
  return (int)hipGetLastError();
}

// the same two products as two launches of the same tile body (one tile per block: what the library launches)
extern "C" int lab_two_launches(int M, int N1, int K1, int N2, int slabs, const float* X, const float* W1, float* H,
                                const float* W2, float* Y, void* stream) {
  rows::Args p1, p2;
  fill(p1, M, N1, K1, X, W1, H, 1);
  fill(p2, M, N2, N1, H, W2, Y, slabs);
  const size_t lds = 2 * 3 * (size_t)(128 + 128) * 80;
  auto k = gemm3_kernel<1, 2, 4, 2, 2, false, rows::EPI_STORE, true, 0, false>;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(k, dim3(8 * ((p1.tiles + 7) / 8), 1, 1), dim3(512), lds, (hipStream_t)stream, p1);
  hipLaunchKernelGGL(k, dim3(8 * ((p2.tiles + 7) / 8), slabs, 1), dim3(512), lds, (hipStream_t)stream, p2);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The same experiment over the FOUR heavy phases of an encoder block's forward (qkv -> proj -> fc1 -> fc2; LayerNorm and the
// attention core left out: proj reads the q third of qkv's result directly -- the dependency structure and the GEMM work are
// the block's, the arithmetic in between is not) as ONE table-driven persistent launch: the host hands an ordered unit
// list (any topological order: phase-major = "four launches without boundaries", or band-group-major = bands drift apart
// and a band's narrow phases share the chip with other bands' wide ones), every unit = (phase, tile, slab, counter to wait
// on + how many arrivals, counter to bump).  Same hand-off as chain3_kernel.
struct Unit { int phase, tile, slab, wait_idx, need, done_idx; };
struct Chain4Args {
  rows::Args p[4];
  int nunits;
  int waits;
  const Unit* units;
  int* cnt;
  int* sync;
  unsigned long long* stamps;    // [blocks][MAXU4][4]
};
constexpr int MAXU4 = 6;

__global__ __launch_bounds__(512) void chain4_kernel(const Chain4Args c) {
  constexpr auto E = rows::EPI_STORE;
  __shared__ int s_epoch;
  const int tid = threadIdx.x;
  if (tid == 0) s_epoch = __hip_atomic_load(&c.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int epoch1 = s_epoch + 1;
  const int pos = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  int k = 0;
  for (int u = pos; u < c.nunits; u += gridDim.x, ++k) {
    const Unit un = c.units[u];
    unsigned long long* st = c.stamps + ((size_t)blockIdx.x * MAXU4 + (k < MAXU4 ? k : MAXU4 - 1)) * 4;
    if (tid == 0) st[0] = __builtin_amdgcn_s_memrealtime();
    if (un.wait_idx >= 0 && c.waits) {
      if (tid == 0) {
        int spins = 0;
        const int target = un.need * epoch1;
        while (__hip_atomic_load(&c.cnt[un.wait_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > 400000) {
            __hip_atomic_fetch_add(&c.sync[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    if (tid == 0) st[1] = __builtin_amdgcn_s_memrealtime();
    const rows::Args& p = c.p[un.phase];
    gemm3_body<1, 2, 4, 2, 2, false, E, true, 0, false>(p, bx_of_tile(p, un.tile), un.slab, 0, 8 * ((p.tiles + 7) >> 3));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (un.done_idx >= 0) __hip_atomic_fetch_add(&c.cnt[un.done_idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st[2] = __builtin_amdgcn_s_memrealtime();
      st[3] = (unsigned long long)u;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(&c.sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(&c.sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&c.sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static void fill4(rows::Args (&p)[4], int M, const float* X, const float* Wqkv, float* QKV, const float* Wproj, float* P,
                  const float* W1, float* H, const float* W2, float* Y, int slabs) {
  fill(p[0], M, 1152, 384, X, Wqkv, QKV, 1);
  fill(p[1], M, 384, 384, QKV, Wproj, P, 1);
  p[1].lda = 1152;                                  // the q third of the qkv rows
  fill(p[2], M, 1536, 384, P, W1, H, 1);
  fill(p[3], M, 384, 1536, H, W2, Y, slabs);
}

extern "C" int lab_chain4(int M, int slabs, const float* X, const float* Wqkv, float* QKV, const float* Wproj, float* P,
                          const float* W1, float* H, const float* W2, float* Y, const void* units, int nunits, int* cnt, int* sync,
                          unsigned long long* stamps, int waits, int blocks, void* stream) {
  Chain4Args c = {};
  fill4(c.p, M, X, Wqkv, QKV, Wproj, P, W1, H, W2, Y, slabs);
  if ((nunits + blocks - 1) / blocks > MAXU4) return -3;
  c.nunits = nunits, c.waits = waits, c.units = reinterpret_cast<const Unit*>(units), c.cnt = cnt, c.sync = sync, c.stamps = stamps;
  const size_t lds = 2 * 3 * (size_t)(128 + 128) * 80;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(chain4_kernel, dim3(blocks), dim3(512), lds, (hipStream_t)stream, c);
  return (int)hipGetLastError();
}

// the same four products as four launches of the same tile body
extern "C" int lab_four_launches(int M, int slabs, const float* X, const float* Wqkv, float* QKV, const float* Wproj, float* P,
                                 const float* W1, float* H, const float* W2, float* Y, void* stream) {
  rows::Args p[4];
  fill4(p, M, X, Wqkv, QKV, Wproj, P, W1, H, W2, Y, slabs);
  const size_t lds = 2 * 3 * (size_t)(128 + 128) * 80;
  auto k = gemm3_kernel<1, 2, 4, 2, 2, false, rows::EPI_STORE, true, 0, false>;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  for (int q = 0; q < 4; ++q)
    hipLaunchKernelGGL(k, dim3(8 * ((p[q].tiles + 7) / 8), q == 3 ? slabs : 1, 1), dim3(512), lds, (hipStream_t)stream, p[q]);
  return (int)hipGetLastError();
}
