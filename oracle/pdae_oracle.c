/*
 * pdae_oracle.c -- CPU restatement of the reference's native operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under point_dae_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / reported baseline.
 *
 * The reference (YBZh/Point-DAE) has no CPU path for any of these operators
 * (every host entry asserts "CPU not supported", e.g.
 * extensions/pointnet2/_ext_src/src/sampling.cpp:84) and its sources are
 * CUDA-only, so they cannot be compiled in this image (no nvcc, THC headers
 * gone from torch): oracle/_ref is therefore not buildable and this file is a
 * line-by-line sequential restatement of the CUDA kernels, simulating their
 * thread layout where the layout decides the result (FPS tie-breaks, EMD
 * tree reductions).
 *
 * Pinning status:
 *   chamfer  - gradient formula pinned by the reference's own gradcheck
 *              (extensions/chamfer_dist/test.py:23-29) re-run in double on
 *              oracle_chamfer_*_f64; forward has no golden vectors upstream.
 *   emd      - pinned by the reference's known-answer test
 *              (extensions/emd/test_emd_loss.py:7-44, optimum 0.71/cloud).
 *   fps, gather, ball_query, group, knn - PARITY UNPINNED: the reference
 *              holds no vectors or tests for them; restated from source
 *              (file:line cited at each function), cross-checked against
 *              brute-force definitions in tests/.  kNN restates the published
 *              algorithm of third-party KNN_CUDA 0.2 (not in the tree).
 *
 * Arithmetic contract: C float semantics exactly as written in the reference
 * source, every operation rounded, no FMA contraction (build with
 * -ffp-contract=off; the HIP kernels are built the same way).  nvcc's default
 * -fmad=true would contract on a real CUDA build; that is a property of that
 * compiler, not of the source, and is not reproduced.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

/* threads used by the per-cloud loops (1 = scalar port) */
void oracle_set_threads(int t) { g_threads = t > 0 ? t : 1; }
int oracle_get_threads(void) { return g_threads; }

#define PARALLEL_CLOUDS _Pragma("omp parallel for schedule(dynamic,1) num_threads(g_threads)")

/* cuda_utils.h:15-21  opt_n_threads(work_size) */
int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

/* ------------------------------------------------------------------ FPS --
 * sampling_gpu.cu:72-176 (kernel), :62-68 (__update), sampling.cpp:67-88
 * (idxs zero-init, temp = 1e10).  One "block" of block_size threads per cloud
 * is simulated: per-thread strided scan (:98-113), then the shared-memory
 * tree (:118-171).                                                          */
static void fps_one(int n, int m, const float* dataset, float* temp,
                    int32_t* idxs, int block_size, float* dists, int* dists_i) {
  if (m <= 0) return;
  for (int k = 0; k < n; ++k) temp[k] = 1e10f;
  for (int j = 0; j < m; ++j) idxs[j] = 0;
  int old = 0;
  idxs[0] = old;
  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[old * 3 + 0];
    const float y1 = dataset[old * 3 + 1];
    const float z1 = dataset[old * 3 + 2];
    for (int tid = 0; tid < block_size; ++tid) {
      int besti = 0;
      float best = -1;
      for (int k = tid; k < n; k += block_size) {
        const float x2 = dataset[k * 3 + 0];
        const float y2 = dataset[k * 3 + 1];
        const float z2 = dataset[k * 3 + 2];
        const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
        if (mag <= 1e-3) continue; /* float vs double literal, as written */
        const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                        (z2 - z1) * (z2 - z1);
        const float d2 = fminf(d, temp[k]);
        temp[k] = d2;
        besti = d2 > best ? k : besti;
        best = d2 > best ? d2 : best;
      }
      dists[tid] = best;
      dists_i[tid] = besti;
    }
    for (int s = block_size / 2; s >= 1; s >>= 1) {
      for (int tid = 0; tid < s; ++tid) {
        const float v1 = dists[tid], v2 = dists[tid + s];
        const int i1 = dists_i[tid], i2 = dists_i[tid + s];
        dists[tid] = fmaxf(v1, v2);
        dists_i[tid] = v2 > v1 ? i2 : i1;
      }
    }
    old = dists_i[0];
    idxs[j] = old;
  }
}

int oracle_furthest_point_sampling(int b, int n, int m, const float* dataset,
                                   int32_t* idxs, float* centres) {
  if (b < 0 || n <= 0 || m < 0) return -1;
  const int bs = oracle_opt_n_threads(n);
  PARALLEL_CLOUDS
  for (int i = 0; i < b; ++i) {
    float* temp = (float*)malloc(sizeof(float) * (size_t)n);
    float* dists = (float*)malloc(sizeof(float) * (size_t)bs);
    int* dists_i = (int*)malloc(sizeof(int) * (size_t)bs);
    fps_one(n, m, dataset + (size_t)i * n * 3, temp, idxs + (size_t)i * m, bs,
            dists, dists_i);
    if (centres) {
      /* utils/misc.py:19: gather_operation(data^T, idx)^T */
      for (int j = 0; j < m; ++j) {
        const int a = idxs[(size_t)i * m + j];
        for (int c = 0; c < 3; ++c)
          centres[((size_t)i * m + j) * 3 + c] =
              dataset[((size_t)i * n + a) * 3 + c];
      }
    }
    free(temp);
    free(dists);
    free(dists_i);
  }
  return 0;
}

/* --------------------------------------------------------------- gather --
 * sampling_gpu.cu:11-23 / :37-50                                            */
int oracle_gather_points(int b, int c, int n, int npoints, const float* points,
                         const int32_t* idx, float* out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j) {
        const int a = idx[(size_t)i * npoints + j];
        out[((size_t)i * c + l) * npoints + j] =
            points[((size_t)i * c + l) * n + a];
      }
  return 0;
}

int oracle_gather_points_grad(int b, int c, int n, int npoints,
                              const float* grad_out, const int32_t* idx,
                              float* grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j) {
        const int a = idx[(size_t)i * npoints + j];
        grad_points[((size_t)i * c + l) * n + a] +=
            grad_out[((size_t)i * c + l) * npoints + j];
      }
  return 0;
}

/* -------------------------------------------------------- dropout_local --
 * datasets/corrupt_util.py:590-612 with the random draws as inputs: per cluster
 * the seed is the rank-th surviving point in index order (the reference takes
 * the first point of a fresh shuffle), then the K survivors nearest to it are
 * cut off (the reference argsorts all distances descending and truncates);
 * ties by lower index.                                                        */
int oracle_dropout_local(int b, int p, const float* xyz_all, const int32_t* nclusters,
                         const int32_t* seed_rank, const int32_t* sizes, unsigned char* alive_all) {
  if (b < 0 || p < 0) return -1;
  float* d = (float*)malloc(sizeof(float) * (size_t)(p > 0 ? p : 1));
  for (int bi = 0; bi < b; ++bi) {
    const float* xyz = xyz_all + (size_t)bi * p * 3;
    unsigned char* alive = alive_all + (size_t)bi * p;
    memset(alive, 1, (size_t)p);
    const int nc = nclusters[bi] < 8 ? nclusters[bi] : 8;
    for (int c = 0; c < nc; ++c) {
      int K = sizes[bi * 8 + c], r = seed_rank[bi * 8 + c], s = -1;
      for (int k = 0; k < p && s < 0; ++k)
        if (alive[k] && r-- == 0) s = k;
      if (s < 0) { free(d); return -2; }
      for (int k = 0; k < p; ++k) {
        const float dx = xyz[k * 3] - xyz[s * 3], dy = xyz[k * 3 + 1] - xyz[s * 3 + 1],
                    dz = xyz[k * 3 + 2] - xyz[s * 3 + 2];
        d[k] = dx * dx + dy * dy + dz * dz;
      }
      for (; K > 0; --K) {                         /* K times: drop the nearest survivor, lowest index first */
        int best = -1;
        for (int k = 0; k < p; ++k)
          if (alive[k] && (best < 0 || d[k] < d[best])) best = k;
        if (best < 0) break;
        alive[best] = 0;
      }
    }
  }
  free(d);
  return 0;
}

/* ------------------------------------------------- three_nn / interpolate --
 * interpolate_gpu.cu:12-62 (three_nn: double running bests initialised to 1e40,
 * float distance compared with strict <), :66-97 (interpolate), :107-137 (grad:
 * atomicAdd scatter; here in ascending j)                                     */
int oracle_three_nn(int b, int n, int m, const float* unknown_all, const float* known_all,
                    float* dist2_all, int32_t* idx_all) {
  if (b < 0 || n < 0 || m < 0) return -1;
  PARALLEL_CLOUDS
  for (int bi = 0; bi < b; ++bi) {
    const float* unknown = unknown_all + (size_t)bi * n * 3;
    const float* known = known_all + (size_t)bi * m * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = unknown[j * 3 + 0], uy = unknown[j * 3 + 1], uz = unknown[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = known[k * 3 + 0], y = known[k * 3 + 1], z = known[k * 3 + 2];
        const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1; best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2; best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float* d2 = dist2_all + ((size_t)bi * n + j) * 3;
      int32_t* id = idx_all + ((size_t)bi * n + j) * 3;
      d2[0] = (float)best1; d2[1] = (float)best2; d2[2] = (float)best3;
      id[0] = besti1; id[1] = besti2; id[2] = besti3;
    }
  }
  return 0;
}

int oracle_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx,
                             const float* weight, float* out) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float* w = weight + ((size_t)bi * n + j) * 3;
        const int32_t* id = idx + ((size_t)bi * n + j) * 3;
        const float* row = points + ((size_t)bi * c + l) * m;
        out[((size_t)bi * c + l) * n + j] = row[id[0]] * w[0] + row[id[1]] * w[1] + row[id[2]] * w[2];
      }
  return 0;
}

int oracle_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int32_t* idx,
                                  const float* weight, float* grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float* w = weight + ((size_t)bi * n + j) * 3;
        const int32_t* id = idx + ((size_t)bi * n + j) * 3;
        float* row = grad_points + ((size_t)bi * c + l) * m;
        const float v = grad_out[((size_t)bi * c + l) * n + j];
        row[id[0]] += v * w[0];
        row[id[1]] += v * w[1];
        row[id[2]] += v * w[2];
      }
  return 0;
}

/* ----------------------------------------------------------- ball query --
 * ball_query_gpu.cu:12-47; idx zero-init ball_query.cpp:22-24               */
int oracle_ball_query(int b, int n, int m, float radius, int nsample,
                      const float* new_xyz_all, const float* xyz_all,
                      int32_t* idx_all) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return -1;
  memset(idx_all, 0, sizeof(int32_t) * (size_t)b * m * nsample);
  PARALLEL_CLOUDS
  for (int bi = 0; bi < b; ++bi) {
    const float* xyz = xyz_all + (size_t)bi * n * 3;
    const float* new_xyz = new_xyz_all + (size_t)bi * m * 3;
    int32_t* idx = idx_all + (size_t)bi * m * nsample;
    const float radius2 = radius * radius;
    for (int j = 0; j < m; ++j) {
      const float new_x = new_xyz[j * 3 + 0];
      const float new_y = new_xyz[j * 3 + 1];
      const float new_z = new_xyz[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = xyz[k * 3 + 0];
        const float y = xyz[k * 3 + 1];
        const float z = xyz[k * 3 + 2];
        const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) +
                         (new_z - z) * (new_z - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) idx[j * nsample + l] = k;
          idx[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
  return 0;
}

/* ---------------------------------------------------------- group points --
 * group_points_gpu.cu:11-31 / :46-67                                        */
int oracle_group_points(int b, int c, int n, int npoints, int nsample,
                        const float* points, const int32_t* idx, float* out) {
  PARALLEL_CLOUDS
  for (int bi = 0; bi < b; ++bi) {
    const float* p = points + (size_t)bi * n * c;
    const int32_t* id = idx + (size_t)bi * npoints * nsample;
    float* o = out + (size_t)bi * npoints * nsample * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          o[((size_t)l * npoints + j) * nsample + k] =
              p[(size_t)l * n + id[j * nsample + k]];
  }
  return 0;
}

int oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                             const float* grad_out, const int32_t* idx,
                             float* grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  PARALLEL_CLOUDS
  for (int bi = 0; bi < b; ++bi) {
    const float* go = grad_out + (size_t)bi * npoints * nsample * c;
    const int32_t* id = idx + (size_t)bi * npoints * nsample;
    float* gp = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          gp[(size_t)l * n + id[j * nsample + k]] +=
              go[((size_t)l * npoints + j) * nsample + k];
  }
  return 0;
}

/* ------------------------------------------------------------------ kNN --
 * Third-party KNN_CUDA 0.2 (unlimblue/KNN_CUDA, wheel pinned in the
 * reference README.md:45; call sites models/PointCAE_transformer.py:59,76).
 * Published algorithm (Garcia et al. kNN-CUDA, which KNN_CUDA wraps):
 *   1. compute_distances: ssd = sum over dims, in order, of (ref-query)^2
 *   2. modified_insertion_sort per query: keep the k smallest in ascending
 *      order; a candidate equal to the current k-th is skipped (>=), the
 *      shift loop uses strict '>' so equal distances keep the earlier index
 *      first
 *   3. sqrt of the k kept distances; indices returned 0-based as int64
 *      (KNN_CUDA's python subtracts the kernel's 1-based offset) after the
 *      transpose_mode=True transposes, i.e. (b, g, k).
 * In-tree structural reference for a kNN kernel:
 * extensions/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-108.         */
int oracle_knn(int b, int n, int g, int k, const float* ref_all,
               const float* query_all, int64_t* idx_all, float* dist_all,
               float* nbr_all) {
  if (b < 0 || n <= 0 || g < 0 || k <= 0 || k > n) return -1;
  PARALLEL_CLOUDS
  for (int bi = 0; bi < b; ++bi) {
    const float* ref = ref_all + (size_t)bi * n * 3;
    const float* query = query_all + (size_t)bi * g * 3;
    float* col = (float*)malloc(sizeof(float) * (size_t)n);
    float* kd = (float*)malloc(sizeof(float) * (size_t)k);
    int* ki = (int*)malloc(sizeof(int) * (size_t)k);
    for (int q = 0; q < g; ++q) {
      for (int i = 0; i < n; ++i) {
        float ssd = 0.f;
        for (int d = 0; d < 3; ++d) {
          const float tmp = ref[i * 3 + d] - query[q * 3 + d];
          ssd += tmp * tmp;
        }
        col[i] = ssd;
      }
      kd[0] = col[0];
      ki[0] = 0;
      for (int i = 1; i < n; ++i) {
        const float curr = col[i];
        if (i >= k && curr >= kd[k - 1]) continue;
        int j = i < k - 1 ? i : k - 1;
        while (j > 0 && kd[j - 1] > curr) {
          kd[j] = kd[j - 1];
          ki[j] = ki[j - 1];
          --j;
        }
        kd[j] = curr;
        ki[j] = i;
      }
      for (int j = 0; j < k; ++j) {
        const size_t o = ((size_t)bi * g + q) * k + j;
        idx_all[o] = ki[j];
        if (dist_all) dist_all[o] = sqrtf(kd[j]);
        if (nbr_all) /* Group.forward, PointCAE_transformer.py:79-85 */
          for (int d = 0; d < 3; ++d)
            nbr_all[o * 3 + d] = ref[ki[j] * 3 + d] - query[q * 3 + d];
      }
    }
    free(col);
    free(kd);
    free(ki);
  }
  return 0;
}

/* -------------------------------------------------------------- Chamfer --
 * chamfer.cu:15-145 (forward, 512-point chunks of xyz2, strict '<' inside a
 * chunk with the chunk's first point as initial value, strict '>' across
 * chunks), :173-201 (backward), host :147-171 / :203-229.  Generated for
 * float (the contract) and double (only to re-run the reference's gradcheck,
 * extensions/chamfer_dist/test.py:23-29, which needs a double build).      */
#define DEFINE_CHAMFER(T, SFX)                                                 \
  static void chamfer_dir_##SFX(int n, const T* xyz1, int m, const T* xyz2,    \
                                T* dist, int32_t* indexes) {                   \
    const int batch = 512;                                                     \
    for (int j = 0; j < n; ++j) {                                              \
      dist[j] = 0;                                                             \
      indexes[j] = 0;                                                          \
    }                                                                          \
    for (int k2 = 0; k2 < m; k2 += batch) {                                    \
      const int end_k = (m < k2 + batch ? m : k2 + batch) - k2;                \
      const T* buf = xyz2 + (size_t)k2 * 3;                                    \
      for (int j = 0; j < n; ++j) {                                            \
        const T x1 = xyz1[j * 3 + 0], y1 = xyz1[j * 3 + 1],                    \
                z1 = xyz1[j * 3 + 2];                                          \
        T best_dist = 0;                                                       \
        int best_dist_index = 0;                                               \
        for (int k = 0; k < end_k; ++k) {                                      \
          const T x2 = buf[k * 3 + 0] - x1;                                    \
          const T y2 = buf[k * 3 + 1] - y1;                                    \
          const T z2 = buf[k * 3 + 2] - z1;                                    \
          const T d = x2 * x2 + y2 * y2 + z2 * z2;                             \
          if (k == 0 || d < best_dist) {                                       \
            best_dist = d;                                                     \
            best_dist_index = k + k2;                                          \
          }                                                                    \
        }                                                                      \
        if (k2 == 0 || dist[j] > best_dist) {                                  \
          dist[j] = best_dist;                                                 \
          indexes[j] = best_dist_index;                                        \
        }                                                                      \
      }                                                                        \
    }                                                                          \
  }                                                                            \
  int oracle_chamfer_forward_##SFX(int b, int n, const T* xyz1, int m,         \
                                   const T* xyz2, T* dist1, T* dist2,          \
                                   int32_t* idx1, int32_t* idx2) {             \
    if (b < 0 || n < 0 || m < 0) return -1;                                    \
    PARALLEL_CLOUDS                                                            \
    for (int i = 0; i < b; ++i) {                                              \
      chamfer_dir_##SFX(n, xyz1 + (size_t)i * n * 3, m,                        \
                        xyz2 + (size_t)i * m * 3, dist1 + (size_t)i * n,       \
                        idx1 + (size_t)i * n);                                 \
      chamfer_dir_##SFX(m, xyz2 + (size_t)i * m * 3, n,                        \
                        xyz1 + (size_t)i * n * 3, dist2 + (size_t)i * m,       \
                        idx2 + (size_t)i * m);                                 \
    }                                                                          \
    return 0;                                                                  \
  }                                                                            \
  /* one launch of chamfer_dist_grad_kernel; the reference's atomicAdd order   \
   * is unspecified, the oracle fixes it to ascending j */                     \
  static void chamfer_grad_dir_##SFX(int n, const T* xyz1, int m,              \
                                     const T* xyz2, const T* grad_dist1,       \
                                     const int32_t* idx1, T* grad_xyz1,        \
                                     T* grad_xyz2) {                           \
    (void)m;                                                                   \
    for (int j = 0; j < n; ++j) {                                              \
      const T x1 = xyz1[j * 3 + 0], y1 = xyz1[j * 3 + 1],                      \
              z1 = xyz1[j * 3 + 2];                                            \
      const int j2 = idx1[j];                                                  \
      const T x2 = xyz2[j2 * 3 + 0], y2 = xyz2[j2 * 3 + 1],                    \
              z2 = xyz2[j2 * 3 + 2];                                           \
      const T g = grad_dist1[j] * 2;                                           \
      grad_xyz1[j * 3 + 0] += g * (x1 - x2);                                   \
      grad_xyz1[j * 3 + 1] += g * (y1 - y2);                                   \
      grad_xyz1[j * 3 + 2] += g * (z1 - z2);                                   \
      grad_xyz2[j2 * 3 + 0] += -(g * (x1 - x2));                               \
      grad_xyz2[j2 * 3 + 1] += -(g * (y1 - y2));                               \
      grad_xyz2[j2 * 3 + 2] += -(g * (z1 - z2));                               \
    }                                                                          \
  }                                                                            \
  int oracle_chamfer_backward_##SFX(                                           \
      int b, int n, const T* xyz1, int m, const T* xyz2, const int32_t* idx1,  \
      const int32_t* idx2, const T* grad_dist1, const T* grad_dist2,           \
      T* grad_xyz1, T* grad_xyz2) {                                            \
    if (b < 0 || n < 0 || m < 0) return -1;                                    \
    memset(grad_xyz1, 0, sizeof(T) * (size_t)b * n * 3);                       \
    memset(grad_xyz2, 0, sizeof(T) * (size_t)b * m * 3);                       \
    PARALLEL_CLOUDS                                                            \
    for (int i = 0; i < b; ++i) {                                              \
      chamfer_grad_dir_##SFX(n, xyz1 + (size_t)i * n * 3, m,                   \
                             xyz2 + (size_t)i * m * 3,                         \
                             grad_dist1 + (size_t)i * n, idx1 + (size_t)i * n, \
                             grad_xyz1 + (size_t)i * n * 3,                    \
                             grad_xyz2 + (size_t)i * m * 3);                   \
      chamfer_grad_dir_##SFX(m, xyz2 + (size_t)i * m * 3, n,                   \
                             xyz1 + (size_t)i * n * 3,                         \
                             grad_dist2 + (size_t)i * m, idx2 + (size_t)i * m, \
                             grad_xyz2 + (size_t)i * m * 3,                    \
                             grad_xyz1 + (size_t)i * n * 3);                   \
    }                                                                          \
    return 0;                                                                  \
  }

DEFINE_CHAMFER(float, f32)
DEFINE_CHAMFER(double, f64)

/* ------------------------------------------------------------------ EMD --
 * emd_kernel.cu:25-158 approxmatch <<<32,512>>>: per cloud, 10 levels
 * level = -4^j (j = 7..-1), last level 0; three phases per level.  Per-thread
 * accumulation orders are kept (suml/sumr run over the other cloud in
 * ascending index).  __expf is the CUDA fast-math exp; the oracle uses expf,
 * so EMD parity is a floating-point-tolerance statement.                    */
static void approxmatch_one(int n, int m, const float* xyz1, const float* xyz2,
                            float* match, float* temp) {
  float* remainL = temp;
  float* remainR = temp + n;
  float* ratioL = temp + n + m;
  float* ratioR = temp + n + m + n;
  float multiL, multiR;
  if (n >= m) {
    multiL = 1;
    multiR = (float)(n / m);
  } else {
    multiL = (float)(m / n);
    multiR = 1;
  }
  for (size_t j = 0; j < (size_t)n * m; ++j) match[j] = 0;
  for (int j = 0; j < n; ++j) remainL[j] = multiL;
  for (int j = 0; j < m; ++j) remainR[j] = multiR;
  for (int j = 7; j >= -2; j--) {
    float level = -powf(4.0f, (float)j);
    if (j == -2) level = 0;
    for (int k = 0; k < n; ++k) {
      const float x1 = xyz1[k * 3 + 0], y1 = xyz1[k * 3 + 1],
                  z1 = xyz1[k * 3 + 2];
      float suml = 1e-9f;
      for (int l = 0; l < m; ++l) {
        const float x2 = xyz2[l * 3 + 0], y2 = xyz2[l * 3 + 1],
                    z2 = xyz2[l * 3 + 2];
        const float d = level * ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                                 (z2 - z1) * (z2 - z1));
        const float w = expf(d) * remainR[l];
        suml += w;
      }
      ratioL[k] = remainL[k] / suml;
    }
    for (int l = 0; l < m; ++l) {
      const float x2 = xyz2[l * 3 + 0], y2 = xyz2[l * 3 + 1],
                  z2 = xyz2[l * 3 + 2];
      float sumr = 0;
      for (int k = 0; k < n; ++k) {
        const float x1 = xyz1[k * 3 + 0], y1 = xyz1[k * 3 + 1],
                    z1 = xyz1[k * 3 + 2];
        const float w =
            expf(level * ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                          (z2 - z1) * (z2 - z1))) *
            ratioL[k];
        sumr += w;
      }
      sumr *= remainR[l];
      const float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
      ratioR[l] = consumption * remainR[l];
      remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
    }
    for (int k = 0; k < n; ++k) {
      const float x1 = xyz1[k * 3 + 0], y1 = xyz1[k * 3 + 1],
                  z1 = xyz1[k * 3 + 2];
      float suml = 0;
      const float rl = ratioL[k];
      for (int l = 0; l < m; ++l) {
        const float x2 = xyz2[l * 3 + 0], y2 = xyz2[l * 3 + 1],
                    z2 = xyz2[l * 3 + 2];
        const float w =
            expf(level * ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                          (z2 - z1) * (z2 - z1))) *
            rl * ratioR[l];
        match[(size_t)l * n + k] += w;
        suml += w;
      }
      remainL[k] = fmaxf(0.0f, remainL[k] - suml);
    }
  }
}

int oracle_emd_approxmatch(int b, int n, int m, const float* xyz1,
                           const float* xyz2, float* match, float* temp) {
  if (b < 0 || n <= 0 || m <= 0) return -1;
  PARALLEL_CLOUDS
  for (int i = 0; i < b; ++i)
    approxmatch_one(n, m, xyz1 + (size_t)i * n * 3, xyz2 + (size_t)i * m * 3,
                    match + (size_t)i * n * m, temp + (size_t)i * (n + m) * 2);
  return 0;
}

/* emd_kernel.cu:200-243 matchcost <<<32,512>>>: 512 strided per-thread
 * partial sums, then the (threadIdx & j)==0 pairwise tree (:230-235).       */
int oracle_emd_matchcost(int b, int n, int m, const float* xyz1_all,
                         const float* xyz2_all, const float* match_all,
                         float* out) {
  if (b < 0 || n <= 0 || m <= 0) return -1;
  const int T = 512;
  PARALLEL_CLOUDS
  for (int i = 0; i < b; ++i) {
    const float* xyz1 = xyz1_all + (size_t)i * n * 3;
    const float* xyz2 = xyz2_all + (size_t)i * m * 3;
    const float* match = match_all + (size_t)i * n * m;
    float allsum[512];
    for (int t = 0; t < T; ++t) {
      float subsum = 0;
      for (int k = t; k < n; k += T) {
        const float x1 = xyz1[k * 3 + 0], y1 = xyz1[k * 3 + 1],
                    z1 = xyz1[k * 3 + 2];
        for (int l = 0; l < m; ++l) {
          const float x2 = xyz2[l * 3 + 0], y2 = xyz2[l * 3 + 1],
                      z2 = xyz2[l * 3 + 2];
          const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                          (z2 - z1) * (z2 - z1);
          subsum += d * match[(size_t)l * n + k];
        }
      }
      allsum[t] = subsum;
    }
    for (int j = 1; j < T; j <<= 1)
      for (int t = 0; t < T; ++t)
        if ((t & j) == 0 && t + j < T && (t & (j - 1)) == 0)
          allsum[t] += allsum[t + j];
    out[i] = allsum[0];
  }
  return 0;
}

/* emd_kernel.cu:333-355 matchcostgrad1 <<<32,512>>> and :286-327
 * matchcostgrad2 <<<dim3(32,32),256>>> (256 strided partials + tree).       */
int oracle_emd_matchcost_grad(int b, int n, int m, const float* grad_cost,
                              const float* xyz1_all, const float* xyz2_all,
                              const float* match_all, float* grad1_all,
                              float* grad2_all) {
  if (b < 0 || n <= 0 || m <= 0) return -1;
  const int T = 256;
  PARALLEL_CLOUDS
  for (int i = 0; i < b; ++i) {
    const float* xyz1 = xyz1_all + (size_t)i * n * 3;
    const float* xyz2 = xyz2_all + (size_t)i * m * 3;
    const float* match = match_all + (size_t)i * n * m;
    float* grad1 = grad1_all + (size_t)i * n * 3;
    float* grad2 = grad2_all + (size_t)i * m * 3;
    for (int l = 0; l < n; ++l) {
      const float x1 = xyz1[l * 3 + 0], y1 = xyz1[l * 3 + 1],
                  z1 = xyz1[l * 3 + 2];
      float dx = 0, dy = 0, dz = 0;
      for (int k = 0; k < m; ++k) {
        const float x2 = xyz2[k * 3 + 0], y2 = xyz2[k * 3 + 1],
                    z2 = xyz2[k * 3 + 2];
        const float d = match[(size_t)k * n + l] * 2;
        dx += (x1 - x2) * d;
        dy += (y1 - y2) * d;
        dz += (z1 - z2) * d;
      }
      grad1[l * 3 + 0] = dx * grad_cost[i];
      grad1[l * 3 + 1] = dy * grad_cost[i];
      grad1[l * 3 + 2] = dz * grad_cost[i];
    }
    float sum_grad[256 * 3];
    for (int k = 0; k < m; ++k) {
      const float x2 = xyz2[k * 3 + 0], y2 = xyz2[k * 3 + 1],
                  z2 = xyz2[k * 3 + 2];
      for (int t = 0; t < T; ++t) {
        float sx = 0, sy = 0, sz = 0;
        for (int j = t; j < n; j += T) {
          const float x1 = x2 - xyz1[j * 3 + 0];
          const float y1 = y2 - xyz1[j * 3 + 1];
          const float z1 = z2 - xyz1[j * 3 + 2];
          const float d = match[(size_t)k * n + j] * 2;
          sx += x1 * d;
          sy += y1 * d;
          sz += z1 * d;
        }
        sum_grad[t * 3 + 0] = sx;
        sum_grad[t * 3 + 1] = sy;
        sum_grad[t * 3 + 2] = sz;
      }
      for (int j = 1; j < T; j <<= 1)
        for (int t = 0; t < T; ++t)
          if ((t & j) == 0 && t + j < T && (t & (j - 1)) == 0) {
            sum_grad[t * 3 + 0] += sum_grad[(t + j) * 3 + 0];
            sum_grad[t * 3 + 1] += sum_grad[(t + j) * 3 + 1];
            sum_grad[t * 3 + 2] += sum_grad[(t + j) * 3 + 2];
          }
      grad2[k * 3 + 0] = sum_grad[0] * grad_cost[i];
      grad2[k * 3 + 1] = sum_grad[1] * grad_cost[i];
      grad2[k * 3 + 2] = sum_grad[2] * grad_cost[i];
    }
  }
  return 0;
}
