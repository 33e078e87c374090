"""numpy front-end of the CPU oracle (oracle/pdae_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (point_dae_amd/) never
imports this module; it has no CPU path and fails loudly without the HIP
library.

Every function takes/returns C-contiguous numpy arrays and mirrors the
reference operator it restates (see the C file for file:line citations).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpdae_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pdae_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpdae_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_threads(t):
    lib().oracle_set_threads(int(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed with status {rc}")


def opt_n_threads(n):
    return int(lib().oracle_opt_n_threads(int(n)))


def furthest_point_sample(xyz, npoint, return_centres=False):
    """(B,N,3) f32 -> (B,npoint) i32 [, (B,npoint,3) f32]."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    idx = np.zeros((B, npoint), np.int32)
    ctr = np.zeros((B, npoint, 3), np.float32) if return_centres else None
    _check(lib().oracle_furthest_point_sampling(B, N, int(npoint), _p(xyz, _f32p),
                                                _p(idx, _i32p), _p(ctr, _f32p)),
           "fps")
    return (idx, ctr) if return_centres else idx


def gather_operation(features, idx):
    """(B,C,N) f32, (B,m) i32 -> (B,C,m)."""
    features, idx = _f32(features), _i32(idx)
    B, C, N = features.shape
    m = idx.shape[1]
    out = np.zeros((B, C, m), np.float32)
    _check(lib().oracle_gather_points(B, C, N, m, _p(features, _f32p),
                                      _p(idx, _i32p), _p(out, _f32p)), "gather")
    return out


def gather_operation_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, m = grad_out.shape
    g = np.zeros((B, C, N), np.float32)
    _check(lib().oracle_gather_points_grad(B, C, int(N), m, _p(grad_out, _f32p),
                                           _p(idx, _i32p), _p(g, _f32p)),
           "gather_grad")
    return g


def dropout_local(xyz, nclusters, seed_rank, sizes):
    """xyz (B,P,3); nclusters (B,), seed_rank / sizes (B,8) i32 -> alive (B,P) u8."""
    xyz = _f32(xyz)
    B, P, _ = xyz.shape
    nclusters, seed_rank, sizes = _i32(nclusters), _i32(seed_rank), _i32(sizes)
    alive = np.zeros((B, P), np.uint8)
    _check(lib().oracle_dropout_local(B, P, _p(xyz, _f32p), _p(nclusters, _i32p), _p(seed_rank, _i32p),
                                      _p(sizes, _i32p), alive.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
           "dropout_local")
    return alive


def three_nn(unknown, known):
    """unknown (B,n,3), known (B,m,3) -> (dist2 (B,n,3) SQUARED, idx (B,n,3) i32)."""
    unknown, known = _f32(unknown), _f32(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((B, n, 3), np.float32)
    idx = np.zeros((B, n, 3), np.int32)
    _check(lib().oracle_three_nn(B, n, m, _p(unknown, _f32p), _p(known, _f32p), _p(d2, _f32p), _p(idx, _i32p)),
           "three_nn")
    return d2, idx


def three_interpolate(points, idx, weight):
    """points (B,c,m), idx (B,n,3) i32, weight (B,n,3) -> (B,c,n)."""
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    B, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((B, c, n), np.float32)
    _check(lib().oracle_three_interpolate(B, c, m, n, _p(points, _f32p), _p(idx, _i32p), _p(weight, _f32p),
                                          _p(out, _f32p)), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    B, c, n = grad_out.shape
    g = np.zeros((B, c, int(m)), np.float32)
    _check(lib().oracle_three_interpolate_grad(B, c, n, int(m), _p(grad_out, _f32p), _p(idx, _i32p),
                                               _p(weight, _f32p), _p(g, _f32p)), "three_interpolate_grad")
    return g


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,m,3) -> (B,m,nsample) i32."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((B, m, nsample), np.int32)
    _check(lib().oracle_ball_query(B, N, m, ctypes.c_float(radius), int(nsample),
                                   _p(new_xyz, _f32p), _p(xyz, _f32p),
                                   _p(idx, _i32p)), "ball_query")
    return idx


def grouping_operation(features, idx):
    """(B,C,N) f32, (B,np,ns) i32 -> (B,C,np,ns)."""
    features, idx = _f32(features), _i32(idx)
    B, C, N = features.shape
    _, npnt, ns = idx.shape
    out = np.zeros((B, C, npnt, ns), np.float32)
    _check(lib().oracle_group_points(B, C, N, npnt, ns, _p(features, _f32p),
                                     _p(idx, _i32p), _p(out, _f32p)), "group")
    return out


def grouping_operation_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, npnt, ns = grad_out.shape
    g = np.zeros((B, C, N), np.float32)
    _check(lib().oracle_group_points_grad(B, C, int(N), npnt, ns,
                                          _p(grad_out, _f32p), _p(idx, _i32p),
                                          _p(g, _f32p)), "group_grad")
    return g


def knn(ref, query, k, return_nbr=False):
    """ref (B,N,3), query (B,G,3) -> dist (B,G,k) f32, idx (B,G,k) i64
    [, nbr (B,G,k,3) = ref[idx] - query]."""
    ref, query = _f32(ref), _f32(query)
    B, N, _ = ref.shape
    G = query.shape[1]
    idx = np.zeros((B, G, k), np.int64)
    dist = np.zeros((B, G, k), np.float32)
    nbr = np.zeros((B, G, k, 3), np.float32) if return_nbr else None
    _check(lib().oracle_knn(B, N, G, int(k), _p(ref, _f32p), _p(query, _f32p),
                            _p(idx, _i64p), _p(dist, _f32p), _p(nbr, _f32p)),
           "knn")
    return (dist, idx, nbr) if return_nbr else (dist, idx)


def chamfer_forward(xyz1, xyz2):
    """-> dist1 (B,n), dist2 (B,m), idx1, idx2 (i32).  float32 or float64
    inputs (float64 only serves the gradcheck of the reference's test.py)."""
    dt = np.float64 if np.asarray(xyz1).dtype == np.float64 else np.float32
    ptr = _f64p if dt == np.float64 else _f32p
    fn = (lib().oracle_chamfer_forward_f64 if dt == np.float64
          else lib().oracle_chamfer_forward_f32)
    xyz1 = np.ascontiguousarray(xyz1, dtype=dt)
    xyz2 = np.ascontiguousarray(xyz2, dtype=dt)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((B, n), dt)
    d2 = np.zeros((B, m), dt)
    i1 = np.zeros((B, n), np.int32)
    i2 = np.zeros((B, m), np.int32)
    _check(fn(B, n, _p(xyz1, ptr), m, _p(xyz2, ptr), _p(d1, ptr), _p(d2, ptr),
              _p(i1, _i32p), _p(i2, _i32p)), "chamfer_forward")
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    dt = np.float64 if np.asarray(xyz1).dtype == np.float64 else np.float32
    ptr = _f64p if dt == np.float64 else _f32p
    fn = (lib().oracle_chamfer_backward_f64 if dt == np.float64
          else lib().oracle_chamfer_backward_f32)
    xyz1 = np.ascontiguousarray(xyz1, dtype=dt)
    xyz2 = np.ascontiguousarray(xyz2, dtype=dt)
    g1 = np.ascontiguousarray(grad_dist1, dtype=dt)
    g2 = np.ascontiguousarray(grad_dist2, dtype=dt)
    idx1, idx2 = _i32(idx1), _i32(idx2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gx1 = np.zeros((B, n, 3), dt)
    gx2 = np.zeros((B, m, 3), dt)
    _check(fn(B, n, _p(xyz1, ptr), m, _p(xyz2, ptr), _p(idx1, _i32p),
              _p(idx2, _i32p), _p(g1, ptr), _p(g2, ptr), _p(gx1, ptr),
              _p(gx2, ptr)), "chamfer_backward")
    return gx1, gx2


def chamfer_distance_l2(xyz1, xyz2):
    """extensions/chamfer_dist/__init__.py:36-44: mean(d1)+mean(d2)."""
    d1, d2, _, _ = chamfer_forward(xyz1, xyz2)
    return d1.mean(dtype=d1.dtype) + d2.mean(dtype=d2.dtype)


def chamfer_distance_l1(xyz1, xyz2):
    """extensions/chamfer_dist/__init__.py:397-417."""
    d1, d2, _, _ = chamfer_forward(xyz1, xyz2)
    return (np.sqrt(d1).mean(dtype=d1.dtype) + np.sqrt(d2).mean(dtype=d2.dtype)) / 2


def emd_approxmatch(xyz1, xyz2):
    """-> match (B,m,n) f32."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.zeros((B, m, n), np.float32)
    temp = np.zeros((B, (n + m) * 2), np.float32)
    _check(lib().oracle_emd_approxmatch(B, n, m, _p(xyz1, _f32p), _p(xyz2, _f32p),
                                        _p(match, _f32p), _p(temp, _f32p)),
           "emd_approxmatch")
    return match


def emd_matchcost(xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.zeros((B,), np.float32)
    _check(lib().oracle_emd_matchcost(B, n, m, _p(xyz1, _f32p), _p(xyz2, _f32p),
                                      _p(match, _f32p), _p(cost, _f32p)),
           "emd_matchcost")
    return cost


def emd_matchcost_grad(grad_cost, xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    grad_cost = _f32(grad_cost)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros((B, n, 3), np.float32)
    g2 = np.zeros((B, m, 3), np.float32)
    _check(lib().oracle_emd_matchcost_grad(B, n, m, _p(grad_cost, _f32p),
                                           _p(xyz1, _f32p), _p(xyz2, _f32p),
                                           _p(match, _f32p), _p(g1, _f32p),
                                           _p(g2, _f32p)), "emd_matchcost_grad")
    return g1, g2


def earth_mover_distance(xyz1, xyz2):
    """extensions/emd/emd.py:29-49: mean over batch of cost / n."""
    match = emd_approxmatch(xyz1, xyz2)
    cost = emd_matchcost(xyz1, xyz2, match)
    return (cost / np.float32(np.asarray(xyz1).shape[1])).mean(dtype=np.float32)


# ---- Point-M2AE hierarchical grouping -------------------------------------------------
def m2ae_group(xyz, num_group, group_size):
    """Group.forward, models/Point_M2AE_modules.py:227-248: centres by FPS (misc.fps), k nearest by KNN, the flat
    index `idx + b * num_points`, the gathered neighbourhood minus its centre.
    -> neighborhood (B,G,M,3) f32, center (B,G,3) f32, idx (B*G*M,) i64."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    _, center = furthest_point_sample(xyz, num_group, return_centres=True)
    _, idx = knn(xyz, center, group_size)                                   # (B,G,M) i64
    flat = (idx + np.arange(B, dtype=np.int64).reshape(-1, 1, 1) * N).reshape(-1)      # :243-245
    neighborhood = xyz.reshape(B * N, 3)[flat].reshape(B, num_group, group_size, 3)   # :246-247
    neighborhood = neighborhood - center[:, :, None, :]                                # :249
    return np.ascontiguousarray(neighborhood), center, flat


def m2ae_hierarchy(pts, num_groups, group_sizes):
    """Point_M2AE.forward, models/Point_M2AE.py:245-263: level 0 groups the points, level i the centres of i - 1."""
    neighborhoods, centers, idxs = [], [], []
    src = _f32(pts)[:, :, :3]
    for g, k in zip(num_groups, group_sizes):
        nb, c, idx = m2ae_group(src, g, k)
        neighborhoods.append(nb), centers.append(c), idxs.append(idx)
        src = c
    return neighborhoods, centers, idxs


def m2ae_multi_scale_mask(top_masked, idxs, centers):
    """H_Encoder.forward, models/Point_M2AE.py:107-121: `idx_masked = ~mask[:, None] * idx` (masked parents send
    index 0), `ones(b * G_child).scatter(0, idx_masked, 0)`; -> masks finest level first (after :121's reverse)."""
    masks = [np.asarray(top_masked, dtype=bool)]
    for i in range(len(idxs) - 1, 0, -1):
        b, g, _ = centers[i].shape
        idx = idxs[i].reshape(b * g, -1)
        idx_masked = (~masks[-1].reshape(-1))[:, None] * idx
        child = np.ones(b * centers[i - 1].shape[1], dtype=bool)
        child[idx_masked.reshape(-1)] = False
        masks.append(child.reshape(b, centers[i - 1].shape[1]))
    masks.reverse()
    return masks
