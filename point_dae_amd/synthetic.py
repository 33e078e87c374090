"""Synthetic ShapeNet-shaped clouds (there is no dataset in the image).

The reference pipeline is datasets/ShapeNet55Dataset.py:90-119: an 8192-point
surface sample -> centre + scale to the unit sphere (datasets/corrupt_util.py
:7-17) -> random subset of npoints.  Here the surface is a random union of 3-6
primitives (sphere / box / cylinder) with random pose and scale.
"""
import numpy as np


def _sphere(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _box(rng, n):
    p = rng.uniform(-1, 1, (n, 3))
    face = rng.integers(0, 3, n)
    p[np.arange(n), face] = rng.choice([-1.0, 1.0], n)
    return p


def _cylinder(rng, n):
    t = rng.uniform(0, 2 * np.pi, n)
    return np.stack([np.cos(t), np.sin(t), rng.uniform(-1, 1, n)], 1)


def _rotation(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q


def shapenet_like_clouds(B, N, seed=0, dense=8192):
    """-> (B, N, 3) float32, centred, max norm 1."""
    rng = np.random.default_rng(seed)
    out = np.empty((B, N, 3), np.float32)
    gens = (_sphere, _box, _cylinder)
    for b in range(B):
        parts = rng.integers(3, 7)
        counts = rng.multinomial(dense, np.ones(parts) / parts)
        pts = []
        for c in counts:
            p = gens[rng.integers(0, 3)](rng, int(c))
            p = p * rng.uniform(0.15, 0.6, 3) @ _rotation(rng).T + rng.uniform(-0.5, 0.5, 3)
            pts.append(p)
        pts = np.concatenate(pts, 0)
        pts = pts - pts.mean(0, keepdims=True)
        pts = pts / np.sqrt((pts ** 2).sum(1)).max()
        sel = rng.choice(pts.shape[0], N, replace=N > pts.shape[0])
        out[b] = pts[sel].astype(np.float32)
    return out


def labelled_clouds(count, N, seed=0, classes=3):
    """Labelled stand-in of ModelNet for the SVM probe: every cloud is ONE primitive family
    (label 0 sphere, 1 box, 2 cylinder, ...) under a random anisotropic scale and pose, unit-sphere
    normalised.  -> (clouds (count,N,3) f32, labels (count,) i64)."""
    rng = np.random.default_rng(seed)
    gens = (_sphere, _box, _cylinder)
    out = np.empty((count, N, 3), np.float32)
    labels = rng.integers(0, min(classes, len(gens)), count)
    for b in range(count):
        p = gens[labels[b]](rng, N) * rng.uniform(0.5, 1.0, 3) @ _rotation(rng).T
        p = p - p.mean(0, keepdims=True)
        out[b] = (p / np.sqrt((p ** 2).sum(1)).max()).astype(np.float32)
    return out, labels.astype(np.int64)
