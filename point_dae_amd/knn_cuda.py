"""`knn_cuda.KNN` of the third-party KNN_CUDA 0.2 wheel, on the gfx950 kernel.

Call sites in the reference: models/PointCAE_transformer.py:59,76,
models/PointCAE_pointnetv2.py:32,49, datasets/corrupt_util_tensor.py:591,600.
KNN_CUDA loops over the batch in Python and launches a distance-matrix kernel
plus an insertion-sort kernel per cloud; here the whole batch is one launch.
"""
import torch

from . import _lib


def knn(ref, query, k, with_neighbourhood=False):
    """ref (B,N,3), query (B,G,3) f32 -> dist (B,G,k) f32, idx (B,G,k) i64
    [, nbr (B,G,k,3) = ref[idx] - query].  No gradients (KNN_CUDA has none)."""
    _lib.require(ref, "ref", torch.float32, 3)
    _lib.require(query, "query", torch.float32, 3)
    B, N, C = ref.shape
    if C != 3 or query.shape[2] != 3 or query.shape[0] != B:
        raise RuntimeError("knn expects ref (B,N,3) and query (B,G,3)")
    G = query.shape[1]
    k = int(k)
    idx = torch.empty((B, G, k), dtype=torch.int64, device=ref.device)
    dist = torch.empty((B, G, k), dtype=torch.float32, device=ref.device)
    nbr = (torch.empty((B, G, k, 3), dtype=torch.float32, device=ref.device)
           if with_neighbourhood else None)
    _lib.call("pdae_knn", ref, B, N, G, k, _lib.ptr(ref), _lib.ptr(query), _lib.ptr(idx),
              _lib.ptr(dist), _lib.ptr(nbr))
    return (dist, idx, nbr) if with_neighbourhood else (dist, idx)


class KNN(torch.nn.Module):
    """KNN(k, transpose_mode)(ref, query) -> (dist, idx); transpose_mode=True
    takes (B,N,dim) tensors and returns (B,G,k); False takes (B,dim,N) and
    returns (B,k,G), as the wheel does."""

    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    @torch.no_grad()
    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(
            ref.shape, query.shape)
        if not self._t:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        dist, idx = knn(ref.contiguous().float(), query.contiguous().float(), self.k)
        if not self._t:
            dist, idx = dist.transpose(1, 2).contiguous(), idx.transpose(1, 2).contiguous()
        return dist, idx
