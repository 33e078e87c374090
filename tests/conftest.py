import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_ops():
    from oracle import ops
    ops.build()
    return ops


def make_clouds(seed, B, N, kind="uniform"):
    """Seeded synthetic clouds used on both sides of every parity test."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":          # what the reference's own tests use (torch.rand)
        return rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    if kind == "shapes":
        from point_dae_amd.synthetic import shapenet_like_clouds
        return shapenet_like_clouds(B, N, seed=seed)
    raise ValueError(kind)


@pytest.fixture(autouse=True, scope="session")
def _one_created_stream():
    """GPU tests run on one created stream (point_dae_amd.graph_step.use_created_stream: NULL-stream
    work between hipGraph replays corrupts the replays on this platform)."""
    import torch
    if torch.cuda.is_available():
        from point_dae_amd.graph_step import use_created_stream
        use_created_stream()
    yield


@pytest.fixture(autouse=True)
def _poison_free_memory(request):
    """PDAE_POISON=1: before every GPU test the caching allocator's free blocks are filled with NaN, so a kernel that
    reads a `torch.empty` buffer it (or its producer) never wrote shows up as NaN instead of passing on memory that
    happened to be zero (a fresh process) or stale-but-plausible (a long suite)."""
    if os.environ.get('PDAE_POISON', '0') in ('', '0') or 'gpu' not in request.keywords:
        yield
        return
    import torch
    big = torch.full((1 << 30,), float('nan'), device='cuda')                  # 4 GB: the large-block pool
    small = [torch.full((1 << k,), float('nan'), device='cuda') for k in range(8, 19) for _ in range(48)]
    del big, small
    yield
