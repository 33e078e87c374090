"""CPU tests of the drop-in boundary: libpdae_hip.so loads without a GPU and
exports exactly the entry points include/pdae.h declares; the Python operator
layer refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pdae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdae_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared()
    for must in ("pdae_furthest_point_sampling", "pdae_knn", "pdae_ball_query", "pdae_group_points",
                 "pdae_gather_points", "pdae_chamfer_forward", "pdae_chamfer_backward",
                 "pdae_emd_approxmatch"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from point_dae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), f"{name} declared in pdae.h but not exported"
    assert sorted(_lib.exported_symbols()) == _declared()      # binding covers the whole ABI
    handle.pdae_version.restype = ctypes.c_char_p
    assert handle.pdae_version().startswith(b"pdae-hip gfx950")


def test_no_cpu_fallback():
    from point_dae_amd import chamfer_dist, pointnet2_utils
    from point_dae_amd.knn_cuda import KNN
    x = torch.rand(2, 64, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        pointnet2_utils.furthest_point_sample(x, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        chamfer_dist.ChamferDistanceL2()(x, x)
    with pytest.raises(RuntimeError, match="GPU"):
        KNN(4, transpose_mode=True)(x, x[:, :3])
    from point_dae_amd import nn_ops
    lin = torch.nn.Linear(8, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        nn_ops.mlp_chain(torch.rand(4, 8), [lin, lin])          # the coarse heads: no CPU / framework path either
    with pytest.raises(RuntimeError, match="GPU"):
        nn_ops.linear_any(torch.rand(4, 8), lin.weight, lin.bias)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under point_dae_amd/ may use it."""
    pkg = os.path.join(ROOT, "point_dae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "pdae_oracle" not in src or f.endswith((".hip", ".h", ".cpp")) , f


def test_product_has_no_library_gemm_call_sites():
    """Every dense layer of the product models runs on the hand-written kernels: no F.linear / torch.mm /
    matmul / bmm / addmm call on device tensors in the package (host-side 3x3 composition of the drawn affine
    maps in corrupt_util_tensor.py / datasets.py / synthetic.py is numpy or CPU torch)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'point_dae_amd')
    host_side = {'corrupt_util_tensor.py', 'datasets.py', 'synthetic.py', 'svm_probe.py'}
    pat = re.compile(r'F\.linear\(|torch\.(mm|bmm|addmm|matmul|einsum)\(|\.matmul\(|F\.conv1d\(|F\.conv2d\(')
    hits = []
    for name in sorted(os.listdir(root)):
        if name.endswith('.py') and name not in host_side:
            for i, line in enumerate(open(os.path.join(root, name)), 1):
                code = line.split('#')[0]
                if pat.search(code) or re.search(r'[\w\)\]] @ [\w\(]', code):
                    hits.append('%s:%d %s' % (name, i, line.strip()))
    assert not hits, hits


def test_library_is_built_without_slp_packed_fp32():
    """csrc/Makefile keeps `-fno-slp-vectorize`: hipcc's SLP vectoriser packs scalar fp32 accumulations into
    v_pk_{fma,mul,add}_f32 with operand selects, which gfx950 computes wrongly (low halves, lanes 16-31) while another
    wave of the CU runs bf16 MFMAs beside ds_read_b128 -- another stream or another process (tools/xproc_repro.hip,
    INTEGRATION.md).  __graft_entry__.build() must not override the flags either."""
    with open(os.path.join(ROOT, 'point_dae_amd', 'csrc', 'Makefile')) as f:
        flags = [ln for ln in f if ln.startswith('FLAGS')]
    assert flags and '-fno-slp-vectorize' in flags[0] and '-ffp-contract=off' in flags[0], flags
    with open(os.path.join(ROOT, '__graft_entry__.py')) as f:
        assert 'FLAGS=' not in f.read()
