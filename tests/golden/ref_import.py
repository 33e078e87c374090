"""Import the reference's PYTHON model code in the dev container, with the native
ops it needs replaced by the CPU oracle.

Used only by tests/golden/make_fixtures.py (and the optional CPU test that
re-checks the oracle model against the live reference when /root/reference is
present).  Nothing here travels as reference source: the reference is imported
from where it lies, read-only, with bytecode writing disabled.

Recipe: stub the modules absent from this image
(ipdb, termcolor, easydict, timm.models.layers, knn_cuda, pointnet2_ops,
chamfer), set builtins.__POINTNET2_SETUP__ so the vendored
extensions/pointnet2/pointnet2_utils.py imports without its _ext, and register
`models` / `datasets` as bare namespace packages so their __init__ side-imports
(h5py, torchvision, every model family) are skipped.
"""
import builtins
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("POINT_DAE_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def _np(t):
    return t.detach().cpu().numpy()


def _install_stubs():
    from oracle import ops as O

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("ipdb", set_trace=lambda *a, **k: None)
    mod("termcolor", colored=lambda s, *a, **k: s)

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    mod("easydict", EasyDict=EasyDict)

    # timm 0.4.5 (requirements.txt:11): DropPath / trunc_normal_
    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            r = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
            r.floor_()
            return x.div(keep) * r

    def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
        return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    timm = mod("timm")
    timm.models = mod("timm.models")
    timm.models.layers = mod("timm.models.layers", DropPath=DropPath, trunc_normal_=trunc_normal_)

    # knn_cuda 0.2
    class KNN(torch.nn.Module):
        def __init__(self, k, transpose_mode=False):
            super().__init__()
            self.k, self._t = k, transpose_mode

        def forward(self, ref, query):
            if not self._t:
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            d, i = O.knn(_np(ref), _np(query), self.k)
            d, i = torch.from_numpy(d), torch.from_numpy(i)
            if not self._t:
                d, i = d.transpose(1, 2).contiguous(), i.transpose(1, 2).contiguous()
            return d, i

    mod("knn_cuda", KNN=KNN)

    # pointnet2_ops (third party; same op names as the vendored extension)
    class _Gather(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, idx):
            ctx.save_for_backward(idx)
            ctx.N = features.shape[2]
            return torch.from_numpy(O.gather_operation(_np(features), _np(idx)))

        @staticmethod
        def backward(ctx, g):
            (idx,) = ctx.saved_tensors
            return torch.from_numpy(O.gather_operation_grad(_np(g), _np(idx), ctx.N)), None

    class _Group(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, idx):
            ctx.save_for_backward(idx)
            ctx.N = features.shape[2]
            return torch.from_numpy(O.grouping_operation(_np(features), _np(idx)))

        @staticmethod
        def backward(ctx, g):
            (idx,) = ctx.saved_tensors
            return torch.from_numpy(O.grouping_operation_grad(_np(g), _np(idx), ctx.N)), None

    def furthest_point_sample(xyz, npoint):
        return torch.from_numpy(O.furthest_point_sample(_np(xyz), npoint))

    def ball_query(radius, nsample, xyz, new_xyz):
        return torch.from_numpy(O.ball_query(radius, nsample, _np(xyz), _np(new_xyz)))

    p2u = mod("pointnet2_ops.pointnet2_utils", furthest_point_sample=furthest_point_sample,
              gather_operation=_Gather.apply, ball_query=ball_query,
              grouping_operation=_Group.apply)
    p2 = mod("pointnet2_ops", pointnet2_utils=p2u)

    # the vendored extension's _ext (extensions/pointnet2/pointnet2_utils.py:23)
    class _Ext:
        @staticmethod
        def furthest_point_sampling(xyz, npoint):
            return furthest_point_sample(xyz, npoint)

        @staticmethod
        def gather_points(features, idx):
            return torch.from_numpy(O.gather_operation(_np(features), _np(idx)))

        @staticmethod
        def gather_points_grad(g, idx, N):
            return torch.from_numpy(O.gather_operation_grad(_np(g), _np(idx), N))

        @staticmethod
        def ball_query(new_xyz, xyz, radius, nsample):
            return ball_query(radius, nsample, xyz, new_xyz)

        @staticmethod
        def group_points(features, idx):
            return torch.from_numpy(O.grouping_operation(_np(features), _np(idx)))

        @staticmethod
        def group_points_grad(g, idx, N):
            return torch.from_numpy(O.grouping_operation_grad(_np(g), _np(idx), N))

    pn2 = mod("pointnet2")
    pn2._ext = mod("pointnet2._ext", **{k: getattr(_Ext, k) for k in dir(_Ext) if not k.startswith("_")})
    builtins.__POINTNET2_SETUP__ = True

    # chamfer pybind module (extensions/chamfer_dist/chamfer_cuda.cpp:36-39)
    def ch_forward(xyz1, xyz2):
        return [torch.from_numpy(a) for a in O.chamfer_forward(_np(xyz1), _np(xyz2))]

    def ch_backward(xyz1, xyz2, idx1, idx2, g1, g2):
        return [torch.from_numpy(a) for a in
                O.chamfer_backward(_np(xyz1), _np(xyz2), _np(idx1), _np(idx2), _np(g1), _np(g2))]

    mod("chamfer", forward=ch_forward, backward=ch_backward)
    return p2


_ready = False


def setup():
    """Make `models.PointCAE_transformer` etc. importable from REF."""
    global _ready
    if _ready:
        return
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    sys.dont_write_bytecode = True
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    p2 = _install_stubs()
    for p in (REF, os.path.join(REF, "extensions", "pointnet2")):
        if p not in sys.path:
            sys.path.append(p)
    for pkg in ("models", "datasets"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        sys.modules[pkg] = m
    # third-party pointnet2_ops.pointnet2_modules = the vendored twin
    # (extensions/pointnet2/pointnet2_modules.py), SURVEY.md 8c
    import importlib
    p2.pointnet2_modules = importlib.import_module("pointnet2_modules")
    sys.modules["pointnet2_ops.pointnet2_modules"] = p2.pointnet2_modules
    _ready = True


def cpu_cuda_noop():
    """`.cuda()` on the parameter-less loss modules must be a no-op on CPU."""
    torch.nn.Module.cuda = lambda self, device=None: self


def seed_all(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
