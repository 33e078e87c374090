"""ctypes binding of libpdae_hip.so -- the C ABI declared in include/pdae.h.

This is the only place the product talks to native code.  There is no CPU
fallback: if the library is missing, or a tensor is not a contiguous float32
(int32 / int64 where the ABI says so) tensor on a HIP device, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDAE_LIB") or os.path.join(_HERE, "libpdae_hip.so")     # PDAE_LIB: another build of the library (A/B runs, tools/lab/ab.sh)

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> argtypes (restype is always int unless listed in _STR)
_SIGNATURES = {
    "pdae_furthest_point_sampling": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_gather_points": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_gather_points_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_ball_query": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp],
    "pdae_group_points": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_group_points_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_dropout_local": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_flatten_group_index": [_i, _i, _i, _vp, _vp, _vp],
    "pdae_pipeline_norm_affine": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_pipeline_add_global": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_pipeline_add_local": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_pipeline_density": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_pipeline_subset": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_mask_propagate": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_three_nn": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_three_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_three_interpolate_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_knn": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_patch_affine": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_chamfer_forward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_chamfer_backward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_chamfer_backward_mean": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_mean_sum2": [ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp],
    "pdae_linear_forward": [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "pdae_linear_backward_data": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_linear_backward_weight": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_rows_gemm": [_i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "pdae_rows_gemm_batched": [_i, _i, _i, _i, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp],
    "pdae_rows_wgrad": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_rows_wgrad_multi": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_rows_wgrad_listed": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_sa_group_rows": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_sa_group_rows_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_embed_conv_store_groupmax": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_embed_conv_groupbias_stats": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_embed_bnrelu_conv_groupmax": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_bnrelu_linear_backward_weight": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_embed_bnrelu_conv_store_groupmax": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_group_max_scatter": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_group_scatter_add": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_bnrelu_backward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "pdae_attention_forward": [_i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "pdae_attention_backward": [_i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_add_layernorm_forward": [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp],
    "pdae_layernorm_backward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "pdae_bias_gelu_forward": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_bias_gelu_backward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pdae_gelu_forward": [ctypes.c_longlong, _vp, _vp, _vp],
    "pdae_gelu_backward": [ctypes.c_longlong, _vp, _vp, _vp, _vp],
    "pdae_scale_residual": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_colsum": [_i, _i, _vp, _vp, _i, _vp],
    "pdae_drop_path_keep": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_tail_rows_gather": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_tail_rows_scatter": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_pos_embed_fc1": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_slab_sum_epi": [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "pdae_residual_layernorm_forward": [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp],
    "pdae_residual_layernorm_backward": [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "pdae_scale_colsum": [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "pdae_embed_conv1_stats": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_embed_conv1_backward_weight": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_bn_finalize": [_i, ctypes.c_longlong, _vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_adamw_step": [ctypes.c_longlong, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _vp],
    "pdae_bnrelu_backward_listed": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "pdae_masked_group_sums": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_group_sum_listed": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_linear_backward_weight_listed": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_group_gemm_scatter": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "pdae_conv_stats": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_bnrelu_group_max": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_group_max_scatter_n": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_pool_bn_backward": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_fold_input": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_fold_input_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_fold_input_rows": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_fold_out_backward": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_deferred_flush": [_vp],
    "pdae_emd_approxmatch": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_emd_matchcost": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_emd_matchcost_grad": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_rows_sqnorm": [_i, _i, _vp, _vp, _vp],
    "pdae_gram_topk": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_xyz_topk": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_knn_reverse": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_edge_gather_stats": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_bn_lrelu_rows": [ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pdae_bn_lrelu_backward_reduce": [ctypes.c_longlong, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_edge_backward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_cloud_pool_stats": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_rows_pad": [ctypes.c_longlong, _i, _i, _vp, _vp, _vp],
    "pdae_edge_weight_stack": [_i, _i, _i, _vp, _vp, _vp],
    "pdae_edge_weight_stack_multi": [_i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_edge_weight_unstack_multi": [_i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_edge_weight_unstack": [_i, _i, _i, _vp, _vp, _vp],
    "pdae_cloud_pool_backward": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_rows_gemm_bnrelu_stats": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_bnrelu_backward_apply": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "pdae_bnrelu_backward_listed_apply": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "pdae_max_plus_mean": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_max_plus_mean_grad": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_hcat": [_i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_pad2d": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "pdae_partials_sum_t": [_i, _i, _i, _vp, _vp, _vp],
    "pdae_multi_copy": [_i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_assemble_tokens": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_assemble_tokens_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pdae_embed_split_conv3_weight": [_i, _i, _vp, _vp, _vp, _vp, _vp],
    "pdae_embed_masked_prep": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_embed_dw3_assemble": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pdae_calib_mfma_bf16": [_i, _i, _vp, _vp, _vp, _vp],
    "pdae_calib_copy": [ctypes.c_longlong, _vp, _vp, _vp],
}
# host-side queries (no stream argument)
_HOST = {
    "pdae_rows_gemm_plan": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "pdae_rows_wgrad_workspace": [_i, _i, _vp, _vp, _vp],
    "pdae_rows_wgrad_multi_workspace": [_i, _vp, _vp, _vp, _vp],
    "pdae_set_deterministic": [_vp, ctypes.c_size_t],
    "pdae_deferred_begin": [_vp, ctypes.c_size_t],
    "pdae_deferred_hold": [_i],
    "pdae_fold_input_grad_parts": [_i, _i],
    "pdae_embed_conv1_backward_weight_parts": [_i],
    "pdae_fold_out_backward_parts": [ctypes.c_longlong],
    "pdae_pool_bn_backward_workspace": [ctypes.c_longlong, _i],
    "pdae_deterministic": [],
    "pdae_set_gemm_arith": [_i],
    "pdae_gemm_arith": [],
    "pdae_edge_parts": [],
    "pdae_ctx_create": [_vp],
    "pdae_ctx_destroy": [_vp],
    "pdae_ctx_set_current": [_vp],
    "pdae_ctx_current": [],
    "pdae_cloud_pool_splits": [_i, _i],
    "pdae_rows_gemm_bnrelu_stats_workspace": [_i, _i],
}
_STR = ("pdae_version", "pdae_last_error")

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def exported_symbols():
    """Every symbol include/pdae.h declares (used by the CPU-side ABI test)."""
    return list(_SIGNATURES) + list(_HOST) + list(_STR)


def lib():
    """Load libpdae_hip.so (built by __graft_entry__.build / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). point_dae_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        for name, argtypes in _HOST.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_longlong if name.endswith('_workspace') and 'wgrad' not in name else
                          ctypes.c_void_p if name == 'pdae_ctx_current' else ctypes.c_int)
        for name in _STR:
            getattr(handle, name).restype = ctypes.c_char_p
        _lib = handle
    return _lib


_det_ws = None


def set_deterministic(on=True, megabytes=64):
    """Register (or drop) the workspace of include/pdae.h's deterministic mode on the current
    device.  `PDAE_DETERMINISTIC=1` in the environment does this at the first kernel call."""
    global _det_ws
    handle = lib()
    if on:
        _det_ws = torch.empty(megabytes << 20, dtype=torch.uint8, device='cuda')
        _check(handle, 'pdae_set_deterministic', handle.pdae_set_deterministic(_det_ws.data_ptr(), _det_ws.numel()))
    else:
        torch.cuda.synchronize()
        _check(handle, 'pdae_set_deterministic', handle.pdae_set_deterministic(None, 0))
        _det_ws = None


def deterministic():
    return bool(lib().pdae_deterministic())


_def_ws = None


def deferred_begin(megabytes=64):
    """Park the LayerNorm parameter-gradient partials until deferred_flush (include/pdae.h)."""
    global _def_ws
    if _def_ws is None or _def_ws.numel() != megabytes << 20:
        _def_ws = torch.empty(megabytes << 20, dtype=torch.uint8, device='cuda')
    handle = lib()
    _check(handle, 'pdae_deferred_begin', handle.pdae_deferred_begin(_def_ws.data_ptr(), _def_ws.numel()))
    global _def_keep
    _def_keep = []


_def_keep = None      # while reductions are parked: the workspaces their partials live in


def deferred_flush(on):
    global _def_keep
    call('pdae_deferred_flush', on)
    _def_keep = None      # (the flush is enqueued: later allocations reuse the memory behind it in stream order)


_env_checked = False


def _env_mode():
    global _env_checked
    _env_checked = True
    if os.environ.get('PDAE_DETERMINISTIC', '0') not in ('', '0') and _det_ws is None:
        set_deterministic(True, int(os.environ.get('PDAE_DETERMINISTIC_MB', '64')))


def stream_ptr():
    """hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return t.data_ptr() if t is not None else None


def require(t, name, dtype=torch.float32, dim=None):
    """Input validation of the reference's CHECK_CUDA / CHECK_CONTIGUOUS /
    CHECK_IS_FLOAT / CHECK_IS_INT macros (extensions/pointnet2/_ext_src/include/
    utils.h:8-28), raised as RuntimeError like AT_ASSERT does."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on the GPU (CPU not supported)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if dim is not None and t.dim() != dim:
        raise RuntimeError(f"{name} must have {dim} dimensions, got {t.dim()}")
    return t


CALL_HOOK = None      # tools / tests: callable(name, args) seen by every C entry call (e.g. tools/dump_gemm_shapes.py)


def call(name, on, *args):
    """Invoke one C entry on torch's current stream of `on`'s device; raise on a
    non-zero status."""
    handle = lib()
    if CALL_HOOK is not None:
        CALL_HOOK(name, args)
    if not _env_checked:
        _env_mode()
    if on.device.index != torch.cuda.current_device():
        with torch.cuda.device(on.device):
            rc = getattr(handle, name)(*args, stream_ptr())
    else:
        rc = getattr(handle, name)(*args, stream_ptr())
    if rc != 0:
        msg = handle.pdae_last_error().decode()
        raise RuntimeError(f"{name} failed with status {rc}: {msg}")


def _check(handle, name, rc):
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc}: {handle.pdae_last_error().decode()}")


class Context:
    """A library context (include/pdae.h pdae_ctx_*): the deterministic-mode workspace, the parked reductions and the
    GEMM arithmetic of the host thread that makes it current.  `with Context():` works on a private context."""

    def __init__(self):
        h = ctypes.c_void_p()
        _check(lib(), 'pdae_ctx_create', lib().pdae_ctx_create(ctypes.byref(h)))
        self.handle, self._prev, self._keep = h, None, []

    def __enter__(self):
        self._prev = lib().pdae_ctx_current()
        lib().pdae_ctx_set_current(self.handle)
        return self

    def set_deterministic(self, megabytes=64):
        """Deterministic mode for THIS context (it must be current): the workspace belongs to the context."""
        ws = torch.empty(megabytes << 20, dtype=torch.uint8, device='cuda')
        self._keep.append(ws)
        _check(lib(), 'pdae_set_deterministic', lib().pdae_set_deterministic(ws.data_ptr(), ws.numel()))

    def __exit__(self, *exc):
        torch.cuda.current_stream().synchronize()
        lib().pdae_ctx_set_current(self._prev)
        return False

    def close(self):
        _check(lib(), 'pdae_ctx_destroy', lib().pdae_ctx_destroy(self.handle))
        self._keep = []


_plan_cache = {}


def rows_gemm_plan(M, N, K, w_kn, may_split):
    """(cfg, splits, stream_blocks) of pdae_rows_gemm for a shape (host-side query, cached per arithmetic)."""
    key = (M, N, K, w_kn, may_split, lib().pdae_gemm_arith())
    hit = _plan_cache.get(key)
    if hit is None:
        handle = lib()
        cfg, splits, sb = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _check(handle, 'pdae_rows_gemm_plan',
               handle.pdae_rows_gemm_plan(M, N, K, int(w_kn), int(may_split), ctypes.byref(cfg), ctypes.byref(splits),
                                          ctypes.byref(sb)))
        hit = _plan_cache[key] = (cfg.value, splits.value, sb.value)
    return hit


_wg_cache = {}

GEMM_F32MFMA, GEMM_BF16X3 = 0, 1


def gemm_arith():
    """The row-GEMM family's arithmetic: GEMM_BF16X3 (exact-split bf16, default) or GEMM_F32MFMA (include/pdae.h)."""
    return lib().pdae_gemm_arith()


def set_gemm_arith(arith):
    """pdae_set_gemm_arith; the cached plans / workspace sizes answered for the old arithmetic are dropped."""
    handle = lib()
    _check(handle, 'pdae_set_gemm_arith', handle.pdae_set_gemm_arith(int(arith)))
    _plan_cache.clear()
    _wg_cache.clear()


def rows_wgrad_workspace(M, Ns, Ks):
    """workspace floats of pdae_rows_wgrad for a group of layers (cached)."""
    key = (M, tuple(Ns), tuple(Ks), lib().pdae_gemm_arith())
    hit = _wg_cache.get(key)
    if hit is None:
        handle = lib()
        n = len(Ns)
        arr = ctypes.c_int * n
        floats = ctypes.c_longlong(0)
        _check(handle, 'pdae_rows_wgrad_workspace',
               handle.pdae_rows_wgrad_workspace(M, n, arr(*Ns), arr(*Ks), ctypes.byref(floats)))
        hit = _wg_cache[key] = floats.value
    return hit


def rows_wgrad(on, M, dYs, Xs, dWs, dbs, workspace):
    """pdae_rows_wgrad over lists of tensors (dbs entries may be None)."""
    n = len(dYs)
    parr, iarr = ctypes.c_void_p * n, ctypes.c_int * n
    call('pdae_rows_wgrad', on, M, n, parr(*[ptr(t) for t in dYs]), parr(*[ptr(t) for t in Xs]),
         parr(*[ptr(t) for t in dWs]), parr(*[ptr(t) for t in dbs]),
         iarr(*[t.shape[1] for t in dYs]), iarr(*[t.shape[1] for t in Xs]), ptr(workspace))


def edge_weights_multi(name, on, cos, cins, kps, srcs, dsts):
    """pdae_edge_weight_stack_multi / _unstack_multi over lists (HOST arrays of sizes and device pointers)."""
    n = len(cos)
    parr, iarr = ctypes.c_void_p * n, ctypes.c_int * n
    call(name, on, n, iarr(*cos), iarr(*cins), iarr(*kps), parr(*[ptr(t) for t in srcs]), parr(*[ptr(t) for t in dsts]))


WGRAD_MULTI_MAX = 48


def multi_copy(pairs, strided=()):
    """pdae_multi_copy: pairs = [(dst, src)] contiguous fp32 device tensors of equal numel (src None: dst is zero-filled);
    strided = [(dst, src2d, cols)]: dst (contiguous, rows * cols elements) <- src2d[:, :cols] of a row-major 2-D tensor.
    One launch per 128 entries."""
    n = len(pairs) + len(strided)
    if n == 0:
        return
    on = pairs[0][0] if pairs else strided[0][0]
    for d, s_ in list(pairs) + [(d, s_) for d, s_, _ in strided]:
        for t in (d, s_):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != on.device):
                raise ValueError('multi_copy: contiguous fp32 tensors on one device')
    for d, s_ in pairs:
        if s_ is not None and d.numel() != s_.numel():
            raise ValueError('multi_copy: %r <- %r' % (tuple(d.shape), tuple(s_.shape)))
    for d, s_, c in strided:
        if s_.dim() != 2 or not 0 < c <= s_.shape[1] or d.numel() != s_.shape[0] * c:
            raise ValueError('multi_copy: %r <- %r[:, :%d]' % (tuple(d.shape), tuple(s_.shape), c))
    parr, larr, iarr = ctypes.c_void_p * n, ctypes.c_longlong * n, ctypes.c_int * n
    src = parr(*([ptr(s_) for _, s_ in pairs] + [ptr(s_) for _, s_, _ in strided]))
    dst = parr(*([ptr(d) for d, _ in pairs] + [ptr(d) for d, _, _ in strided]))
    cnt = larr(*([d.numel() for d, _ in pairs] + [d.numel() for d, _, _ in strided]))
    cols = sld = None
    if strided:
        cols = iarr(*([0] * len(pairs) + [c for _, _, c in strided]))
        sld = iarr(*([0] * len(pairs) + [s_.shape[1] for _, s_, _ in strided]))
    call('pdae_multi_copy', on, n, src, dst, cnt, cols, sld)


def rows_wgrad_multi(jobs):
    """pdae_rows_wgrad_multi: jobs = [(dY (M,N), X (M,K), dW (N,K), db (N,) or None)], every layer with its own row
    count, at most WGRAD_MULTI_MAX per launch (longer lists are cut).  The workspace is allocated here."""
    for i in range(0, len(jobs), WGRAD_MULTI_MAX):
        part = jobs[i:i + WGRAD_MULTI_MAX]
        n = len(part)
        parr, iarr = ctypes.c_void_p * n, ctypes.c_int * n
        Ms, Ns, Ks = (iarr(*v) for v in ([j[0].shape[0] for j in part], [j[0].shape[1] for j in part],
                                        [j[1].shape[1] for j in part]))
        key = ('multi', tuple(Ms), tuple(Ns), tuple(Ks), lib().pdae_gemm_arith())
        floats = _wg_cache.get(key)
        if floats is None:
            f = ctypes.c_longlong(0)
            handle = lib()
            _check(handle, 'pdae_rows_wgrad_multi_workspace', handle.pdae_rows_wgrad_multi_workspace(n, Ms, Ns, Ks, ctypes.byref(f)))
            floats = _wg_cache[key] = f.value
        on = part[0][0]
        ws = torch.empty(max(floats, 1), device=on.device, dtype=torch.float32)
        call('pdae_rows_wgrad_multi', on, n, Ms, parr(*[ptr(j[0]) for j in part]), parr(*[ptr(j[1]) for j in part]),
             parr(*[ptr(j[2]) for j in part]), parr(*[ptr(j[3]) for j in part]), Ns, Ks, ptr(ws))
