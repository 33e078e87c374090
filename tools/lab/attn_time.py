"""LAB: the attention core's launch times (back to back, HIP events) at the step's shapes; PDAE_ATTN=f32 keeps the fp32-input kernels."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib
from point_dae_amd.graph_step import use_created_stream
use_created_stream()
for (B, T, H) in ((128, 64, 6), (128, 23, 6), (128, 47, 6), (32, 64, 6)):
    qkv = torch.randn(B * T, 3 * H * 64, device='cuda')
    o = torch.empty(B * T, H * 64, device='cuda'); lse = torch.empty(B, H, T, device='cuda')
    go = torch.randn_like(o); dqkv = torch.empty_like(qkv)
    f = lambda: _lib.call('pdae_attention_forward', qkv, B, T, H, 64, 0.125, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse))
    b = lambda: _lib.call('pdae_attention_backward', qkv, B, T, H, 64, 0.125, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse), _lib.ptr(go), _lib.ptr(dqkv))
    out = []
    for fn in (f, b):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f'B={B} T={T} H={H}: forward {out[0]:.1f} us, backward {out[1]:.1f} us  (PDAE_ATTN={os.environ.get("PDAE_ATTN", "default")})', flush=True)
