"""LAB: the library's explicitly packed kernels (Chamfer forward, tiled / many / packed forms) looped beside the reduced
GEMM loop of tools/xproc_repro.hip running in another process: any iteration whose outputs differ from the first?"""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib  # noqa: E402
from point_dae_amd.graph_step import use_created_stream  # noqa: E402
use_created_stream()
torch.manual_seed(0)
cases = {'tiled 8x16384x1024': (8, 16384, 1024), 'square 16x1024x1024': (16, 1024, 1024), 'packed 5248x32x32': (5248, 32, 32)}
agg = subprocess.Popen([os.path.join(ROOT, 'tools', 'lab', 'lab_xproc'), 'G', str(float(os.environ.get('SECS', '14')))]) if os.environ.get('AGG', '1') == '1' else None
time.sleep(1.5)
for name, (B, n, m) in cases.items():
    a, b = torch.randn(B, n, 3, device='cuda'), torch.randn(B, m, 3, device='cuda')
    d1, d2 = torch.empty(B, n, device='cuda'), torch.empty(B, m, device='cuda')
    i1, i2 = torch.empty(B, n, dtype=torch.int32, device='cuda'), torch.empty(B, m, dtype=torch.int32, device='cuda')
    ref, bad, its = None, 0, 0
    t0 = time.time()
    while time.time() - t0 < 3.0:
        _lib.call('pdae_chamfer_forward', a, B, n, _lib.ptr(a), m, _lib.ptr(b), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2))
        cur = (d1.clone(), d2.clone(), i1.clone(), i2.clone())
        torch.cuda.synchronize()
        if ref is None:
            ref = cur
        else:
            bad += not all(torch.equal(x, y) for x, y in zip(cur, ref))
        its += 1
    print(f'chamfer forward {name}: {its} iterations, {bad} differing from the first', flush=True)
if agg:
    agg.wait()
