"""Lab: sweep the tile shapes / split counts of pdae_rows_gemm on the Transformer-block shapes and
compare with the planned choice and with torch.mm (hipBLASLt).  usage: lab_rows.py [M,M,...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import _lib
from point_dae_amd.graph_step import use_created_stream

use_created_stream()


def timeit(fn, iters=20, warm=2, reps=3):
    """us per call inside a replayed hipGraph of `iters` back-to-back calls (what the step does; eager
    launches of 5-20 us kernels measure the Python launch path instead)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for _ in range(iters): fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


Ms = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1664, 2944, 4096, 8192]
# (name, N, K, w_kn, epi)
layers = [('qkv', 1152, 384, 0, 0), ('proj', 384, 384, 0, 0), ('fc1+gelu', 1536, 384, 0, 2), ('fc2', 384, 1536, 0, 0),
          ('dh*gelu\'', 1536, 384, 1, 3), ('dn2', 384, 1536, 1, 0), ('do', 384, 384, 1, 0), ('dn1', 384, 1152, 1, 0)]
CFG = ['128x128', '64x128', '128x64', '64x64', '64x192', '96x128', '128x96', '128x192']
for M in Ms:
    tot_best = tot_plan = tot_lib = 0.
    for name, N, K, w_kn, epi in layers:
        x = torch.randn(M, K, device='cuda')
        w = torch.randn(K, N, device='cuda') * 0.05 if w_kn else torch.randn(N, K, device='cuda') * 0.05
        b = torch.randn(N, device='cuda') if epi == 2 else None
        z = torch.randn(M, N, device='cuda') if epi in (2, 3) else None
        y = torch.empty(4, M, N, device='cuda')
        fl = 2.0 * M * N * K / 1e6
        res = {}
        for cfg in range(8):
            for sp in ((1, 2, 3, 4) if epi == 0 and N == 384 else (1,)):
                f = lambda: _lib.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), w_kn, _lib.ptr(b), epi,
                                      _lib.ptr(z), y.data_ptr(), cfg, sp, 0)
                res[(cfg, sp)] = timeit(f)
        pc, ps, psb = _lib.rows_gemm_plan(M, N, K, w_kn, epi == 0 and N == 384)
        if psb:
            res[(pc, ps)] = timeit(lambda: _lib.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), w_kn, None, 0,
                                                       None, y.data_ptr(), pc, ps, psb))
            name = name + '*'          # planned = stream-K
        lib = timeit((lambda: torch.mm(x, w)) if w_kn else (lambda: torch.mm(x, w.t())))
        best = min(res, key=res.get)
        top = sorted(res, key=res.get)[:4]
        tot_best += res[best]; tot_plan += res[(pc, ps)]; tot_lib += lib
        print(f"M{M:5d} {name:9s} N{N:4d} K{K:4d}: plan {CFG[pc]}/{ps} {res[(pc, ps)]:6.1f}us | best " +
              "  ".join(f"{CFG[c]}/{s_} {res[(c, s_)]:5.1f}us ({fl / res[(c, s_)]:5.1f}TF)" for c, s_ in top) +
              f" | lib {lib:6.1f}us ({fl / lib:5.1f}TF)", flush=True)
    print(f"M{M:5d} totals: best {tot_best:.1f}us  plan {tot_plan:.1f}us  lib {tot_lib:.1f}us", flush=True)
    # grouped weight gradients of a block vs four library GEMMs
    dims = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]
    dys = [torch.randn(M, n, device='cuda') for n, _ in dims]
    xs = [torch.randn(M, k, device='cuda') for _, k in dims]
    dws = [torch.empty(n, k, device='cuda') for n, k in dims]
    dbs = [None, None, torch.empty(1536, device='cuda'), None]
    Ns, Ks = [n for n, _ in dims], [k for _, k in dims]
    ws = torch.empty(_lib.rows_wgrad_workspace(M, Ns, Ks), device='cuda')
    t = timeit(lambda: _lib.rows_wgrad(dys[0], M, dys, xs, dws, dbs, ws))
    print(f"M{M:5d} wgrad group (stream-K, {ws.numel() * 4 / 1e6:.0f} MB of partials): {t:6.1f}us ({sum(2.0 * M * n * k for n, k in dims) / 1e6 / t:5.1f}TF)", flush=True)
    tl = timeit(lambda: [torch.mm(dy.t(), x) for dy, x in zip(dys, xs)] + [dys[2].sum(0)])
    print(f"M{M:5d} wgrad lib (4 mm + colsum): {tl:6.1f}us", flush=True)
