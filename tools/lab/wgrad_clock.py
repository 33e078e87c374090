"""LAB: shader clock while the grouped weight-gradient kernel runs (diagnostic build tools/lab/libpdae_lab.so:
bash tools/lab/build_lab.sh).  One encoder-stack-like launch (12 blocks x 4 layers, M = 2944 rows) and one 256-wide
single problem; prints duration, TFLOP/s, the in-kernel clock and the fraction of the matrix pipe's cycles AT THAT CLOCK."""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libpdae_lab.so'))
vp, ci = ctypes.c_void_p, ctypes.c_int
L.pdae_rows_wgrad_multi.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
L.pdae_rows_wgrad_multi_workspace.argtypes = [ci, vp, vp, vp, vp]
L.pdae_lab_wgrad_clock.argtypes = [vp]


def run(tag, probs):
    n = len(probs)
    Ms = (ci * n)(*[p[0] for p in probs])
    Ns = (ci * n)(*[p[1] for p in probs])
    Ks = (ci * n)(*[p[2] for p in probs])
    dys = [torch.randn(m, nn, device='cuda') for m, nn, _ in probs]
    xs = [torch.randn(m, k, device='cuda') for m, _, k in probs]
    dws = [torch.empty(nn, k, device='cuda') for _, nn, k in probs]
    arr = lambda ts: (vp * n)(*[t.data_ptr() for t in ts])
    f = ctypes.c_longlong(0)
    assert L.pdae_rows_wgrad_multi_workspace(n, Ms, Ns, Ks, ctypes.byref(f)) == 0
    ws = torch.empty(max(f.value, 1), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: L.pdae_rows_wgrad_multi(n, Ms, arr(dys), arr(xs), arr(dws), None, Ns, Ks, ws.data_ptr(), s)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    clk = (ctypes.c_longlong * 2)()
    assert L.pdae_lab_wgrad_clock(clk) == 0
    ghz = clk[0] / (clk[1] * 10.0)
    flop = sum(2.0 * m * nn * k for m, nn, k in probs)
    tf = flop / us / 1e6
    print(f"{tag}: {us:8.1f} us (kernel + reduce)  {tf:6.1f} TFLOP/s  in-kernel clock {ghz:.2f} GHz  "
          f"=> {tf / (157.3 * ghz / 2.4):.2f} of the matrix pipe's rate at that clock ({tf / 157.3:.2f} of the 2.4 GHz peak)", flush=True)


run('encoder stack, 12 blocks, M = 2944', [(2944, 1152, 384), (2944, 384, 384), (2944, 1536, 384), (2944, 384, 1536)] * 12)
run('decoder stack,  4 blocks, M = 8192', [(8192, 1152, 384), (8192, 384, 384), (8192, 1536, 384), (8192, 384, 1536)] * 4)
run('one problem 262144 x 512 x 512 (256-wide tile)', [(262144, 512, 512)])
