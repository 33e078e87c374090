"""LAB: shader clock inside rows3p::gemm3p_kernel's k-loop (s_memtime / s_memrealtime stamps, build with
LABFLAGS=-DP3_STAMPS): a long reduction on 16 CUs against the same on every CU."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libp3_lab.so'))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lab.lab_split3.argtypes = [vp, i64, i32, vp, vp]
lab.lab_gemm3p.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]
st = lambda: torch.cuda.current_stream().cuda_stream
def split3(x):
    out = torch.empty(3, *x.shape, device=x.device, dtype=torch.int16)
    assert lab.lab_split3(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), st()) == 0
    return out
for (M, N, K, zero) in [tuple(int(t) for t in sh.split('x')) for sh in os.environ.get('SHAPES', '512x512x16384x0,2048x2048x16384x0,2048x2048x16384x1,4096x4096x4096x0,65536x512x512x0').split(',')]:
    A = torch.randn(M, K, device='cuda') * (0 if zero else 1)
    W = torch.randn(N, K, device='cuda') * (0 if zero else 1)
    A3, W3 = split3(A), split3(W)
    C = torch.empty(M, N, device='cuda')
    for v in [int(t) for t in os.environ.get('VARIANTS', '0,5').split(',')]:
        stamps = torch.zeros(4096, 4, dtype=torch.int64, device='cuda')
        f = lambda: lab.lab_gemm3p(v, M, N, K, A3.data_ptr(), W3.data_ptr(), C.data_ptr(), 1, st(), stamps.data_ptr())
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        s = stamps[stamps[:, 2] > 0].double()
        cyc, rt = (s[:, 2] - s[:, 0]), (s[:, 3] - s[:, 1])
        ghz = (cyc / rt * 0.1).median().item()
        print(f"{(M, N, K)} zero={zero} v{v}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s-eq | k-loop {cyc.median().item() / (K / 32):7.0f} cyc/k-tile "
              f"{rt.median().item() * 10 / (K / 32):6.1f} ns/k-tile clock {ghz:.2f} GHz (blocks {len(s)})", flush=True)
