"""LAB: shader clock while the SHIPPED row-GEMM kernels run (build with -DPDAE_LAB_CLOCK -DPDAE_LAB_OCC: two counters read by
block 0, registers and occupancy as shipped).  bash tools/lab/rows_clock.sh"""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libpdae_lab.so'))
vp, ci = ctypes.c_void_p, ctypes.c_int
L.pdae_rows_gemm.argtypes = [ci, ci, ci, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, vp]
L.pdae_rows_gemm_plan.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp]
b, a, m = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
L.pdae_lab_occupancy(36864, ctypes.byref(b), ctypes.byref(a), ctypes.byref(m))
print('64x64 kernel: %d blocks per CU' % b.value)
for (M, N, K, kn) in [(3584, 1152, 384, 0), (2944, 1536, 384, 0), (3584, 384, 1536, 1), (8192, 1536, 384, 0), (65536, 512, 512, 0)]:
    x = torch.randn(M, K, device='cuda')
    w = torch.randn((K, N) if kn else (N, K), device='cuda') * 0.05
    y = torch.empty(M, N, device='cuda')
    cfg, sp, sb = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    L.pdae_rows_gemm_plan(M, N, K, kn, 0, ctypes.byref(cfg), ctypes.byref(sp), ctypes.byref(sb))
    s = torch.cuda.current_stream().cuda_stream
    f = lambda: L.pdae_rows_gemm(M, N, K, x.data_ptr(), w.data_ptr(), kn, None, 0, None, y.data_ptr(), cfg.value, 1, 0, s)
    for _ in range(20):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    clk = (ctypes.c_longlong * 2)()
    L.pdae_lab_rows_clock(clk)
    ghz = clk[0] / (clk[1] * 10.0)
    tf = 2.0 * M * N * K / us / 1e6
    print(f"rows_gemm {M} x {N} x {K} {'[K,N]' if kn else '[N,K]'} cfg {cfg.value}: {us:7.1f} us {tf:6.1f} TFLOP/s, block 0 lived "
          f"{clk[1] / 100.0:6.1f} us at {ghz:.2f} GHz => {tf / (157.3 * ghz / 2.4):.2f} of the pipe's rate at that clock", flush=True)
