"""Lab: where a row-GEMM block spends its time (diagnostic build tools/lab/libpdae_lab.so with in-kernel
stamps).  usage: stamps.py M N K w_kn epi cfg splits"""
import ctypes, os, sys
import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libpdae_lab%s.so' % os.environ.get('LABTAG', '')))
vp, ci = ctypes.c_void_p, ctypes.c_int
L.pdae_rows_gemm.argtypes = [ci, ci, ci, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, vp]
L.pdae_lab_set_stamps.argtypes = [vp]
L.pdae_last_error.restype = ctypes.c_char_p


def run(M, N, K, w_kn, epi, cfg, splits):
    x = torch.randn(M, K, device='cuda')
    w = torch.randn(K, N, device='cuda') * 0.05 if w_kn else torch.randn(N, K, device='cuda') * 0.05
    b = torch.randn(N, device='cuda') if epi == 2 else None
    z = torch.randn(M, N, device='cuda') if epi in (2, 3) else None
    y = torch.empty(max(splits, 1), M, N, device='cuda')
    st = torch.zeros(8192 * 8, dtype=torch.int64, device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: L.pdae_rows_gemm(M, N, K, x.data_ptr(), w.data_ptr(), w_kn, b.data_ptr() if b is not None else None, epi,
                                    z.data_ptr() if z is not None else None, y.data_ptr(), cfg, splits, 0, s)
    L.pdae_lab_set_stamps(None)
    for _ in range(3):
        assert call() == 0, L.pdae_last_error()
    torch.cuda.synchronize()
    L.pdae_lab_set_stamps(st.data_ptr())
    # a preceding kernel so the launch is back to back like in the step
    call(); st.zero_(); call()
    torch.cuda.synchronize()
    a = st.cpu().numpy().reshape(-1, 8)
    a = a[a[:, 4] != 0]
    t0 = a[:, 4].min()
    start = (a[:, 4] - t0) / 100.0          # us (100 MHz)
    end = (a[:, 6] - t0) / 100.0
    cyc = lambda i, j: (a[:, j] - a[:, i])
    clk = (a[:, 3] - a[:, 0]) / np.maximum((a[:, 6] - a[:, 4]), 1) * 100e6 / 1e9
    xcc = a[:, 5] >> 32
    print(f"M{M} N{N} K{K} kn{w_kn} epi{epi} cfg{cfg} splits{splits}: blocks {len(a)}  span {end.max():.1f}us  "
          f"start skew p50 {np.median(start):.2f} max {start.max():.2f}us  end p10 {np.percentile(end, 10):.1f} p50 {np.median(end):.1f}")
    for name, i, j in (('prologue', 0, 1), ('k-loop', 1, 2), ('epilogue', 2, 3)):
        c = cyc(i, j)
        print(f"   {name:9s} cycles p10 {np.percentile(c, 10):8.0f} p50 {np.median(c):8.0f} p90 {np.percentile(c, 90):8.0f} max {c.max():8.0f}")
    print(f"   clock (s_memtime/s_memrealtime) p50 {np.median(clk):.2f} GHz;  blocks per XCC {np.bincount(xcc.astype(int), minlength=8).tolist()}")
    # blocks per CU: HW_ID bits (cu id 8..11, sh 12, se 13..15 on gfx9) + xcc
    hw = a[:, 5] & 0xffffffff
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    per = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f"   distinct CUs {len(per)}  blocks per CU: " + " ".join(f"{k}:{v}" for k, v in zip(*np.unique(per, return_counts=True))))
    # blocks that started after another block had already ended (a second residency round), and concurrency per CU
    first_end = end.min()
    late = start > first_end
    print(f"   first block ends at {first_end:.1f} us; {int(late.sum())} blocks start after that (second round); "
          f"late blocks per CU: " + " ".join(f"{k}:{v}" for k, v in zip(*np.unique(np.bincount(np.unique(cu, return_inverse=True)[1], weights=late).astype(int), return_counts=True))))


if __name__ == '__main__':
    args = [int(v) for v in sys.argv[1:]]
    if args:
        run(*args)
    else:
        for cfgsp in ((3, 1), (0, 1), (1, 1)):
            run(2944, 384, 384, 0, 0, *cfgsp)
        for cfgsp in ((0, 1), (2, 1), (1, 1), (3, 1)):
            run(2944, 1152, 384, 0, 0, *cfgsp)
        run(2944, 384, 1536, 0, 0, 3, 3)
        run(8192, 1536, 384, 0, 2, 0, 1)
        run(8192, 1536, 384, 0, 2, 6, 1)
