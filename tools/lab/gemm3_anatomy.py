"""LAB: where a step-sized launch of gemm3_kernel spends its time.  The kernel built -DR3_STAMP writes s_memrealtime
(100 MHz) per block at entry, before the first k-tile, behind the last one and behind the epilogue; this script prints, per
shape: the kernel's duration by HIP events, the spread of the blocks' entry times (launch ramp), the core clock inside the k-loop (s_memtime), and the mean / max of the
prologue (first loads + split + first fragments), the k-loop and the epilogue.
    LABFLAGS=-DR3_STAMP bash tools/lab/build_rows3_lab.sh && gpurun -- python tools/lab/gemm3_anatomy.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'librows3_lab.so'))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lab.lab_gemm3_stamped.argtypes = [i32, i32, i32, i32, vp, vp, i32, vp, vp, vp]
TILES = {0: (128, 128), 2: (128, 192), 3: (128, 64)}


def main():
    torch.manual_seed(0)
    s = torch.cuda.Stream()
    shapes = [(2944, 1152, 384, 0, 0), (2944, 1536, 384, 0, 2), (2944, 384, 384, 0, 3), (2944, 384, 1152, 1, 3),
              (8192, 1536, 384, 0, 0), (8192, 384, 1536, 0, 0)]
    with torch.cuda.stream(s):
        for (M, N, K, kn, v) in shapes:
            A = torch.randn(M, K, device='cuda')
            W = (torch.randn(K, N, device='cuda') if kn else torch.randn(N, K, device='cuda')) * K ** -0.5
            C = torch.empty(M, N, device='cuda')
            bm, bn = TILES[v]
            blocks = 8 * ((-(-M // bm) * -(-N // bn) + 7) // 8)
            stamps = torch.zeros(blocks, 8, dtype=torch.int64, device='cuda')
            run = lambda: lab.lab_gemm3_stamped(v, M, N, K, A.data_ptr(), W.data_ptr(), kn, C.data_ptr(), stamps.data_ptr(),
                                                s.cuda_stream)
            for _ in range(5):
                assert run() == 0
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20):
                run()
            e1.record(s)
            s.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            raw = stamps.cpu().double()
            st = raw[:, :4] * 0.01                                   # microseconds
            live = st[:, 3] > 0                                      # blocks without a tile return before the stamps
            st, raw = st[live], raw[live]
            ghz = ((raw[:, 6] - raw[:, 5]) / ((raw[:, 2] - raw[:, 1]) * 10.0)).mean()     # core cycles per ns in the k-loop
            t0 = st[:, 0].min()
            ent, pro, loop, epi, end = st[:, 0] - t0, st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 3] - t0
            print(f"{M}x{N}x{K} {'KN' if kn else 'NT'} tile {bm}x{bn}: {int(live.sum())} blocks, {us:5.1f} us by events | "
                  f"entry spread {ent.max():4.1f} | prologue {pro.mean():4.1f} (max {pro.max():4.1f}) | k-loop {loop.mean():4.1f} "
                  f"(max {loop.max():4.1f}, {K // 32} tiles) | epilogue {epi.mean():4.1f} (max {epi.max():4.1f}) | "
                  f"last block done at {end.max():4.1f} | clock in the k-loop {ghz:4.2f} GHz = {loop.mean() * ghz * 1e3 / (K // 32):5.0f} cycles per tile",
                  flush=True)


if __name__ == '__main__':
    main()
