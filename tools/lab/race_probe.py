"""LAB: are the exact-split kernels' results independent of what else runs on the GPU?  Each product is computed alone
(reference bits), then again while another stream keeps the chip busy with a different product; any differing bit is
a race (or a read of memory the launch did not write)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib as L  # noqa: E402


def gemm(x, w, kn, cfg, splits, y):
    M, K = x.shape
    N = w.shape[1] if kn else w.shape[0]
    L.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), kn, None, 0, None, y.data_ptr(), cfg, splits, 0)


def main():
    L.set_gemm_arith(int(os.environ.get('ARITH', '1')))
    torch.manual_seed(0)
    side = torch.cuda.Stream()
    nx, nw = torch.randn(8192, 1536, device='cuda'), torch.randn(384, 1536, device='cuda')
    ny = torch.empty(8192, 384, device='cuda')
    bad = 0
    for (M, N, K, kn, ms) in [(3584, 1152, 384, 0, 0), (3584, 384, 1536, 0, 1), (1664, 384, 384, 1, 1), (8192, 384, 1152, 1, 1),
                              (3584, 1536, 384, 0, 0), (256, 384, 384, 0, 1), (96, 1152, 384, 0, 0)]:
        x = torch.randn(M, N if kn else K, device='cuda')
        w = torch.randn(N, K, device='cuda') if not kn else torch.randn(N, K, device='cuda')
        mm, nn, kk = (M, N, K) if not kn else (M, K, N)
        cfg, sp, sb = L.rows_gemm_plan(mm, nn, kk, bool(kn), bool(ms))
        y0 = torch.full((sp, mm, nn), float('nan'), device='cuda')
        gemm(x, w, kn, cfg, sp, y0)
        torch.cuda.synchronize()
        for it in range(30):
            y = torch.full((sp, mm, nn), float('nan'), device='cuda')
            with torch.cuda.stream(side):
                for _ in range(3):
                    gemm(nx, nw, 0, -1, 1, ny)
            gemm(x, w, kn, cfg, sp, y)
            torch.cuda.synchronize()
            if not torch.equal(y.view(torch.int32), y0.view(torch.int32)):
                bad += 1
                d = (y != y0).nonzero()
                print('GEMM differs', (mm, nn, kk, kn), 'plan', (cfg, sp), 'iteration', it, 'elements', d.shape[0], 'first', d[0].tolist())
                break
    # grouped weight gradients
    for jobs in ([(3584, n, k) for n, k in ((1152, 384), (384, 384), (1536, 384), (384, 1536))] * 3, [(8192, 1536, 384)], [(4096, 256, 128)],
                 [(2944, 1152, 384), (2944, 384, 384), (2944, 1536, 384), (2944, 384, 1536)]):
        ts = [(torch.randn(m, n, device='cuda'), torch.randn(m, k, device='cuda'), torch.empty(n, k, device='cuda'), torch.empty(n, device='cuda'))
              for m, n, k in jobs]
        L.rows_wgrad_multi(ts)
        torch.cuda.synchronize()
        ref = [(t[2].clone(), t[3].clone()) for t in ts]
        for it in range(20):
            for t in ts:
                t[2].fill_(float('nan')), t[3].fill_(float('nan'))
            with torch.cuda.stream(side):
                for _ in range(3):
                    gemm(nx, nw, 0, -1, 1, ny)
            L.rows_wgrad_multi(ts)
            torch.cuda.synchronize()
            if not all(torch.equal(t[2], r[0]) and torch.equal(t[3], r[1]) for t, r in zip(ts, ref)):
                bad += 1
                print('WGRAD differs', jobs[:2], 'iteration', it)
                break
    print('differing cases:', bad)


if __name__ == '__main__':
    main()
