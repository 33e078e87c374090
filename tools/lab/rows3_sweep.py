"""LAB: every (tile shape, split) of the exact-split family on the Transformer blocks' products at several row counts,
next to what pdae_rows_gemm_plan picks -> calibration data for plan3_cost (csrc/rows_gemm.hip).
    gpurun -- python tools/lab/rows3_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib  # noqa: E402


def timed(fn, reps=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    _lib.set_gemm_arith(1)
    # name, N, K, w_kn, epi, may_split
    types = [('qkv', 1152, 384, 0, 0, 0), ('proj', 384, 384, 0, 0, 1), ('fc1+gelu', 1536, 384, 0, 2, 0), ('fc2', 384, 1536, 0, 0, 1),
             ('dh=dy.W2*gelu', 1536, 384, 1, 3, 0), ('dn2=dz.W1', 384, 1536, 1, 0, 1), ('do=ds.Wp', 384, 384, 1, 0, 1),
             ('dn1=dqkv.Wqkv', 384, 1152, 1, 0, 1)]
    for M in (1664, 2560, 3584, 4096, 8192):
        for name, N, K, kn, epi, ms in types:
            x = torch.randn(M, K if not kn else N, device='cuda')          # A is [M, reduction]
            Kr = x.shape[1]
            Nout = N if not kn else K
            # forward: C[M,N] = x[M,K] . W[N,K]^T ; data gradient: C[M,K_in] = dy[M,N_out] . W[N_out,K_in]
            if not kn:
                w = torch.randn(N, K, device='cuda') / K ** 0.5
                mm, nn, kk = M, N, K
            else:
                w = torch.randn(N, K, device='cuda') / K ** 0.5           # (out, in): read as [K'=N][N'=K]
                mm, nn, kk = M, K, N
            z = torch.randn(mm, nn, device='cuda')
            bias = torch.randn(nn, device='cuda') if epi == 2 else None
            y = torch.empty(4, mm, nn, device='cuda')
            res = {}
            for cfg in (16, 17, 18, 19):
                for sp in ((1, 2, 3, 4) if ms else (1,)):
                    if sp > 1 and kk // sp < 128:
                        continue
                    f = lambda: _lib.call('pdae_rows_gemm', x, mm, nn, kk, x.data_ptr(), w.data_ptr(), kn, _lib.ptr(bias), epi,
                                          z.data_ptr(), y.data_ptr(), cfg, sp, 0)
                    res[(cfg, sp)] = timed(f)
            pc, ps, _ = _lib.rows_gemm_plan(mm, nn, kk, bool(kn), bool(ms))
            best = min(res, key=res.get)
            tops = sorted(res.items(), key=lambda kv: kv[1])[:4]
            print(f"M={M:5d} {name:>14} ({mm},{nn},{kk}) plan ({pc},{ps}) {res.get((pc, ps), float('nan')):6.1f} us | best {best} {res[best]:6.1f} us | "
                  + ' '.join(f"{k}:{v:.1f}" for k, v in tops), flush=True)
            del x, w, z, y
    _lib.set_gemm_arith(0)
    print('fp32 kernels, planned:')
    for M in (3584, 8192):
        for name, N, K, kn, epi, ms in types:
            if not kn:
                x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); mm, nn, kk = M, N, K
            else:
                x = torch.randn(M, N, device='cuda'); w = torch.randn(N, K, device='cuda'); mm, nn, kk = M, K, N
            z = torch.randn(mm, nn, device='cuda'); bias = torch.randn(nn, device='cuda') if epi == 2 else None
            pc, ps, sb = _lib.rows_gemm_plan(mm, nn, kk, bool(kn), bool(ms))
            y = torch.empty(max(ps, 1), mm, nn, device='cuda')
            t = timed(lambda: _lib.call('pdae_rows_gemm', x, mm, nn, kk, x.data_ptr(), w.data_ptr(), kn, _lib.ptr(bias), epi,
                                        z.data_ptr(), y.data_ptr(), pc, ps, sb))
            print(f"M={M:5d} {name:>14} fp32 plan ({pc},{ps},{sb}) {t:6.1f} us", flush=True)


if __name__ == '__main__':
    main()
