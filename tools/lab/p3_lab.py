"""LAB: rows3p::gemm3p_kernel (plane operands, LDS-DMA) against the shipped exact-split gemm3_kernel: bit-equality of
the results and time per launch (back to back, HIP events; and cold: operands evicted by a 600 MB fill between launches).
    bash tools/lab/build_p3_lab.sh && gpurun -- python tools/lab/p3_lab.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib, nn_ops  # noqa: E402

lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libp3_lab.so'))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lab.lab_split3.argtypes = [vp, i64, i32, vp, vp]
lab.lab_gemm3p.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]
NAMES = {5: '128x128 mid', 6: '128x64 mid', 0: '128x128 3buf', 1: '128x128 2buf', 2: '128x192 2buf', 3: '128x64 3buf', 4: '128x64 2buf'}


def stream():
    return torch.cuda.current_stream().cuda_stream


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def split3(x):
    R, C = x.shape
    out = torch.empty(3, R, C, device=x.device, dtype=torch.int16)
    rc = lab.lab_split3(x.data_ptr(), R, C, out.data_ptr(), stream())
    assert rc == 0
    return out


def main():
    variants = [int(v) for v in os.environ.get('VARIANTS', '0,5,3,6').split(',')]
    torch.manual_seed(0)
    shapes = [(2944, 1152, 384), (2944, 1536, 384), (2944, 384, 1536), (2944, 384, 384), (2944, 384, 1152),
              (1664, 1152, 384), (4096, 1536, 384),
              (8192, 1152, 384), (8192, 1536, 384), (8192, 384, 1536), (8192, 384, 384), (8192, 384, 1152),
              (3000, 1100, 384), (65536, 512, 512), (512, 512, 32768)]
    if os.environ.get('SHAPES'):
        shapes = [tuple(int(x) for x in sh.split('x')) for sh in os.environ['SHAPES'].split(',')]
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device='cuda')
        W = torch.randn(N, K, device='cuda') * K ** -0.5
        A3, W3 = split3(A), split3(W)
        # the split is exact: h + m + l == x in fp32 arithmetic order (h + m) + l
        planes = lambda p: (p.view(torch.bfloat16).float())
        assert torch.equal((planes(A3)[0] + planes(A3)[1]) + planes(A3)[2], A)
        refs = {}
        row = f"{str((M, N, K)):>20} |"
        for cfg in (16, 18, 19):
            C0 = torch.empty(M, N, device='cuda')
            f = lambda: _lib.call('pdae_rows_gemm', A, M, N, K, _lib.ptr(A), _lib.ptr(W), 0, None, 0, None, _lib.ptr(C0), cfg, 1, 0)
            f()
            refs[cfg] = C0.clone()
            row += f" gemm3 cfg{cfg} {timed(f):6.1f} |"
        ref = refs[16]
        assert torch.equal(ref, refs[18]) and torch.equal(ref, refs[19])
        planned = lambda: nn_ops.rows_gemm(A, W, may_split=(N == 384))
        row += f" planned {timed(planned):6.1f} |"
        C = torch.empty(M, N, device='cuda')
        for v in variants:
            for splits in ((1, 2, 3) if (N == 384 and v in (0, 3, 5, 6)) else (1,)):
                C = torch.full((splits, M, N), float('nan'), device='cuda')
                f = lambda: lab.lab_gemm3p(v, M, N, K, A3.data_ptr(), W3.data_ptr(), C.data_ptr(), splits, stream(), None)
                rc = f()
                torch.cuda.synchronize()
                if rc != 0:
                    row += f" v{v} rc={rc} |"
                    continue
                ok = torch.equal(C[0], ref) if splits == 1 else bool(((C.sum(0) - ref).abs().max() / ref.abs().max()) < 1e-5)
                t = timed(f)
                row += f" {NAMES[v]}{'/s%d' % splits if splits > 1 else ''} {t:6.1f} {'==' if ok else 'DIFF'} |"
        print(row, flush=True)


if __name__ == '__main__':
    main()
