"""LAB: the exact-split bf16 row GEMM variants of tools/lab/rows3_lab.hip against the fp32-MFMA rows_gemm: time per
launch (back to back, HIP events) and max |C - fp64| / max |C|.
    bash tools/lab/build_rows3_lab.sh && gpurun -- python tools/lab/rows3_lab.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib, nn_ops  # noqa: E402

lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', os.environ.get('LABSO', 'librows3_lab.so')))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lab.lab_gemm3.argtypes = [i32, i32, i32, i32, vp, vp, i32, vp, vp]
NAMES = {0: '128x128 8w k32', 1: '128x128 8w k32 1acc', 2: '128x192 8w k32', 3: '128x64 8w k32', 4: '64x128 4w k16',
         5: '128x64 4w(64x32) k16', 6: '128x128 4w k16', 7: '128x128 8w k16', 8: '128x64 4w(32x64) k16', 9: '64x128 4w k32',
         10: 'v0 noGload', 11: 'v0 noSplit', 12: 'v0 noLstore', 13: 'v0 noBarrier', 14: 'v0 noFrag', 17: 'v0 MFMA only',
         20: 'v4 noGload', 21: 'v4 noSplit', 22: 'v4 noLstore', 23: 'v4 noBarrier', 24: 'v4 noFrag', 27: 'v4 MFMA only'}


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


def main():
    variants = [int(v) for v in os.environ.get('VARIANTS', '0,1,2,3,4,6,7').split(',')]
    torch.manual_seed(0)
    shapes = [(3584, 1152, 384, 0), (3584, 1536, 384, 0), (3584, 384, 1536, 0), (3584, 384, 384, 0),
              (8192, 1152, 384, 0), (8192, 1536, 384, 0), (8192, 384, 1536, 0),
              (3584, 384, 1152, 1), (3584, 1536, 384, 1), (3584, 384, 1536, 1), (8192, 1536, 384, 1),
              (1664, 1152, 384, 0), (2944, 1536, 384, 0), (65536, 512, 512, 0), (3000, 1100, 384, 0), (3000, 1100, 384, 1),
              (1024, 512, 4096, 0), (1024, 512, 8192, 0), (512, 512, 32768, 0)]
    if os.environ.get('SHAPES'):
        shapes = [tuple(int(x) for x in sh.split('x')) for sh in os.environ['SHAPES'].split(',')]
    for (M, N, K, kn) in shapes:
        A = torch.randn(M, K, device='cuda')
        W = (torch.randn(K, N, device='cuda') if kn else torch.randn(N, K, device='cuda')) * K ** -0.5
        ref = A.double() @ (W.double() if kn else W.double().t())
        scale = ref.abs().max().item()
        f32 = lambda: nn_ops.rows_gemm(A, W, w_kn=bool(kn))
        t32 = timed(f32)
        e32 = (f32().double() - ref).abs().max().item() / scale
        row = f"{str((M, N, K)):>20} {'KN' if kn else 'NT'} | fp32 {t32:7.1f} us {e32:.1e} |"
        C = torch.empty(M, N, device='cuda')
        for v in variants:
            C.fill_(float('nan'))
            f = lambda: lab.lab_gemm3(v, M, N, K, A.data_ptr(), W.data_ptr(), kn, C.data_ptr(), stream())
            rc = f()
            torch.cuda.synchronize()
            if rc != 0:
                row += f" v{v} rc={rc} |"
                continue
            err = (C.double() - ref).abs().max().item() / scale
            t = timed(f)
            row += f" {NAMES[v]} {t:6.1f} us {err:.1e} |"
        print(row, flush=True)


if __name__ == '__main__':
    main()
