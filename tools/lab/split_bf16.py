"""LAB: what would bf16x3 split arithmetic buy the step's GEMMs?  (tools/lab/split_bf16_gemm.hip)

Times C = A . B^T for the step's dominant shapes three ways -- the product's fp32-MFMA rows_gemm, the lab kernel with
6 / 3 / 1 bf16 products on PRE-SPLIT operands (the split pass timed separately) -- and reports each one's error against
an fp64 product.  Not part of the product path: the parity contract is fp32 arithmetic.
    gpurun -- python tools/lab/split_bf16.py
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import nn_ops  # noqa: E402

SO = os.path.join(ROOT, 'gpurun_out', 'libsplit_bf16.so')
os.makedirs(os.path.dirname(SO), exist_ok=True)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC',
                       os.path.join(ROOT, 'tools', 'lab', 'split_bf16_gemm.hip'), '-o', SO])
lab = ctypes.CDLL(SO)
SO2 = os.path.join(ROOT, 'gpurun_out', 'libsplit_bf16_2.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC',
                       os.path.join(ROOT, 'tools', 'lab', 'split_bf16_gemm2.hip'), '-o', SO2])
lab2 = ctypes.CDLL(SO2)
SO3 = os.path.join(ROOT, 'gpurun_out', 'libsplit_bf16_3.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC',
                       os.path.join(ROOT, 'tools', 'lab', 'split_bf16_gemm3.hip'), '-o', SO3])
lab3 = ctypes.CDLL(SO3)
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lab.lab_split3.argtypes = [i64, vp, vp, vp]
lab.lab_gemm_bf16x3.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp]
lab2.lab_gemm_split.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp]
lab3.lab_gemm_split3.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp]


def stream():
    return torch.cuda.current_stream().cuda_stream


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def split(x):
    p = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    lab.lab_split3(x.numel(), x.data_ptr(), p.data_ptr(), stream())
    return p


def main():
    torch.manual_seed(0)
    print(f"{'shape (M,N,K)':>22} {'fp32 MFMA us':>13} {'err':>9} | {'bf16x6 us':>10} {'err':>9} | {'bf16x3 us':>10} "
          f"{'err':>9} | {'bf16x1 us':>10} {'err':>9} | {'split A us':>10}")
    for (M, N, K) in [(3584, 1152, 384), (3584, 1536, 384), (3584, 384, 1536), (8192, 1152, 384), (8192, 1536, 384),
                      (8192, 384, 1536), (131072, 256, 128), (131072, 512, 256)]:
        A = torch.randn(M, K, device='cuda')
        W = torch.randn(N, K, device='cuda') * K ** -0.5
        ref = (A.double() @ W.double().t())
        scale = ref.abs().max().item()
        t32 = timed(lambda: nn_ops.rows_gemm(A, W))
        e32 = (nn_ops.rows_gemm(A, W).double() - ref).abs().max().item() / scale
        A3, W3 = split(A), split(W)
        C = torch.empty(M, N, device='cuda')
        row = f"{str((M, N, K)):>22} {t32:13.1f} {e32:9.1e} |"
        for nprod in (6, 3, 1):
            f = lambda: lab.lab_gemm_bf16x3(M, N, K, A3.data_ptr(), W3.data_ptr(), C.data_ptr(), nprod, stream())
            t = timed(f)
            f()
            err = (C.double() - ref).abs().max().item() / scale
            row += f" {t:10.1f} {err:9.1e} |"
        ts = timed(lambda: lab.lab_split3(A.numel(), A.data_ptr(), A3.data_ptr(), stream()))
        row += f" {ts:10.1f} || in-kernel split:"
        for nprod in (6, 3, 1):
            f = lambda: lab2.lab_gemm_split(M, N, K, A.data_ptr(), W.data_ptr(), C.data_ptr(), nprod, stream())
            C.zero_()
            t = timed(f)
            f()
            err = (C.double() - ref).abs().max().item() / scale
            row += f" x{nprod} {t:7.1f} us {err:8.1e}"
        row += " || interleaved:"
        for nprod in (6, 3, 1):
            f = lambda: lab3.lab_gemm_split3(M, N, K, A.data_ptr(), W.data_ptr(), C.data_ptr(), nprod, stream())
            C.zero_()
            t = timed(f)
            f()
            err = (C.double() - ref).abs().max().item() / scale
            row += f" x{nprod} {t:7.1f} us {err:8.1e}"
        print(row, flush=True)


if __name__ == '__main__':
    main()
