"""LAB aggressor 2: loop one variant of tools/lab/rows3_lab.hip (gemm3_kernel and its ablations) for SECS seconds."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'librows3_lab.so'))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lab.lab_gemm3.argtypes = [i32, i32, i32, i32, vp, vp, i32, vp, vp]
v = int(os.environ.get('V', '0'))
M, N, K = (int(t) for t in os.environ.get('SHAPE', '2944x1152x384').split('x'))
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
s = torch.cuda.Stream()
secs = float(os.environ.get('SECS', '8'))
t0, n = time.time(), 0
with torch.cuda.stream(s):
    while time.time() - t0 < secs:
        for _ in range(50):
            rc = lab.lab_gemm3(v, M, N, K, A.data_ptr(), W.data_ptr(), 0, C.data_ptr(), s.cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        n += 50
print('agg2: variant %d, %d launches' % (v, n), flush=True)
