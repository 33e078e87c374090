"""LAB: what cold operands cost a step-sized row GEMM.  Time per launch of pdae_rows_gemm (planned tile) replayed from a
hipGraph: back to back on resident operands, against each launch behind a 640 MB fill (Infinity Cache and L2 flushed;
the fill's own time measured separately and subtracted), with the weight alone re-warmed, and with the rows re-warmed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib, nn_ops  # noqa: E402


def graph_time(body, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    torch.manual_seed(0)
    junk = torch.empty(160 << 20, device='cuda')          # 640 MB
    for (M, N, K, split) in [(2944, 1152, 384, False), (2944, 1536, 384, False), (2944, 384, 1536, True), (2944, 384, 384, True),
                             (8192, 1536, 384, False), (8192, 384, 1536, True)]:
        A = torch.randn(M, K, device='cuda')
        W = torch.randn(N, K, device='cuda') * K ** -0.5
        gemm = lambda: nn_ops.rows_gemm(A, W, may_split=split)
        warmW = lambda: W.sum()
        warmA = lambda: A.sum()
        fill = lambda: junk.fill_(1.0)
        t_hot = graph_time(gemm)
        t_fill = graph_time(fill)
        t_cold = graph_time(lambda: (fill(), gemm())) - t_fill
        t_fw = graph_time(lambda: (fill(), warmW()))
        t_wW = graph_time(lambda: (fill(), warmW(), gemm())) - t_fw
        t_fa = graph_time(lambda: (fill(), warmA()))
        t_wA = graph_time(lambda: (fill(), warmA(), gemm())) - t_fa
        print(f"{(M, N, K)}: hot {t_hot:6.1f} us | cold {t_cold:6.1f} | cold, weight re-warmed {t_wW:6.1f} | cold, rows re-warmed {t_wA:6.1f}  (fill {t_fill:.0f} us)", flush=True)


if __name__ == '__main__':
    main()
