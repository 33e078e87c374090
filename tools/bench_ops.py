"""Micro-benchmark of the geometry kernels at the BASELINE shapes (run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import chamfer_dist, pointnet2_utils as pu, emd
from point_dae_amd.knn_cuda import knn
from point_dae_amd.synthetic import shapenet_like_clouds


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=0)).cuda()
    res = {}
    res["fps_1024_64"] = timeit(lambda: pu.furthest_point_sample_with_centres(x, 64))
    res["fps_1024_512"] = timeit(lambda: pu.furthest_point_sample_with_centres(x, 512))
    _, c512 = pu.furthest_point_sample_with_centres(x, 512)
    res["fps_512_128"] = timeit(lambda: pu.furthest_point_sample_with_centres(c512, 128))
    _, c64 = pu.furthest_point_sample_with_centres(x, 64)
    res["knn_1024_64_32"] = timeit(lambda: knn(x, c64, 32, with_neighbourhood=True))
    res["ball_1024_512_r.2_32"] = timeit(lambda: pu.ball_query(0.2, 32, x, c512))
    _, c128 = pu.furthest_point_sample_with_centres(c512, 128)
    res["ball_512_128_r.4_64"] = timeit(lambda: pu.ball_query(0.4, 64, c512, c128))
    idx = pu.ball_query(0.4, 64, c512, c128)
    feats = torch.randn(B, 131, 512, device="cuda")
    res["group_131x512_128x64"] = timeit(lambda: pu.grouping_operation(feats, idx))
    go = torch.randn(B, 131, 128, 64, device="cuda")
    from point_dae_amd import _lib
    gp = torch.empty(B, 131, 512, device="cuda")
    res["group_grad_131x512_128x64"] = timeit(lambda: _lib.call(
        "pdae_group_points_grad", go, B, 131, 512, 128, 64, go.data_ptr(), idx.data_ptr(), gp.data_ptr()))
    M = 41
    a = torch.rand(B * M, 32, 3, device="cuda")
    b = torch.rand(B * M, 32, 3, device="cuda")
    res["chamfer_fwd_%dx32x32" % (B * M)] = timeit(lambda: chamfer_dist.forward(a, b))
    d1, d2, i1, i2 = chamfer_dist.forward(a, b)
    res["chamfer_bwd_%dx32x32" % (B * M)] = timeit(lambda: chamfer_dist.backward(a, b, i1, i2, d1, d2))
    a = torch.rand(B, 1024, 3, device="cuda"); b = torch.rand(B, 1024, 3, device="cuda")
    res["chamfer_fwd_1024x1024"] = timeit(lambda: chamfer_dist.forward(a, b))
    a = torch.rand(B, 16384, 3, device="cuda")
    res["chamfer_fwd_16384x1024"] = timeit(lambda: chamfer_dist.forward(a, b))
    d1, d2, i1, i2 = chamfer_dist.forward(a, b)
    res["chamfer_bwd_16384x1024"] = timeit(lambda: chamfer_dist.backward(a, b, i1, i2, d1, d2))
    a = torch.rand(8, 1024, 3, device="cuda"); b = torch.rand(8, 1024, 3, device="cuda")
    res["emd_match_8x1024x1024"] = timeit(lambda: emd.approxmatch_forward(a, b), iters=3, warm=1)
    for k, v in res.items():
        print(f"{k:32s} {v:10.1f} us")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_ops.json", "w"), indent=1)


if __name__ == "__main__":
    main()
