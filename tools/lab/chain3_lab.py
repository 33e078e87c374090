"""LAB (round 6, VERDICT r5 item 1, stage gate): a GEMM -> GEMM seam inside ONE persistent launch (band-local counters,
tools/lab/chain3_lab.hip) against the same two products as two launches, at the encoder MLP pair's size.
    bash tools/lab/build_chain3_lab.sh && gpurun -- python tools/lab/chain3_lab.py
Prints: bit-equality of both results, time per pair from graph replays (HIP events), and the in-kernel timeline of the
chained launch (s_memrealtime stamps per block and unit): when the fc2 units start, how long they wait, what the seam costs."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib, nn_ops  # noqa: E402
from point_dae_amd.graph_step import use_created_stream  # noqa: E402

lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'lab', 'libchain3_lab.so'))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lab.lab_chain3.argtypes = [i32] * 5 + [vp] * 8 + [i32, i32, vp]
lab.lab_two_launches.argtypes = [i32] * 5 + [vp] * 5 + [vp]
MAXU = 4


def stream():
    return torch.cuda.current_stream().cuda_stream


def graph_ms(fn, per_graph=20, replays=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / replays / per_graph * 1e3      # us per pair


def main():
    use_created_stream()
    torch.manual_seed(0)
    N1, K1, N2, S = 1536, 384, 384, 3
    blocks = 256
    for T in (23, 13, 32):
        M = 128 * T
        X = torch.randn(M, K1, device='cuda')
        W1 = torch.randn(N1, K1, device='cuda') * K1 ** -0.5
        W2 = torch.randn(N2, N1, device='cuda') * N1 ** -0.5
        H0, Y0 = torch.empty(M, N1, device='cuda'), torch.empty(S, M, N2, device='cuda')
        H1, Y1 = torch.full((M, N1), float('nan'), device='cuda'), torch.full((S, M, N2), float('nan'), device='cuda')
        bands = (M + 127) // 128
        cnt = torch.zeros(bands * S, dtype=torch.int32, device='cuda')
        sync = torch.zeros(4, dtype=torch.int32, device='cuda')
        stamps = torch.zeros(blocks * MAXU * 4, dtype=torch.int64, device='cuda')

        def two():
            rc = lab.lab_two_launches(M, N1, K1, N2, S, X.data_ptr(), W1.data_ptr(), H0.data_ptr(), W2.data_ptr(), Y0.data_ptr(), stream())
            assert rc == 0, rc

        def chain(waits=1):
            rc = lab.lab_chain3(M, N1, K1, N2, S, X.data_ptr(), W1.data_ptr(), H1.data_ptr(), W2.data_ptr(), Y1.data_ptr(),
                                cnt.data_ptr(), sync.data_ptr(), stamps.data_ptr(), waits, blocks, stream())
            assert rc == 0, rc

        def shipped():      # what the step launches today: the planned tile shapes (128 x 192 for fc1, slabs for fc2)
            h = nn_ops.rows_gemm(X, W1)
            cfg, splits, sb = _lib.rows_gemm_plan(M, N2, N1, False, 8)
            y = torch.empty(max(splits, 1), M, N2, device='cuda')
            _lib.call('pdae_rows_gemm', h, M, N2, N1, _lib.ptr(h), _lib.ptr(W2), 0, None, 0, None, _lib.ptr(y), cfg, splits, sb)
            return y
        two()
        chain()
        torch.cuda.synchronize()
        ok_h, ok_y = torch.equal(H0, H1), torch.equal(Y0, Y1)
        ref = (X.double() @ W1.double().t()) @ W2.double().t()
        err = (Y1.sum(0).double() - ref).abs().max().item() / ref.abs().max().item()
        # 200 chained launches back to back, results re-checked: a stale read shows up under load, not on the first launch
        bad = 0
        for _ in range(200):
            Y1.fill_(float('nan'))
            chain()
            bad += int(not torch.equal(Y0, Y1))
        t_two, t_chain, t_ship = graph_ms(two), graph_ms(chain), graph_ms(shipped)
        t_free = graph_ms(lambda: chain(0))
        chain()
        torch.cuda.synchronize()
        print(f'M = {M} (T_vis {T}): H equal {ok_h}, Y equal {ok_y}, 200 repeats differing {bad}, give-ups {int(sync[2])}, err vs fp64 {err:.1e}')
        print(f'   two launches (128x128 tiles both) {t_two:6.1f} us | shipped plan (two launches) {t_ship:6.1f} us | ONE launch, chained '
              f'{t_chain:6.1f} us | one launch without the waits (wrong results: what is left is packing) {t_free:6.1f} us')
        # timeline of the last chained launch (100 MHz ticks -> us, relative to the earliest block start)
        st = stamps.view(blocks, MAXU, 4).cpu().double()
        t0 = st[:, 0, 0].min()
        n1 = (M // 128) * (N1 // 128)
        rows = []
        for b in range(blocks):
            for k in range(MAXU):
                if st[b, k, 2] > 0:
                    rows.append((int(st[b, k, 3]), (st[b, k, 0] - t0) / 100, (st[b, k, 1] - t0) / 100, (st[b, k, 2] - t0) / 100))
        f1 = [r for r in rows if r[0] < n1]
        f2 = [r for r in rows if r[0] >= n1]
        if f1 and f2:
            import statistics as stt
            print('   timeline (us from the first block\'s start): fc1 tiles end %.1f .. %.1f (median %.1f); fc2 units: start %.1f .. %.1f, '
                  'wait for the dependency median %.2f max %.2f, body median %.1f, launch ends %.1f' % (
                      min(r[3] for r in f1), max(r[3] for r in f1), stt.median(r[3] for r in f1),
                      min(r[1] for r in f2), max(r[1] for r in f2), stt.median(r[2] - r[1] for r in f2), max(r[2] - r[1] for r in f2),
                      stt.median(r[3] - r[2] for r in f2), max(r[3] for r in rows)))
            # the seam as a unit sees it: time from the LAST of its four producers' publish to its own first instruction behind the wait
            pub = {r[0]: r[3] for r in f1}
            seam = []
            for r in f2:
                v = r[0] - n1
                bs = v // (N2 // 128)
                s_, band = bs % S, bs // S
                prod = [pub.get(band * (N1 // 128) + j) for j in range(4 * s_, 4 * s_ + 4)]
                if all(p is not None for p in prod) and r[1] <= max(prod):        # it really waited
                    seam.append(r[2] - max(prod))
            if seam:
                print('   hand-off latency (last producer published -> consumer past its acquire), units that waited: n %d median %.2f us max %.2f us'
                      % (len(seam), stt.median(seam), max(seam)))


if __name__ == '__main__':
    main()
