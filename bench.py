"""bench.py -- pretraining throughput of the Point-DAE hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...        # N > 1 without a launcher: starts the N ranks itself (child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` is binding: the run refuses (exit != 0) when fewer than N GPUs are visible or when a
launcher's WORLD_SIZE differs from N, instead of reporting an N-GPU number from fewer ranks.

A "step" is one full optimisation step (FPS -> kNN grouping -> in-forward
corruption -> patch embedder -> masked Transformer encoder/decoder -> Chamfer
loss -> backward -> AdamW [-> gradient all-reduce when N > 1]) over one batch of
synthetic ShapeNet-shaped clouds already resident in HBM.

Workload (config.workload): BASELINE.json configs[2]/[3] -- the Transformer
experiment `..._maskpatch_p0005_whole.yaml`, local B=128, N=1024, G=64, k=32 --
because the metric is quoted "(N=1024,G=64) at 1/2/4/8 MI355X": its N=1 point
is cfg3 and its N>1 points are cfg4 (local B=128 per GPU, weak scaling).
`--workload cfg2` times the PointNet++ experiment instead (when built).

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
`roofline` for the dominant kernel (timed live with HIP events on the launch
stream), `roofline_geometry` (FPS / kNN / Chamfer launch times against the HBM
and the vector-op rooflines), `also.cfg2` (BASELINE config 2 as a second, short
timed leg with the roofline of its large Chamfer kernel) and `cpu_baseline`, the
CPU oracle (oracle/, "port") timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG3 = 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'
CFG2 = 'cfgs/pretrain_PointCAE_affine_r3_dropout_local_4xlonger.yaml'
CFG5 = 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_double.yaml'
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA (v_mfma_f32_32x32x2_f32), dense
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (AMD's 5 PF headline includes 2:1 sparsity)
# the row-GEMM family's default arithmetic: fp32 operands split exactly into three bf16 terms, SIX bf16 products per
# fp32 product (include/pdae.h PDAE_GEMM_BF16X3) => its matrix-pipe ceiling in fp32-equivalent FLOP/s
BF16X3_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
HBM_PEAK_GBS = 8000.0
VALU_F32_PEAK_TOPS = 78.6         # fp32 vector peak counted WITHOUT fused multiply-add (157.3 / 2): the geometry
                                  # kernels are sub / mul / add / min chains
EXP_PEAK_TOPS = 9.8               # v_exp_f32 per second (T/s): the transcendental rate VERDICT r3 prices EMD against
ROUND = 6                         # profiles/*_rNN.json this bench refers to


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--workload', default='cfg3', choices=['cfg3', 'cfg2'])
    p.add_argument('--model-name', default=None,
                   help='override model.NAME of the cfg3 YAML (e.g. PointCAE_transformer_fc_global_folding_local, the published runs)')
    p.add_argument('--batch', type=int, default=128, help='clouds per GPU')
    p.add_argument('--npoints', type=int, default=1024)
    p.add_argument('--num_group', type=int, default=64)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--eager', action='store_true', help='launch kernel by kernel instead of replaying hipGraphs')
    p.add_argument('--cpu-batch', type=int, default=8)
    p.add_argument('--cpu-steps', type=int, default=1000, help='upper bound; the sample stops after ~20 s')
    p.add_argument('--probe-steps', type=int, default=5, help='eager steps after the timed region that time the roofline kernel')
    p.add_argument('--split', default='auto', choices=['auto', 'on', 'off'],
                   help='two-phase graphed step with the all-reduce under the embedder backward (auto: when ranks > 1)')
    p.add_argument('--no-also', action='store_true', help='skip the geometry-kernel rooflines and the cfg2 leg')
    p.add_argument('--also-steps', type=int, default=6, help='timed steps of the cfg2 and published-variant legs')
    p.add_argument('--cpu-worker', type=int, default=-1,
                   help='(internal) run the cpu_baseline sample in this process pinned to logical CPUs [16 k, 16 k + 16) and print it')
    p.add_argument('--no-calibration', action='store_true', help='skip the box calibration block and the three extra timed regions')
    p.add_argument('--no-tvis-table', action='store_true',
                   help='skip the per-visible-token-count replay timings and the straggler model built on them')
    return p.parse_args()


def _event_time_us(fn, iters=20, warm=3):
    """Average launch duration with HIP events on the launch stream (torch's current stream)."""
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def geometry_rooflines(args, clouds, shape=None):
    """FPS / kNN+group / Chamfer forward+backward at the workload's shapes: average launch time (HIP
    events on the launch stream, back-to-back launches) against BOTH rooflines north_star names --
    algorithmic bytes over the 8 TB/s HBM peak, and pair operations (9 per point pair:
    3 sub, 3 mul, 2 add, 1 min/compare) over the 78.6 T op/s non-FMA fp32 vector peak.  These kernels
    move 26 MB per step in total; they are latency / VALU bound by construction (63 dependent FPS
    iterations), which is what the two fractions show."""
    from point_dae_amd import _lib
    B, N, G, k = args.batch, args.npoints, args.num_group, 32
    if shape is not None:
        B, N, G = shape
    x = clouds[:B].contiguous()
    dev = x.device
    idx = torch.empty(B, G, dtype=torch.int32, device=dev)
    ctr = torch.empty(B, G, 3, device=dev)
    nn_idx = torch.empty(B, G, k, dtype=torch.int64, device=dev)
    dist_ = torch.empty(B, G, k, device=dev)
    nbr = torch.empty(B, G, k, 3, device=dev)
    out = []

    def row(name, us, nbytes, pairs):
        out.append({'kernel': name, 'avg_us': us, 'algorithmic_bytes': nbytes,
                    'hbm_frac': nbytes / (us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                    'pair_ops': pairs * 9, 'valu_frac': pairs * 9 / (us * 1e-6) / (VALU_F32_PEAK_TOPS * 1e12)})
    us = _event_time_us(lambda: _lib.call('pdae_furthest_point_sampling', x, B, N, G, _lib.ptr(x), _lib.ptr(idx), _lib.ptr(ctr)))
    row('fps_kernel (B=%d, %d->%d, + centre gather)' % (B, N, G), us, B * (12 * N + 4 * G + 12 * G), B * (G - 1) * N)
    us = _event_time_us(lambda: _lib.call('pdae_knn', x, B, N, G, k, _lib.ptr(x), _lib.ptr(ctr), _lib.ptr(nn_idx),
                                          _lib.ptr(dist_), _lib.ptr(nbr)))
    row('knn_kernel (B=%d, %dx%d -> %d, idx + dist + centred patches)' % (B, G, N, k), us,
        B * (12 * N + 12 * G + 8 * G * k + 4 * G * k + 12 * G * k), B * G * N)
    if shape is not None:                                    # (a second shape: the two kernels whose cost depends on N, G)
        return out
    P = B * int(0.65 * G)                                    # masked patches at the mean ratio
    a = nbr.reshape(-1, k, 3)[:P].contiguous()
    b = torch.roll(a, 1, 0).contiguous()
    d1, d2 = torch.empty(P, k, device=dev), torch.empty(P, k, device=dev)
    i1, i2 = torch.empty(P, k, dtype=torch.int32, device=dev), torch.empty(P, k, dtype=torch.int32, device=dev)
    us = _event_time_us(lambda: _lib.call('pdae_chamfer_forward', a, P, k, _lib.ptr(a), k, _lib.ptr(b), _lib.ptr(d1),
                                          _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2)))
    row('chamfer_fwd_packed (%d patches of %dx%d)' % (P, k, k), us, P * (12 * 2 * k + 8 * 2 * k), P * 2 * k * k)
    g1, g2 = torch.ones(P, k, device=dev), torch.ones(P, k, device=dev)
    ga, gb = torch.empty_like(a), torch.empty_like(b)
    us = _event_time_us(lambda: _lib.call('pdae_chamfer_backward', a, P, k, _lib.ptr(a), k, _lib.ptr(b), _lib.ptr(i1),
                                          _lib.ptr(i2), _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(ga), _lib.ptr(gb)))
    row('chamfer_bwd_packed (%d patches of %dx%d)' % (P, k, k), us, P * (12 * 2 * k + 4 * 2 * k + 4 * 2 * k + 12 * 2 * k), P * 2 * k)
    # EMD (extensions/emd: the north star names it; no model of the reference calls it): approxmatch = 10 levels x 3
    # phases over every point pair, one v_exp_f32 and ~13 other vector operations per pair and phase.  Ceilings: the
    # exp rate (EXP_PEAK: quarter-rate transcendental unit) and the fp32 vector peak.
    from point_dae_amd import emd

    def emd_row(name, a, b):
        Bp, n, m = a.shape[0], a.shape[1], b.shape[1]
        us = _event_time_us(lambda: emd.approxmatch_forward(a, b), iters=10)
        pairs = Bp * n * m * 30
        match = emd.approxmatch_forward(a, b)
        us_cost = _event_time_us(lambda: emd.matchcost_forward(a, b, match), iters=10)
        out.append({'kernel': name, 'avg_us': us, 'algorithmic_bytes': Bp * (12 * (n + m) + 4 * n * m),
                    'hbm_frac': Bp * (12 * (n + m) + 4 * n * m) / (us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                    'pair_ops': pairs * 14, 'valu_frac': pairs * 14 / (us * 1e-6) / (VALU_F32_PEAK_TOPS * 1e12),
                    'exp_per_s': pairs / (us * 1e-6), 'exp_frac': pairs / (us * 1e-6) / EXP_PEAK_TOPS / 1e12,
                    'matchcost_avg_us': us_cost})
    emd_row('emd approxmatch, one wave per pair (%d patches of %dx%d)' % (P, k, k), a, b)
    big = clouds[:16].contiguous()
    emd_row('emd approxmatch, one launch per phase (8 clouds of %dx%d)' % (N, N), big[:8].contiguous(), big[8:16].contiguous())
    return out


def box_calibration(device):
    """Two kernels of known work, ~1 s in all, so that a round's delta can be read against the box it was measured on
    (MI355X boxes of this pool differ by +-4 % on the step, more on matrix loops: the clock a device holds under load):
    a register-only bf16 MFMA loop on pseudo-random operands (one wave per SIMD, every CU; csrc/calib.hip) with the shader
    clock it held, and a 16 B-per-lane streaming copy of 1 GiB (the microarchitecture guide measures 6.29 TB/s)."""
    import ctypes
    from point_dae_amd import _lib
    out = {}
    try:
        blocks, iters = 256, 60000
        sink = torch.zeros(4, device=device)
        clk = torch.zeros(blocks, 2, dtype=torch.int64, device=device)
        flops = ctypes.c_double(0.0)
        run = lambda it: _lib.call('pdae_calib_mfma_bf16', sink, blocks, it, _lib.ptr(sink), _lib.ptr(clk), ctypes.byref(flops))
        run(2000)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run(iters)
            e.record()
            torch.cuda.synchronize()
            ms.append(s.elapsed_time(e))
        c = clk.double()
        ghz = (c[:, 0] / c[:, 1].clamp(min=1)).median().item() * 0.1
        best = min(ms)
        out['mfma_bf16'] = {'tflops': flops.value / best / 1e9, 'ms': best, 'ms_all': ms, 'shader_clock_ghz': ghz,
                            'frac_of_dense_peak': flops.value / best / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                            'what': '256 blocks x 4 waves x %d x 32 v_mfma_f32_32x32x16_bf16, register operands with fresh '
                                    'mantissa bits every iteration; clock = s_memtime / s_memrealtime, median over blocks' % iters}
        n = 1 << 30
        src = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 255)
        dst = torch.empty_like(src)
        cp = lambda: _lib.call('pdae_calib_copy', src, n, _lib.ptr(src), _lib.ptr(dst))
        cp()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            cp()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 5 * 1e3
        out['copy_f4'] = {'gbs': 2.0 * n / (us * 1e-6) / 1e9, 'avg_us': us, 'bytes_copied': n,
                          'frac_of_hbm_peak': 2.0 * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          'what': 'dst = src, 1 GiB, 16 B per lane, grid-stride; read + write bytes over the launch time'}
        del src, dst
    except Exception as err:                                   # a measurement aid: never fails the bench line
        out['error'] = repr(err)[:200]
    return out


def cfg2_geometry(device, x):
    """the geometry kernels of Point_CAE_PointNetv2's encoder at its shapes (B=128) -- FPS
    1024 -> 512 and 512 -> 128, ball query (512 x 32, r 0.2 on 1024 points; 128 x 64, r 0.4 on 512), the grouping
    (sa_group_rows, both levels) and the coarse Chamfer loss (1024 x 1024) -- as average launch time (HIP events on the
    launch stream, back to back) against the HBM roofline (algorithmic bytes / 8 TB/s) and the non-FMA fp32 vector
    roofline (9 operations per point pair / 78.6 T op/s; for the ball query the pair count is the full scan, an upper
    bound: a lane stops once its ball is full)."""
    from point_dae_amd import _lib
    B, N = x.shape[0], x.shape[1]
    out = []

    def row(name, us, nbytes, pairs, note=None):
        r = {'kernel': name, 'avg_us': us, 'algorithmic_bytes': nbytes,
             'hbm_frac': nbytes / (us * 1e-6) / (HBM_PEAK_GBS * 1e9),
             'pair_ops': pairs * 9, 'valu_frac': pairs * 9 / (us * 1e-6) / (VALU_F32_PEAK_TOPS * 1e12)}
        if note:
            r['note'] = note
        out.append(r)
    x = x.contiguous()
    levels, src = [], x
    for (np_, ns, radius) in ((512, 32, 0.2), (128, 64, 0.4)):
        n = src.shape[1]
        idx = torch.empty(B, np_, dtype=torch.int32, device=device)
        ctr = torch.empty(B, np_, 3, device=device)
        us = _event_time_us(lambda: _lib.call('pdae_furthest_point_sampling', src, B, n, np_, _lib.ptr(src), _lib.ptr(idx), _lib.ptr(ctr)))
        row('fps_kernel (B=%d, %d->%d, + centre gather)' % (B, n, np_), us, B * (12 * n + 4 * np_ + 12 * np_), B * (np_ - 1) * n)
        bq = torch.empty(B, np_, ns, dtype=torch.int32, device=device)
        us = _event_time_us(lambda: _lib.call('pdae_ball_query', src, B, n, np_, float(radius), ns, _lib.ptr(ctr), _lib.ptr(src), _lib.ptr(bq)))
        row('ball_query (B=%d, %d centres x %d samples, r %.1f, %d points)' % (B, np_, ns, radius, n), us,
            B * (12 * n + 12 * np_ + 4 * np_ * ns), B * np_ * n, 'pair count = full scan (upper bound)')
        levels.append((src, ctr, bq, n, np_, ns))
        src = ctr
    for (pts, ctr, bq, n, np_, ns), C in zip(levels, (0, 128)):
        feats = torch.randn(B * n, C, device=device) if C else None
        rows = torch.empty(B * np_ * ns, 4 + C, device=device)
        us = _event_time_us(lambda: _lib.call('pdae_sa_group_rows', pts, B, n, np_, ns, C, _lib.ptr(pts), _lib.ptr(ctr), _lib.ptr(bq),
                                              _lib.ptr(feats), _lib.ptr(rows)), iters=10)
        row('sa_group_rows (B=%d, %d x %d rows of 4+%d from %d points)' % (B, np_, ns, C, n), us,
            B * (12 * n + 12 * np_ + 4 * np_ * ns + 4 * n * C + 4 * np_ * ns * (4 + C)), 0)
        if C:
            dfeat = torch.empty(B * n, C, device=device)
            us = _event_time_us(lambda: _lib.call('pdae_sa_group_rows_grad', rows, B, n, np_, ns, C, _lib.ptr(bq), _lib.ptr(rows),
                                                  _lib.ptr(dfeat)), iters=10)
            row('sa_group_rows_grad (B=%d, %d x %d rows of 4+%d onto %d points)' % (B, np_, ns, C, n), us,
                B * (4 * np_ * ns + 4 * np_ * ns * (4 + C) + 4 * n * C), 0)
    a, b = x, torch.roll(x, 1, 0).contiguous()
    d1, d2 = torch.empty(B, N, device=device), torch.empty(B, N, device=device)
    i1, i2 = torch.empty(B, N, dtype=torch.int32, device=device), torch.empty(B, N, dtype=torch.int32, device=device)
    us = _event_time_us(lambda: _lib.call('pdae_chamfer_forward', a, B, N, _lib.ptr(a), N, _lib.ptr(b), _lib.ptr(d1), _lib.ptr(d2),
                                          _lib.ptr(i1), _lib.ptr(i2)), iters=10)
    row('chamfer forward (B=%d, %d x %d: the coarse loss)' % (B, N, N), us, B * (12 * 2 * N + 8 * 2 * N), B * 2 * N * N)
    g1, g2 = torch.ones(B, N, device=device), torch.ones(B, N, device=device)
    ga, gb = torch.empty_like(a), torch.empty_like(b)
    us = _event_time_us(lambda: _lib.call('pdae_chamfer_backward', a, B, N, _lib.ptr(a), N, _lib.ptr(b), _lib.ptr(i1), _lib.ptr(i2),
                                          _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(ga), _lib.ptr(gb)), iters=10)
    row('chamfer backward (B=%d, %d x %d)' % (B, N, N), us, B * (12 * 2 * N + 4 * 2 * N + 4 * 2 * N + 12 * 2 * N), B * 2 * N)
    return out


def tvis_table(step, batch, replays=6):
    """ms per replay of every captured step graph, keyed by its visible-token count T_vis (HIP events on the launch
    stream, `replays` replays back to back after one warm replay).  The mask ratio is drawn per batch AND per rank
    (seed + rank, main.py:81 of the reference), so in a data-parallel job every step waits for the rank that drew the
    most visible tokens: this table is what bounds that straggler effect."""
    out = {}
    step.pts.copy_(batch)
    for tvis in sorted(step.graphs):
        g = step.graphs[tvis]
        gs = g if isinstance(g, tuple) else (g,)
        for x in gs:
            x.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(replays):
            for x in gs:
                x.replay()
        e.record()
        torch.cuda.synchronize()
        out[tvis] = s.elapsed_time(e) / replays
    return out


def per_tvis_families(step, batch, nn_ops, want_replay=True):
    """Deterministic per-T_vis tables of the row-GEMM families: for every captured visible-token count ONE eager run of
    the graphed step's own body with exactly that many visible tokens (graph_step._draw(tvis)) under the launch-site
    probe -> algorithmic FLOPs and bytes per family (functions of the shapes only: the same for every run), and the
    families' device time from their launches captured into one hipGraph per family and replayed back to back.
    -> {tvis: {family: {'gflop', 'gbytes', 'launches', 'ms'}}}"""
    out = {}
    step.pts.copy_(batch)
    for tvis in sorted(step.graphs):
        probe = nn_ops.Probe()
        probe.keep_calls = want_replay
        nn_ops.set_probe(probe)
        try:
            step._fwd_bwd(step._draw(tvis))
            step.model.zero_grad()
            torch.cuda.synchronize()
        finally:
            nn_ops.set_probe(None)
        fam = probe.family_summary()
        rep = {}
        if want_replay:
            probe.keep_calls = False
            try:
                rep = probe.family_replay_ms(replays=6)
            except Exception as err:                     # a measurement aid: never fails the bench line
                rep = {'error': repr(err)[:200]}
        probe.calls = {}
        out[tvis] = {k: {'gflop': r['flops'] / 1e9, 'gbytes': r['bytes'] / 1e9, 'launches': r['launches'],
                         'ms': rep.get(k) if isinstance(rep.get(k), float) else None} for k, r in fam.items()}
    return out


def tvis_distribution(G):
    """P(T_vis) under MaskTransformer._mask_center_rand (:395-422): ratio ~ U(0.5, 0.8), num_mask = int(ratio G)."""
    lo, hi = 0.5, 0.8
    p = {}
    for m in range(int(lo * G), int(hi * G) + 1):
        a, b = max(lo, m / G), min(hi, (m + 1) / G)
        if b > a:
            p[G - m] = (b - a) / (hi - lo)
    return p


def ddp_model(table, G, ms_per_step, ranks=(2, 4, 8), allreduce_mb=116.0):
    """Straggler bound on data-parallel scaling from the T_vis table: per step every rank replays the graph of ITS
    draw and the collective waits for the slowest.  E[max of n draws] from the exact distribution of T_vis;
    `other_ms` = the measured step minus the expected replay time (AdamW, host gaps) is added unchanged.
    The all-reduce itself is modelled separately: the early slice overlaps the embedder backward, ~2 MB
    are exposed.  A MODEL, not a measurement: no multi-GPU node was available to this build."""
    p = tvis_distribution(G)
    ts = sorted(t for t in p if t in table)
    if not ts:
        return None
    z = sum(p[t] for t in ts)
    mean = sum(p[t] * table[t] for t in ts) / z
    other = max(ms_per_step - mean, 0.0)
    order = sorted(ts, key=lambda t: table[t])             # by replay time
    out = {'p_tvis': {str(t): round(p[t] / z, 5) for t in ts}, 'expected_replay_ms': mean, 'other_ms': other,
           'slowest_replay_ms': max(table[t] for t in ts), 'fastest_replay_ms': min(table[t] for t in ts),
           'note': 'straggler bound only (every rank draws its own mask ratio); all-reduce of %.0f MB fp32 not included: '
                   'its early slice runs under the embedder backward, see DESIGN.md 6' % allreduce_mb,
           'predicted': {}}
    for n in ranks:
        cdf, emax = 0.0, 0.0
        for t in order:
            c2 = cdf + p[t] / z
            emax += (c2 ** n - cdf ** n) * table[t]
            cdf = c2
        out['predicted'][str(n)] = {'expected_max_replay_ms': emax, 'step_ms': emax + other,
                                    'efficiency': (mean + other) / (emax + other),
                                    'speedup': n * (mean + other) / (emax + other)}
    return out


def dominant_roofline(per_tvis, G, bf16x3):
    """The bench line's `roofline`: the row GEMM family (every Linear / 1x1 conv forward and data gradient) and the
    grouped weight gradients (+ their reductions) -- the kernels that own ~65 % of the step -- as ONE time-weighted
    figure.  Work and time are EXPECTATIONS over the mask-ratio distribution (tvis_distribution): per visible-token
    count the families' algorithmic FLOPs / bytes at the launch sites (deterministic) and their device time (one step's
    launches of a family captured into a hipGraph and replayed back to back), weighted with P(T_vis).  Peak: the
    arithmetic these kernels run -- exact-split bf16 (six bf16 MFMA products per fp32 product: bf16 dense / 6) by
    default, the fp32-input MFMA peak under PDAE_GEMM=f32mfma; the fraction of the fp32-input MFMA peak alongside."""
    if not per_tvis:
        return None
    p = tvis_distribution(G)
    ts = [t for t in sorted(per_tvis) if t in p]
    fams = ('rows_gemm', 'rows_wgrad')
    if not ts or any(per_tvis[t].get(f, {}).get('ms') is None for t in ts for f in fams):
        return None
    z = sum(p[t] for t in ts)
    ex = lambda f, key: sum(p[t] * per_tvis[t][f][key] for t in ts) / z
    rows, gflop, ms, gbytes = {}, 0.0, 0.0, 0.0
    for f in fams:
        rows[f] = {'launches_per_step': ex(f, 'launches'), 'gflop_per_step': ex(f, 'gflop'), 'ms_per_step': ex(f, 'ms'),
                   'algorithmic_gbytes_per_step': ex(f, 'gbytes'), 'achieved': ex(f, 'gflop') / ex(f, 'ms')}
        gflop += rows[f]['gflop_per_step']
        ms += rows[f]['ms_per_step']
        gbytes += rows[f]['algorithmic_gbytes_per_step']
    peak = BF16X3_PEAK_TFLOPS if bf16x3 else MFMA_F32_PEAK_TFLOPS
    ach = gflop / ms                                  # GFLOP / ms = TFLOP/s
    traffic = traffic_src = None
    for rnd in range(ROUND, 0, -1):
        pmc = os.path.join(ROOT, 'profiles', 'pmc_r%02d.json' % rnd)
        if os.path.exists(pmc):
            rec = json.load(open(pmc)).get('rows_families_hbm_bytes_per_step')
            if rec:
                traffic, traffic_src = rec, ('profiles/pmc_r%02d.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950 '
                                             'FETCH correction, bytes of the two families per STEP averaged over that run\'s mask draws; '
                                             'fabric-side counters: Infinity-Cache hits included)' % rnd)
                break
    return {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
            'traffic': traffic, 'traffic_source': traffic_src, 'traffic_measured_in_this_run': False,
            # the expectation runs over the captured visible-token counts: the default run captures the whole support of
            # P(T_vis) (mass 1.0); a run that covers less than 0.95 of it is flagged, its figure is not comparable
            'covered_probability_mass': z, 'comparable': bool(z >= 0.95),
            'algorithmic_bytes': gbytes * 1e9,
            'algorithmic_bytes_note': 'per step: every operand and every result of every launch of the two families once, 4 B per '
                                      'element (M K + N K + slabs M N per product, M (N + K) + N K per weight gradient: DESIGN.md 4b)',
            'kernel': 'pdae::rows3::gemm3_kernel + pdae::rows3::wgrad3t_kernel (+ wgrad_reduce_kernel): the row-GEMM family' if bf16x3
                      else 'pdae::rows::rows_gemm_kernel + pdae::rows::wgrad_kernel (+ wgrad_reduce_kernel): the row-GEMM family',
            'ms_per_step': ms, 'gflop_per_step': gflop,
            'frac_of_f32_mfma_peak': ach / MFMA_F32_PEAK_TFLOPS, 'families': rows,
            'timing': 'per visible-token count: one step\'s launches of each family (the graphed step\'s own body) captured into a '
                      'hipGraph per family and replayed back to back, HIP events around 6 replays; expectation over P(T_vis); '
                      'profiles/kernel_summary_r%02d.txt has the same kernels\' durations inside the step\'s replays' % ROUND,
            'expectation_over': {str(t): round(p[t] / z, 5) for t in ts},
            'peak_note': ('fp32-equivalent ceiling of the exact-split arithmetic: bf16 MFMA dense peak %.0f TFLOP/s / 6 products '
                          '(MI355X_MICROARCH.md) at the 2.4 GHz peak clock; dense bf16 loops hold 1.6-1.9 GHz on this part (in-kernel '
                          's_memtime / s_memrealtime, tools/lab/p3_clock.py), i.e. a sustained ceiling of ~280-330; the fp32-input '
                          'MFMA peak is %.1f' % (MFMA_BF16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS))
                         if bf16x3 else 'fp32-input MFMA (v_mfma_f32_32x32x2_f32) dense peak, MI355X_MICROARCH.md'}


def cfg2_leg(args, device, rank):
    """BASELINE config 2 (Point_CAE_PointNetv2, B=128, N=1024: FPS, ball query, grouping, the 1024^2 and
    16384x1024 Chamfer losses) as a second, shorter timed leg of the same run -> {value, ms_per_step,
    roofline of chamfer_fwd_tiled on the 16384x1024 fine loss (pair rate vs the vector peak)}."""
    from point_dae_amd import _lib, builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedStaticStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(ROOT, CFG2))
    model = FlatDataParallel(builder.model_builder(config.model).to(device), broadcast=False, process_group=None)
    model.world_size = 1                         # a local leg on rank 0: no collective
    optimizer, _ = builder.build_opti_sche(model, config)
    model.train()
    model.zero_grad()
    B, N = 128, 1024
    x = torch.from_numpy(shapenet_like_clouds(2 * B, N, seed=300 + rank)).to(device)
    gstep = GraphedStaticStep(model, optimizer, lambda a, b: a + float(config.normal_weight) * b * 0.5, B, N)
    for i in range(4):
        gstep(x[:B], x[B:])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.also_steps):
        gstep(x[:B], x[B:])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the one large geometry kernel of this config: Chamfer of the 16384-point fine cloud against the 1024-point input
    fine, gt = torch.randn(B, 16384, 3, device=device), x[:B].contiguous()
    d1, d2 = torch.empty(B, 16384, device=device), torch.empty(B, N, device=device)
    i1 = torch.empty(B, 16384, dtype=torch.int32, device=device)
    i2 = torch.empty(B, N, dtype=torch.int32, device=device)
    us = _event_time_us(lambda: _lib.call('pdae_chamfer_forward', fine, B, 16384, _lib.ptr(fine), N, _lib.ptr(gt),
                                          _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2)), iters=5, warm=2)
    pairs = B * 2 * 16384 * N
    # the leg's dominant kernels are GEMMs too (the FoldingNet stage's 2.1 M-row products and their weight gradients): one
    # eager pass of the step body under the launch-site probe, the families' launches replayed as one hipGraph each
    from point_dae_amd import nn_ops
    gemm_roof = None
    try:
        probe = nn_ops.Probe()
        probe.keep_calls = True
        nn_ops.set_probe(probe)
        gstep._fwd_bwd()
        torch.cuda.synchronize()
        nn_ops.set_probe(None)
        probe.keep_calls = False
        fam = probe.family_summary()
        rep = probe.family_replay_ms(replays=3)
        probe.calls = {}
        keys = [k for k in ('rows_gemm', 'rows_wgrad') if k in fam and k in rep]
        gf, ms = sum(fam[k]['flops'] for k in keys) / 1e9, sum(rep[k] for k in keys)
        gemm_roof = {'kernel': 'rows3::gemm3_kernel + rows3::wgrad3t_kernel (row-GEMM family of the cfg2 step)', 'bound': 'mfma',
                     'gflop_per_step': gf, 'ms_per_step': ms, 'achieved': gf / ms, 'peak': BF16X3_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': gf / ms / BF16X3_PEAK_TFLOPS, 'frac_of_f32_mfma_peak': gf / ms / MFMA_F32_PEAK_TFLOPS,
                     'algorithmic_bytes': sum(fam[k]['bytes'] for k in keys),
                     'families': {k: {'gflop': fam[k]['flops'] / 1e9, 'ms': rep[k], 'launches': fam[k]['launches']} for k in keys}}
    except Exception as err:                                   # a measurement aid: never fails the bench line
        nn_ops.set_probe(None)
        gemm_roof = {'error': repr(err)[:200]}
    model.zero_grad()
    del model, optimizer, gstep
    try:
        geo = cfg2_geometry(device, x[:B])
    except Exception as err:                                   # a measurement aid: never fails the bench line
        geo = {'error': repr(err)[:200]}
    return {'workload': 'cfg2: pretrain_PointCAE_affine_r3_dropout_local_4xlonger.yaml (Point_CAE_PointNetv2), '
                        'B=128, N=1024, full train step, hipGraph replay',
            'value': B * args.also_steps / dt, 'unit': 'clouds/s', 'ms_per_step': dt / args.also_steps * 1e3,
            'steps': args.also_steps,
            'roofline_gemm': gemm_roof,
            'roofline_geometry': geo,
            'roofline': {'kernel': 'chamfer_fwd_tiled (128 x 16384 x 1024)', 'bound': 'valu', 'avg_us': us,
                         'achieved': pairs * 9 / (us * 1e-6) / 1e12, 'peak': VALU_F32_PEAK_TOPS, 'unit': 'T op/s',
                         'frac': pairs * 9 / (us * 1e-6) / (VALU_F32_PEAK_TOPS * 1e12),
                         'hbm_frac': B * (12 * (16384 + N) + 8 * (16384 + N)) / (us * 1e-6) / (HBM_PEAK_GBS * 1e9)}}


def cfg5_leg(args, device, rank):
    """BASELINE config 5's per-GPU workload (`..._p0005_double.yaml` at N=2048, G=128, local B=32 of the global 256 over
    8 GPUs: the large-cloud stress of the LDS tiling) as a short timed leg: the same graphed train step, G = 128 patch
    tokens per cloud (T_vis 26..64).  The host RNG is re-seeded before the warm-up and before the timed steps, so the timed
    steps draw the mask ratios the warm-up already captured graphs for: replays only inside the timed region."""
    import random
    import numpy as np
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(ROOT, CFG5))
    B, N, G = 32, 2048, 128
    config.npoints, config.model.num_group = N, G
    model = FlatDataParallel(builder.model_builder(config.model).to(device), broadcast=False, process_group=None)
    model.world_size = 1                         # a local leg on rank 0: no collective
    optimizer, _ = builder.build_opti_sche(model, config)
    model.train()
    model.zero_grad()
    x = torch.from_numpy(shapenet_like_clouds(2 * B, N, seed=900 + rank)).to(device).split(B)
    step = GraphedTrainStep(model, optimizer, config, B, N, split=False, warmup_eager=1)

    def seed(v):
        random.seed(v), np.random.seed(v), torch.manual_seed(v)
    n = args.also_steps
    seed(77)
    for i in range(n + 1):                       # one eager step, then a capture + replay per new mask ratio
        step(x[i % 2])
    seed(77)
    step(x[0])                                   # (the draw the eager step consumed: captured now)
    for i in range(1, n + 1):
        step(x[i % 2])
    torch.cuda.synchronize()
    seed(77)
    tv = []
    t0 = time.perf_counter()
    for i in range(n + 1):
        out = step(x[i % 2])
        tv.append(step.last_tvis)
    loss = out[0].detach().clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    geo = geometry_rooflines(args, torch.cat(x), shape=(B, N, G))
    del model, optimizer, step
    return {'workload': 'cfg5 per-GPU shape: ..._maskpatch_p0005_double.yaml at N=2048, G=128, k=32, local B=32 (global 256 over '
                        '8 GPUs), full train step, hipGraph replay',
            'value': B * (n + 1) / dt, 'unit': 'clouds/s', 'ms_per_step': dt / (n + 1) * 1e3, 'steps': n + 1,
            'tvis_of_the_timed_steps': tv, 'loss_last_step': float(loss), 'roofline_geometry': geo}


def published_leg(args, device, rank):
    """The published Transformer variant (`--model_name PointCAE_transformer_fc_global_folding_local`, SURVEY F5 /
    row a13: global feature -> FC coarse + two FoldingNet stages per masked token, two Chamfer losses) on the
    cfg3 YAML, B=128, as a short timed leg -> {value, ms_per_step}."""
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(ROOT, CFG3))
    config.model.NAME = 'PointCAE_transformer_fc_global_folding_local'
    model = FlatDataParallel(builder.model_builder(config.model).to(device), broadcast=False, process_group=None)
    model.world_size = 1                         # a local leg on rank 0: no collective
    optimizer, _ = builder.build_opti_sche(model, config)
    model.train()
    model.zero_grad()
    B, N, G = 128, 1024, config.model.num_group
    x = torch.from_numpy(shapenet_like_clouds(2 * B, N, seed=500 + rank)).to(device).split(B)
    step = GraphedTrainStep(model, optimizer, config, B, N, split=False)
    for tvis in range(G - int(0.8 * G), G - int(0.5 * G) + 1):
        step.pts.copy_(x[0])
        step._draw()
        if tvis not in step.graphs:
            step._capture(tvis)
    step.eager_left = 0
    model.zero_grad()
    for i in range(4):
        step(x[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.also_steps):
        out = step(x[i % 2])
    loss = out[0].detach().clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del model, optimizer, step
    return {'workload': 'cfg3 YAML with --model_name PointCAE_transformer_fc_global_folding_local (the published '
                        'runs\' model), B=128, N=1024, G=64, full train step, hipGraph replay',
            'value': B * args.also_steps / dt, 'unit': 'clouds/s', 'ms_per_step': dt / args.also_steps * 1e3,
            'steps': args.also_steps, 'loss_last_step': float(loss)}


def dgcnn_leg(args, device, rank):
    """`--model_name Point_CAE_DGCNN_FCOnly` (the published non-Transformer runs' model, rerun.sh:37-40: total_bs 256
    over 8 GPUs = 32 clouds of 1024 points per GPU) on the cfg2 YAML as a short timed leg -> {value, ms_per_step}."""
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedStaticStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(ROOT, CFG2))
    config.model.NAME = 'Point_CAE_DGCNN_FCOnly'
    model = FlatDataParallel(builder.model_builder(config.model).to(device), broadcast=False, process_group=None)
    model.world_size = 1                         # a local leg on rank 0: no collective
    optimizer, _ = builder.build_opti_sche(model, config)
    model.train()
    model.zero_grad()
    B, N = 32, 1024
    x = torch.from_numpy(shapenet_like_clouds(2 * B, N, seed=700 + rank)).to(device)
    gstep = GraphedStaticStep(model, optimizer, lambda a, b: a + b, B, N)
    for i in range(4):
        gstep(x[:B], x[B:])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.also_steps):
        out = gstep(x[:B], x[B:])
    loss = out[0].detach().clone() if isinstance(out, (tuple, list)) else None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del model, optimizer, gstep
    return {'workload': 'cfg2 YAML with --model_name Point_CAE_DGCNN_FCOnly (rerun.sh:37-40; 256 clouds over 8 GPUs = '
                        'B=32 per GPU), N=1024, k=20, full train step, hipGraph replay',
            'value': B * args.also_steps / dt, 'unit': 'clouds/s', 'ms_per_step': dt / args.also_steps * 1e3,
            'steps': args.also_steps, 'loss_last_step': float(loss) if loss is not None else None}


def cpu_baseline(config, args):
    """The CPU oracle (plain-PyTorch model + C restatement of the native ops)
    on a bounded sample of the same workload: `cpu_steps` optimisation steps of
    B=`cpu_batch` clouds (about 10-30 s), all host cores."""
    import numpy as np
    from oracle import model as OM, ops as O
    from point_dae_amd.synthetic import shapenet_like_clouds
    # all host cores up to 16: beyond that torch's intra-op pools only contend on
    # these small GEMMs (256 threads on the GPU box ran 20x SLOWER than 16)
    cores = min(os.cpu_count() or 1, 16)
    if getattr(args, 'cpu_worker', -1) >= 0:     # one of the pinned processes of cpu_baseline_box
        os.sched_setaffinity(0, set(range(16 * args.cpu_worker, 16 * args.cpu_worker + 16)))
    torch.set_num_threads(cores)
    O.build()
    O.set_threads(cores)
    torch.manual_seed(0)
    model = OM.PointCAE_transformer(config.model).train()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith('.bias') or 'token' in n) else decay).append(p)
    opt = torch.optim.AdamW([{'params': no_decay, 'weight_decay': 0.},
                             {'params': decay, 'weight_decay': config.optimizer.kwargs.weight_decay}],
                            lr=config.optimizer.kwargs.lr)
    x = torch.from_numpy(shapenet_like_clouds(args.cpu_batch, args.npoints, seed=0))

    def step():
        loss, ln = model(x, x)
        (loss + float(config.normal_weight) * ln.sum()).backward()
        opt.step()
        opt.zero_grad()
    t0 = time.time()
    step()                                   # warm-up (allocations, thread pools)
    warm = time.time() - t0
    t0 = time.time()
    done = 0
    while done < args.cpu_steps and (done == 0 or time.time() - t0 + warm < 20.0):
        step()                               # bounded sample: about 20 s of CPU work
        done += 1
    dt = time.time() - t0
    args.cpu_steps = done
    cpu_model = None
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    cpu_model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': args.cpu_batch * args.cpu_steps / dt, 'unit': 'clouds/s', 'cores': cores,
            'cpu_model': cpu_model, 'threads': cores, 'host_logical_cpus': os.cpu_count(),
            'kind': 'port',
            'sample': '%d steps of B=%d (N=%d, G=%d, k=32) full train step on the CPU oracle '
                      '(oracle/model.py + oracle/pdae_oracle.c), %.1f s' % (
                          args.cpu_steps, args.cpu_batch, args.npoints, args.num_group, dt)}


def cpu_baseline_box(args, one):
    """The same sample on the whole box: floor(logical CPUs / 16) processes, each pinned to 16 logical CPUs and running
    16 threads (the size at which ONE process is fastest: cpu_baseline), all at the same time; their rates summed.
    `one` = the single-process figure, kept as cpu_baseline.value."""
    import subprocess
    n = (os.cpu_count() or 1) // 16
    if n < 2:
        return None
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='', OMP_NUM_THREADS='16')
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-batch', str(args.cpu_batch), '--npoints', str(args.npoints),
           '--num_group', str(args.num_group)]
    procs = [subprocess.Popen(cmd + ['--cpu-worker', str(k)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for k in range(n)]
    vals = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=180)
            vals.append(json.loads(out.strip().splitlines()[-1])['value'])
        except Exception:                                      # a worker that failed counts as zero
            pr.kill()
    if not vals:
        return None
    return {'value': sum(vals), 'unit': 'clouds/s', 'processes': len(vals), 'threads_per_process': 16, 'cores': 16 * len(vals),
            'per_process_min': min(vals), 'per_process_max': max(vals), 'kind': 'port',
            'sample': 'the cpu_baseline sample in %d processes at once, each pinned to 16 logical CPUs (of %d); rates summed; '
                      'one process alone on the idle box: %.1f clouds/s' % (len(vals), os.cpu_count(), one['value'])}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as a
    CHILD process (`python -m torch.distributed.run`, one rank per GPU over RCCL), relay rank 0's
    JSON line and return the child's exit code.  Nothing in this (parent) process touches the GPU:
    torch.cuda.device_count() does not initialise it on this image, and the child is a new process,
    never an exec of this one.  The reference's launcher is `torch.distributed.launch` around
    main.py (utils/dist_utils.py:9-29, main.py:57-59)."""
    import socket
    import subprocess
    backend = os.environ.get('PDAE_BENCH_BACKEND', 'nccl')
    have = torch.cuda.device_count()
    if backend == 'nccl' and have < args.gpus:      # RCCL needs one device per rank
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) are visible; refusing to report a '
                         '%d-GPU number from fewer devices\n' % (args.gpus, have, args.gpus))
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for ln in child.stdout:
        if ln.startswith('{"metric"'):
            lines += 1
        sys.stdout.write(ln)
        sys.stdout.flush()
    rc = child.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write('bench.py: the ranks exited 0 but printed %d result lines\n' % lines)
        return 3
    return rc


def main():
    args = parse()
    if args.cpu_worker >= 0:                       # a pinned CPU-oracle worker of cpu_baseline_box: never touches a GPU
        from point_dae_amd.config import cfg_from_yaml_file
        config = cfg_from_yaml_file(os.path.join(ROOT, CFG3))
        config.npoints, config.model.num_group = args.npoints, args.num_group
        print(json.dumps(cpu_baseline(config, args)), flush=True)
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d does not match WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path in point_dae_amd)')
    backend = os.environ.get('PDAE_BENCH_BACKEND', 'nccl')      # nccl = RCCL over xGMI (gloo: code-path test on one GPU)
    if world > 1 and backend == 'nccl' and torch.cuda.device_count() < world:
        raise SystemExit('bench.py: %d ranks but %d GPU(s): RCCL needs one device per rank' % (
            world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend)
    device = torch.device('cuda', torch.cuda.current_device())
    from point_dae_amd.graph_step import use_created_stream
    use_created_stream(device)      # one created stream for everything: NULL-stream work breaks hipGraph replays here

    from point_dae_amd import builder, nn_ops
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.misc import set_random_seed
    from point_dae_amd.graph_step import GraphedStaticStep, GraphedTrainStep
    from point_dae_amd.runner_pretrain import train_step
    from point_dae_amd.synthetic import shapenet_like_clouds

    config = cfg_from_yaml_file(os.path.join(ROOT, CFG3 if args.workload == 'cfg3' else CFG2))
    if args.model_name:
        config.model.NAME = args.model_name
    config.npoints = args.npoints
    config.model.num_group = args.num_group
    if args.workload == 'cfg2':
        args.no_cpu_baseline = True         # the cpu_baseline leg times the cfg3 oracle model
    set_random_seed(0 + rank)
    model = FlatDataParallel(builder.model_builder(config.model).to(device))
    optimizer, _ = builder.build_opti_sche(model, config)
    model.train()
    model.zero_grad()

    pool = 4
    clouds = torch.from_numpy(shapenet_like_clouds(args.batch * pool, args.npoints, seed=100 + rank)).to(device)
    batches = list(clouds.split(args.batch))

    if args.workload == 'cfg2':
        # Point_CAE_PointNetv2: corrupted input = another cloud of the pool (the reference corrupts in the
        # data loader); loss mix 'xyznormal_gradual' at mid-training (gradual weight 0.5)
        gstep = GraphedStaticStep(model, optimizer, lambda a, b: a + float(config.normal_weight) * b * 0.5,
                                  args.batch, args.npoints)

        def step(x):
            return gstep(torch.roll(x, 1, 0), x)
    elif args.eager:
        def step(x):
            return train_step(model, optimizer, config, x, x)
    else:
        step = GraphedTrainStep(model, optimizer, config, args.batch, args.npoints,
                                split=None if args.split == 'auto' else args.split == 'on')
        # capture every graph (one per visible-token count) before the warm-up
        # and timed steps; captures are set-up, not steps
        for tvis in range(args.num_group - int(0.8 * args.num_group), args.num_group - int(0.5 * args.num_group) + 1):
            step.pts.copy_(batches[0])
            step._draw()
            if tvis not in step.graphs:
                step._capture(tvis)
        step.eager_left = 0
        model.zero_grad()
        for i in range(2):                    # first replays upload the graphs: part of the set-up, whatever --warmup is
            step(batches[i % pool])
    for i in range(args.warmup):
        step(batches[i % pool])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    probe = nn_ops.Probe()
    if args.eager:
        nn_ops.set_probe(probe)
    t0 = time.perf_counter()
    loss_first = loss_last = None
    for i in range(args.steps):
        out = step(batches[i % pool])
        if i == 0:
            loss_first = out[0].detach().clone()      # (a replayed graph's output buffer is reused by later replays)
    loss_last = out[0].detach().clone()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # three more timed regions of 20 steps each (same bracketing; every rank runs them): their median next to the single
    # figure above, and the box calibration, so that a round-to-round delta can be told from box-to-box variance
    regions, calib = None, None
    if not args.no_calibration:
        regions = []
        for _ in range(3):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for i in range(20):
                step(batches[i % pool])
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            regions.append((time.perf_counter() - t1) / 20 * 1e3)
        if rank == 0:
            calib = box_calibration(device)
    probe_mode = 'HIP events around every launch of the kernel inside the timed region (eager launches)'
    per_tvis = None
    if not args.eager and rank == 0 and args.workload == 'cfg3':
        # hipGraph replay hides individual launches from host-recorded events, so the single largest kernel (the
        # embedder's conv3: `roofline_best`) is timed with HIP events on its launch stream over `probe_steps` eager
        # forward + backward passes of the same workload, run right after the timed region (same process, same buffers,
        # same shapes); profiles/ holds the rocprofv3 --kernel-trace average of the same kernel inside the graph replays.
        nn_ops.set_probe(probe)
        model.require_sync = False            # rank-0-only steps: no collective (the other ranks are not in them)
        for i in range(args.probe_steps):       # forward + backward only: the replicas' parameters stay in step
            if isinstance(step, GraphedTrainStep):
                step.pts.copy_(batches[i % pool])
                step._fwd_bwd(step._draw())
            else:
                lx, ln = model(batches[i % pool], batches[i % pool])
                (lx + float(config.normal_weight) * ln.sum()).backward()
            model.zero_grad()
        torch.cuda.synchronize()
        nn_ops.set_probe(None)
        if isinstance(step, GraphedTrainStep) and args.probe_steps:
            per_tvis = per_tvis_families(step, batches[0], nn_ops)      # the `roofline` families, per visible-token count
        model.require_sync = True
        probe_mode = ('HIP events around every launch of the kernel over %d eager steps run right after the '
                      'timed hipGraph region' % args.probe_steps)
    nn_ops.set_probe(None)
    table = None
    if not args.eager and rank == 0 and args.workload == 'cfg3' and not args.no_tvis_table and isinstance(step, GraphedTrainStep):
        table = tvis_table(step, batches[0])
    in_sync = None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        # data-parallel sanity after the timed steps: every replica must hold bit-identical parameters
        # (same broadcast start, same averaged gradients, same update) -- max and min over ranks of a
        # checksum pair agree exactly or the collective path is broken
        flat = model.flat_param
        chk = torch.stack([flat.double().sum(), flat.double().abs().sum(), flat[::1009].double().pow(2).sum()])
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool(torch.equal(hi, lo)) and bool(torch.isfinite(chk).all())

    if rank == 0:
        clouds_per_s = args.batch * world * args.steps / elapsed
        kern = probe.summary()
        from point_dae_amd import _lib as _L
        bf16x3 = _L.gemm_arith() == _L.GEMM_BF16X3
        roof = None
        if kern:
            ach = kern['flops'] / (kern['avg_ms'] * 1e-3) / 1e12
            # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be collected
            # inside this run); the newest committed profile that has this kernel at this shape, or null
            kname = 'rows3::conv3_kernel<0, 3>' if bf16x3 else 'gemm_nt_kernel<256,256,NONE,GROUPBIAS_STATS>'
            traffic, traffic_src = None, None
            for rnd in range(ROUND, 0, -1):
                pmc = os.path.join(ROOT, 'profiles', 'pmc_r%02d.json' % rnd)
                if os.path.exists(pmc) and args.batch == 128 and args.num_group == 64 and args.npoints == 1024:
                    rec = json.load(open(pmc)).get(kname)
                    if rec:
                        traffic, traffic_src = rec['hbm_bytes_per_launch'], 'profiles/pmc_r%02d.json (rocprofv3 --pmc, not this run)' % rnd
                        break
            peak = BF16X3_PEAK_TFLOPS if bf16x3 else MFMA_F32_PEAK_TFLOPS
            R_, c3_, c2_ = args.batch * args.num_group * 32, 512, 256
            roof = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                    'frac_of_f32_mfma_peak': ach / MFMA_F32_PEAK_TFLOPS,
                    'traffic': traffic, 'traffic_source': traffic_src, 'traffic_measured_in_this_run': False,
                    'algorithmic_bytes': 4.0 * (R_ * c2_ + c3_ * c2_ + R_ * c3_ + (R_ // 32) * c3_),
                    'kernel': ('pdae::' + kname) + ': ' + kern['name'],
                    'avg_us': kern['avg_ms'] * 1e3, 'launches': kern['launches'],
                    'flops_per_launch': kern['flops'], 'timing': probe_mode,
                    'peak_note': ('exact-split bf16 ceiling (bf16 dense %.0f / 6 products); fp32-input MFMA peak %.1f alongside'
                                  % (MFMA_BF16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS)) if bf16x3
                                 else 'fp32-input MFMA (v_mfma_f32_32x32x2_f32) dense peak, MI355X_MICROARCH.md'}
        best = roof                                   # the single largest hand-written kernel (the embedder's conv3)
        dom = dominant_roofline(per_tvis, args.num_group, bf16x3)
        line = {
            'metric': 'pretrain point-clouds/sec (N=%d,G=%d)' % (args.npoints, args.num_group), 'value': clouds_per_s, 'unit': 'clouds/s',
            'n_gpus': world, 'rccl_ranks': (dist.get_world_size() if world > 1 and backend == 'nccl' else (1 if world == 1 else 0)),
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 (3xbf16 exact-split operands, fp32 accumulate)' if bf16x3 else 'f32', 'data': 'synthetic',
            'config': {'workload': (('model ' + args.model_name + ' on ' if args.model_name else '') +
                                    'cfg3/cfg4: pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml '
                                    if args.workload == 'cfg3' else
                                    'cfg2: pretrain_PointCAE_affine_r3_dropout_local_4xlonger.yaml (Point_CAE_PointNetv2) ') +
                                   'full train step (fwd+loss+bwd+AdamW%s)' % ('+RCCL all-reduce' if world > 1 else ''),
                       'local_batch': args.batch, 'global_batch': args.batch * world, 'npoints': args.npoints,
                       'num_group': args.num_group, 'group_size': 32, 'parallelism': 'dp%d' % world, 'launch': 'eager' if args.eager else 'hipGraph replay',
                       'grad_allreduce': ('none (1 rank)' if world == 1 else
                                          'Transformer slice under the embedder backward (two graphs)' if getattr(step, 'split', False)
                                          else 'one flat all-reduce after the replay'),
                       'dense_layers': ('hand-written MFMA kernels, no BLAS library in the step: Linear layers and weight gradients on '
                                        'exact-split bf16 (csrc/rows3_kernel.h: fp32 = three bf16 terms, six products, fp32 accumulate; '
                                        'PDAE_GEMM=f32mfma selects the fp32-input kernels), the patch embedder\'s fused convolutions on the same '
                                        'arithmetic (rows3::conv3_kernel); only the K = 3 / K = 4 layers stay on fp32-input MFMA')
                                       if bf16x3 else 'hand-written fp32 MFMA kernels (csrc/rows_gemm.hip, gemm.hip); no BLAS library in the step'},
            'roofline': dom if dom else best,
            'roofline_best': best if dom else None,
            'box_calibration': calib,
            'timed_regions': ({'ms_per_step': regions, 'median_ms_per_step': sorted(regions)[1], 'steps_each': 20,
                               'note': 'three more timed regions run right after the one `value` is computed from'}
                              if regions else None),
            # sanity of the timed steps: Chamfer loss of the first and of the last timed optimisation step
            'loss': {'first_timed_step': float(loss_first), 'last_timed_step': float(loss_last)},
        }
        if per_tvis:
            # every dense FLOP of the step (row GEMMs, weight gradients, the embedder's fused convolutions) as the
            # EXPECTATION over P(T_vis) of the per-T_vis launch-site counts -- deterministic -- over the matching
            # expectation of the replay time (tvis_table) and over the timed ms/step (which includes AdamW)
            pt = tvis_distribution(args.num_group)
            ts = [t for t in sorted(per_tvis) if t in pt]
            z = sum(pt[t] for t in ts)
            by_fam = {}
            for t in ts:
                for k, r in per_tvis[t].items():
                    by_fam[k] = by_fam.get(k, 0.0) + pt[t] / z * r['gflop']
            gf = sum(by_fam.values())
            line['whole_step'] = {'gflop_executed': gf, 'unit': 'TFLOP/s', 'gflop_by_family': by_fam,
                                  'covered_probability_mass': z,
                                  'timed_ms_per_step': elapsed / args.steps * 1e3,
                                  'achieved_on_timed_ms': gf / (elapsed / args.steps * 1e3),
                                  'note': 'fp32 FLOPs of every dense product the step executes, counted at the launch sites per '
                                          'visible-token count and weighted with P(T_vis); `achieved` = over the same expectation of '
                                          'the forward + loss + backward replay time (tvis_table), `achieved_on_timed_ms` over the '
                                          'timed ms/step (its own mask draws, AdamW included)'}
            if table and all(t in table for t in ts):
                exp_ms = sum(pt[t] / z * table[t] for t in ts)
                line['whole_step'].update(expected_replay_ms=exp_ms, achieved=gf / exp_ms,
                                          frac_of_f32_mfma_peak=gf / exp_ms / MFMA_F32_PEAK_TFLOPS,
                                          frac_of_bf16x3_peak=gf / exp_ms / BF16X3_PEAK_TFLOPS)
        if table:
            line['tvis_table'] = {'unit': 'ms per replay (forward + loss + backward graph(s), no AdamW)',
                                  'ms': {str(t): round(v, 4) for t, v in table.items()}}
            line['ddp_model'] = ddp_model(table, args.num_group, elapsed / args.steps * 1e3)
        if world > 1:
            line['replicas_in_sync'] = in_sync          # parameters bit-identical on all ranks after the timed steps
        if args.workload == 'cfg3' and not args.no_also:
            line['roofline_geometry'] = geometry_rooflines(args, clouds)
            if world == 1:
                del step
                line['also'] = {'cfg2': cfg2_leg(args, device, rank)}
                line['also']['published_variant'] = published_leg(args, device, rank)
                line['also']['dgcnn'] = dgcnn_leg(args, device, rank)
                line['also']['cfg5_shape'] = cfg5_leg(args, device, rank)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(config, args)
            line['cpu_baseline']['whole_box'] = cpu_baseline_box(args, line['cpu_baseline'])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
