"""bench.py's launcher contract on the CPU side: `--gpus N` is binding."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PDAE_BENCH_BACKEND')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=300)


def test_gpus_n_without_devices_is_refused():
    """No launcher, --gpus 8, fewer than 8 visible GPUs: exit != 0 and no result line (the driver
    must never record a single-GPU number as an 8-GPU point)."""
    import torch
    if torch.cuda.device_count() >= 8:
        return
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert '{"metric"' not in r.stdout
    assert 'refusing' in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(['--gpus', '4', '--steps', '1', '--warmup', '0'],
             {'RANK': '0', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'does not match' in (r.stderr + r.stdout)
    assert '{"metric"' not in r.stdout


def test_ddp_model_expected_maximum_matches_a_monte_carlo_draw():
    """bench.py's `ddp_model` (the only scaling figure a one-GPU pool can give): E[max over n ranks of the replay time of
    each rank's own T_vis draw], computed from the exact distribution of T_vis, against a direct simulation -- the mask
    ratio drawn as MaskTransformer._mask_center_rand draws it (uniform in [0.5, 0.8), num_mask = int(ratio * G))."""
    import importlib.util
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    G = 64
    p = bench.tvis_distribution(G)
    assert abs(sum(p.values()) - 1.0) < 1e-12 and min(p) == G - int(0.8 * G) and max(p) == G - int(0.5 * G)
    rng = np.random.default_rng(0)
    table = {t: 9.0 + 0.18 * t + 0.3 * np.sin(t) for t in p}           # ms per replay, not monotone in T_vis
    ratio = rng.uniform(0.5, 0.8, size=(400000, 8))
    tvis = G - (ratio * G).astype(np.int64)
    emp = {t: float((tvis == t).mean()) for t in p}
    assert max(abs(emp[t] - p[t]) for t in p) < 2e-3                    # the distribution itself
    ms = np.vectorize(table.get)(tvis)
    model = bench.ddp_model(table, G, ms_per_step=ms[:, 0].mean() + 0.4)
    assert abs(model['expected_replay_ms'] - ms[:, 0].mean()) < 5e-3 and abs(model['other_ms'] - 0.4) < 5e-3
    for n in (2, 4, 8):
        sim = ms[:, :n].max(axis=1).mean()
        got = model['predicted'][str(n)]
        assert abs(got['expected_max_replay_ms'] - sim) < 5e-3, (n, got, sim)
        assert abs(got['speedup'] - n * (ms[:, 0].mean() + 0.4) / (sim + 0.4)) < 5e-3
