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
