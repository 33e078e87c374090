"""Re-tune the library-GEMM selections for the benchmarked workload (run on an MI355X):
    python tools/retune_gemms.py && cp gpurun_out/tunableop_retuned_0.csv point_dae_amd/tunableop_gfx950.csv
Drives bench.py's set-up (all 20 visible-token graphs are warmed eagerly, which is where TunableOp tunes)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
env = dict(os.environ, PDAE_RETUNE='1', PDAE_RETUNE_OUT=os.path.join(root, 'gpurun_out', 'tunableop_retuned_%d.csv'))
sys.exit(subprocess.call([sys.executable, os.path.join(root, 'bench.py'), '--steps', '5', '--warmup', '2',
                          '--no-cpu-baseline'], env=env))
