"""Library-GEMM solution selection for the plain (unfused) GEMMs.

The plain Linear layers of the Transformer blocks and the data-gradient GEMMs go
to hipBLASLt / rocBLAS through torch.mm.  PyTorch's default heuristic picks
mediocre solutions for the step's small-M shapes (M = 128 * T_vis rows); its
TunableOp facility times the libraries' solutions per shape.  The selections for
the benchmarked workload (B=128, N=1024, G=64: ~425 shapes) were tuned once on an
MI355X (`PYTORCH_TUNABLEOP_TUNING=1`, 5.4 min) and are shipped as
tunableop_gfx950.csv; at run time they are only looked up (tuning off), shapes
not in the file use the default heuristic.  20.3 -> 18.0 ms per step.

Call enable_tuned_gemms() before the first GEMM runs.
"""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
MASTER = os.environ.get('PDAE_TUNABLEOP_CSV') or os.path.join(_HERE, 'tunableop_gfx950.csv')


def enable_tuned_gemms(retune=False, out=None):
    """Point TunableOp at the shipped selections (or, with retune=True, tune
    every new shape and write the results next to `out`)."""
    if os.environ.get('PDAE_NO_TUNABLEOP') == '1' or not os.path.exists(MASTER):
        return False
    os.environ['PYTORCH_TUNABLEOP_ENABLED'] = '1'
    if retune:
        os.environ['PYTORCH_TUNABLEOP_TUNING'] = '1'
        os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS', '40')
        os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS', '5')
        os.environ['PYTORCH_TUNABLEOP_FILENAME'] = out or os.path.join(os.getcwd(), 'tunableop_retuned_%d.csv')
        return True
    os.environ['PYTORCH_TUNABLEOP_TUNING'] = '0'
    # TunableOp reads <name with %d -> device ordinal>; give every local device its copy
    d = os.path.join(tempfile.gettempdir(), 'pdae_tunableop_%d' % os.getuid())
    os.makedirs(d, exist_ok=True)
    ordinal = int(os.environ.get('LOCAL_RANK', '0'))
    for o in {0, ordinal}:
        dst = os.path.join(d, '%s_%d.csv' % (os.path.basename(MASTER)[:-4], o))
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(MASTER):
            tmp = dst + '.%d.tmp' % os.getpid()
            shutil.copyfile(MASTER, tmp)
            os.replace(tmp, dst)
    os.environ['PYTORCH_TUNABLEOP_FILENAME'] = os.path.join(d, os.path.basename(MASTER)[:-4] + '_%d.csv')
    return True
