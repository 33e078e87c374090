"""point_dae_amd -- MI355X-native Point-DAE pretraining hot path.

The package holds the HIP kernels (csrc/, built into libpdae_hip.so behind the
C ABI of include/pdae.h) and the Python host side that mirrors the reference's
operator / model interface for that path.  There is no CPU fallback.
"""
__version__ = "0.1.0"
