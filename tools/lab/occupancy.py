"""LAB (diagnostic build): how many 64x64 row-GEMM blocks does the runtime say fit a CU, by dynamic LDS size?"""
import ctypes, os
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libpdae_lab.so'))
b, a, m = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
for lds in (0, 16384, 32768, 34816, 36864, 40960, 49152, 53248, 65536):
    rc = L.pdae_lab_occupancy(lds, ctypes.byref(b), ctypes.byref(a), ctypes.byref(m))
    print(f"dynamic LDS {lds:6d} B -> {b.value} blocks/CU (rc {rc}); device: {a.value} B LDS per CU, {m.value} B max per block")
