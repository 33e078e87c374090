#!/bin/bash
# LAB: occupancy of the SHIPPED 64x64 row-GEMM kernel as the runtime sees it (-DPDAE_LAB_OCC adds only the query), next to
# the stamped diagnostic build's.  gpurun -- bash tools/lab/occupancy.sh
cd "$GRAFT_REPO_ROOT/point_dae_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPDAE_LAB_OCC -shared -o ../../tools/lab/libpdae_lab.so -x hip rows_gemm.hip abi.cpp 2>&1 | grep error
cd "$GRAFT_REPO_ROOT"
echo "== shipped kernels"; python tools/lab/occupancy.py 2>&1 | grep -v amdgpu | sed -n '1p;5p;9p'
bash tools/lab/build_lab.sh 2>&1 | grep error
echo "== stamped build"; python tools/lab/occupancy.py 2>&1 | grep -v amdgpu | sed -n '1p;5p'
