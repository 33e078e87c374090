#!/bin/bash
cd "$GRAFT_REPO_ROOT/point_dae_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPDAE_LAB_CLOCK -DPDAE_LAB_OCC -shared -o ../../tools/lab/libpdae_lab.so -x hip rows_gemm.hip abi.cpp 2>&1 | grep error
cd "$GRAFT_REPO_ROOT" && python tools/lab/rows_clock.py 2>&1 | grep -v amdgpu
