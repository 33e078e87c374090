#!/bin/bash
# diagnostic build of the row-GEMM kernels with in-kernel stamps (never shipped): tools/lab/libpdae_lab.so
cd "$(dirname "$0")/../../point_dae_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off \
  -DPDAE_ROWS_STAMPS $LABFLAGS -shared -o ../../tools/lab/libpdae_lab${LABTAG}.so -x hip rows_gemm.hip abi.cpp
