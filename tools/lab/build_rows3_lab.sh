#!/bin/bash
# lab build (never shipped): tools/lab/librows3_lab.so -- variants of the exact-split bf16 row GEMM
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $LABFLAGS \
  -shared -o tools/lab/librows3_lab.so tools/lab/rows3_lab.hip
