#!/bin/bash
# lab build (never shipped): tools/lab/libchain3_lab.so -- the in-launch GEMM -> GEMM seam prototype (chain3_lab.hip)
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
  -shared -o tools/lab/libchain3_lab.so tools/lab/chain3_lab.hip
