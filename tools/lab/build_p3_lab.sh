#!/bin/bash
# lab build (never shipped): tools/lab/libp3_lab.so
cd "$(dirname "$0")/../.." && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $LABFLAGS \
  -shared -o tools/lab/libp3_lab.so tools/lab/p3_lab.hip
