#!/bin/bash
# LAB: fragment reads per slot behind the k-tile barrier of rows3::gemm3_kernel (R3_FPS = 3 shipped): the barrier moves
# earlier as R3_FPS shrinks (more MFMA slots cover the next tile's first fragment reads, fewer slots carry the split).
# Build here:  bash tools/lab/fps_sweep.sh build      Run on the GPU box:  bash tools/lab/fps_sweep.sh
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  for f in 3 2 1; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DR3_FPS=$f -shared \
      -o tools/lab/librows3_lab_fps$f.so tools/lab/rows3_lab.hip &
  done
  wait; exit 0
fi
for f in 3 2 1; do
  echo "== R3_FPS=$f"; LABSO=librows3_lab_fps$f.so VARIANTS=0,2,3 SHAPES=2944x1152x384x0,2944x1536x384x0,2944x384x1536x0,8192x1536x384x0,2944x384x1152x1,65536x512x512x0,512x512x32768x0 python tools/lab/rows3_lab.py 2>&1 | grep -v amdgpu.ids
done
