#!/bin/bash
# lab: the whole step with the grouped weight-gradient kernel's grid forced to N blocks
cd "$GRAFT_REPO_ROOT/point_dae_amd/csrc" && touch rows_gemm.hip && make -j8 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DPDAE_LAB_PLAN" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for n in 512 768 1024 256 640; do
  echo "wgrad blocks $n: $(PDAE_WGRAD_BLOCKS=$n python bench.py --no-cpu-baseline --no-also --probe-steps 0 --steps 30 2>&1 | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
