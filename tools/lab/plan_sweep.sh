#!/bin/bash
# lab: rebuild the library with the plan override compiled in, run the step with each tile shape forced
cd "$GRAFT_REPO_ROOT/point_dae_amd/csrc" && touch rows_gemm.hip && make -j8 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DPDAE_LAB_PLAN" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for f in "-1,-1,-1" "-1,0,-1" "3,0,0" "1,0,0" "2,0,0" "0,0,0" "4,0,0" "6,0,0"; do
  echo "force $f: $(PDAE_ROWS_FORCE=$f python bench.py --no-cpu-baseline --no-also --probe-steps 0 --steps 30 2>&1 | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
