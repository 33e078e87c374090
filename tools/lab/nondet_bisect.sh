#!/bin/bash
# (lab) which exact-split kernel family in the OTHER process disturbs a process's small reductions?  Process A always runs
# the fp32-input kernels; process B runs: everything exact-split | only the row GEMMs | only the weight gradients | fp32.
cd "$GRAFT_REPO_ROOT"
run() {
  echo "--- B: $1"
  ITERS=${ITERS:-150} PDAE_GEMM=f32mfma python tools/lab/model_nondet.py A_f32 2>&1 | grep -v amdgpu | grep -v intermediate | cut -c1-260 &
  env $2 ITERS=${ITERS:-150} python tools/lab/model_nondet.py "B_$1" 2>&1 | grep -v amdgpu | grep -v intermediate | cut -c1-260 &
  wait
}
run bf16x3_all "PDAE_X=0"
run bf16x3_rows_only "PDAE_GEMM_ONLY=rows"
run bf16x3_wgrad_only "PDAE_GEMM_ONLY=wgrad"
run f32 "PDAE_GEMM=f32mfma"
