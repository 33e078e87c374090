#!/bin/bash
cd "$GRAFT_REPO_ROOT"
ITERS=${ITERS:-200} PDAE_GEMM=f32mfma python tools/lab/model_nondet.py A_f32 2>&1 | grep -v amdgpu | cut -c1-300 &
PDAE_GEMM_ONLY=rows ITERS=${ITERS:-200} python tools/lab/model_nondet.py B_rows 2>&1 | grep -v amdgpu | cut -c1-300 &
wait
