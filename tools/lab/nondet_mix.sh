#!/bin/bash
# (lab) one process on the fp32-input kernels, one on the exact-split kernels, sharing the GPU: who sees differing bits?
cd "$GRAFT_REPO_ROOT"
ITERS=120 PDAE_GEMM=f32mfma python tools/lab/model_nondet.py A_f32 2>&1 | grep -v amdgpu | grep -v intermediate | cut -c1-200 &
ITERS=120 python tools/lab/model_nondet.py B_bf16x3 2>&1 | grep -v amdgpu | grep -v intermediate | cut -c1-200 &
wait
echo "--- f32 process next to a process that only runs torch matmuls"
ITERS=120 PDAE_GEMM=f32mfma python tools/lab/model_nondet.py A_f32 2>&1 | grep -v amdgpu | grep -v intermediate | cut -c1-200 &
python -c "
import torch, time
a=torch.randn(4096,4096,device='cuda'); t=time.time()
while time.time()-t<12: b=a@a
torch.cuda.synchronize()" &
wait
echo "--- bf16x3 process next to torch matmuls"
ITERS=120 python tools/lab/model_nondet.py B_bf16x3 2>&1 | grep -v amdgpu | grep -v intermediate | cut -c1-200 &
python -c "
import torch, time
a=torch.randn(4096,4096,device='cuda'); t=time.time()
while time.time()-t<12: b=a@a
torch.cuda.synchronize()" &
wait
