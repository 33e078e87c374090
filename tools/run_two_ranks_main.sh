#!/bin/bash
# main.py's multi-rank path (--launcher pytorch) with two gloo ranks sharing one GPU: a code-path check
# (RCCL needs one device per rank).
C3=cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml
PDAE_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29544 -m point_dae_amd.main --config $C3 --launcher pytorch --total_bs 32 --max_epoch 1 \
  --steps_per_epoch 20 --exp_name two --root_folder gpurun_out/exp 2>&1 | grep -E "Batch 20|rror|Traceback" | cut -c1-200
rm -rf gpurun_out/exp
