CFG=cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml
timeout 600 python -m point_dae_amd.main --config $CFG --max_epoch 4 --steps_per_epoch 100 --exp_name dbg --root_folder gpurun_out/exp 2>&1 | grep -E "Batch 100" ; rm -rf gpurun_out/exp
