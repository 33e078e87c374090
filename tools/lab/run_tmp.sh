cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_block.py -x -q 2>&1 | tail -4
python tools/lab/attn_time.py 2>&1 | grep -v amdgpu
PDAE_ATTN=f32 python tools/lab/attn_time.py 2>&1 | grep -v amdgpu
ROUNDS=3 bash tools/lab/abn.sh PDAE_ATTN f32 b
