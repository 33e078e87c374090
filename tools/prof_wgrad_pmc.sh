#!/bin/bash
# Counter passes for VERDICT r4 item 4: what limits the k-tile of the stack-level weight gradients?  Three --pmc passes of a
# short bench run (SQ issue / wait mix; MFMA, VMEM and LDS side; L1 -> L2 requests, latency, L2 hit rate), summarised per
# (kernel, grid) for rows3::wgrad3b_kernel<false> next to the large persistent rows3::gemm3_kernel launches on the same box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/wgrad_pmc; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o pmc -- python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --steps 3 --warmup 1 --probe-steps 0 > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
run sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU"
run mem "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
python tools/pmc_kernel_counters.py $OUT/sq1 $OUT/sq2 $OUT/mem > $OUT/wgrad_counters.json
rm -rf $OUT/sq1 $OUT/sq2 $OUT/mem
python - <<'PY'
import json
d = json.load(open('gpurun_out/wgrad_pmc/wgrad_counters.json'))
for k, v in d.items():
    print(k); print('   ', {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
