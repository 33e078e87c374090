#!/bin/bash
# HIP runtime knobs vs ms/step of the default bench (one process per setting; same box).
# Usage (GPU box): bash tools/lab/env_sweep.sh > gpurun_out/env_sweep.txt
cd "$GRAFT_REPO_ROOT"
run() {
  local tag="$1"; shift
  local line
  line=$(env "$@" python bench.py --no-cpu-baseline --no-also --probe-steps 0 --steps 40 --warmup 10 2>/dev/null | grep '"metric"' | tail -1)
  echo "$tag $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>/dev/null)"
}
run base            PDAE_X=0
run base2           PDAE_X=0
run dev_kernarg1    HIP_FORCE_DEV_KERNARG=1
run dev_kernarg0    HIP_FORCE_DEV_KERNARG=0
run pkt_capture1    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pkt_capture0    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph_batch1    DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graph_batch64   DEBUG_HIP_GRAPH_BATCH_SIZE=64
run sysscope0       ROC_SYSTEM_SCOPE_SIGNAL=0
run optflush0       AMD_OPT_FLUSH=0
run kernarg_copyopt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run fgs_kernarg0    ROC_USE_FGS_KERNARG=0
run base3           PDAE_X=0
