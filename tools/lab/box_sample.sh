#!/bin/bash
# One line per box: the default step time next to the box calibration (bench.py box_calibration) -- run it several times,
# every gpurun call lands on another box:   gpurun -- bash tools/lab/box_sample.sh
cd "$GRAFT_REPO_ROOT"
python bench.py --no-cpu-baseline --no-also --no-tvis-table --probe-steps 0 2>/dev/null | grep '"metric"' | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
c, r = d["box_calibration"], d["timed_regions"]
print("ms/step %.3f  median of three more regions %.3f  | bf16 MFMA loop %.0f TFLOP/s at %.2f GHz, copy %.0f GB/s" % (
    d["ms_per_step"], r["median_ms_per_step"], c["mfma_bf16"]["tflops"], c["mfma_bf16"]["shader_clock_ghz"], c["copy_f4"]["gbs"]))'
