#!/bin/bash
# gpurun -- bash tools/xproc_repro.sh     (binaries built here, see the header of tools/xproc_repro.hip)
cd "$GRAFT_REPO_ROOT"
X=tools/lab/lab_xproc
echo "--- victim alone";                                          $X B 3 | tail -1
echo "--- victim beside the reduced GEMM loop (another process)"; $X G 6 & sleep 1.5; $X B 3 | tail -3; wait
echo "--- the same victim built with -fno-slp-vectorize";         $X G 6 & sleep 1.5; ${X}_noslp B 3 | tail -1; wait
