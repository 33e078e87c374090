#!/bin/bash
cd "$GRAFT_REPO_ROOT"
X=tools/lab/lab_xproc
one() { echo "--- victim next to: $*"; env "$@" SECS=9 python tools/lab/xproc_agg.py 2>&1 | grep -v amdgpu & sleep 4; $X B 3 | tail -60; wait; }
one ARITH=bf16x3 CFG=16
