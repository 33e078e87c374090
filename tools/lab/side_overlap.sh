#!/bin/bash
# start / end of the fold_input launches and the kernels around them in a few cfg2 steps (kernel trace).  Written for the
# side-stream experiment of tools/lab/NOTES.md (fold_input in slices beside the h2 GEMM: that patch is not in the tree; on the
# shipped step the table shows the one serial launch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/side_overlap; rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --workload cfg2 --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 3 --warmup 3 > $OUT.log 2>&1
t=$(find $OUT -name '*kernel_trace.csv' | head -1)
python - "$t" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'fold_input' in r[2] and 'grad' not in r[2]]
i0 = idx[-4] if len(idx) >= 4 else idx[0]
t0 = rows[i0][0]
for s, e, n in rows[i0:i0 + 12]:
    print('%9.1f -> %9.1f us  (%7.1f)  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $OUT
