cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/labpmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/labpmc -o p -- tools/lab/lab2_64x64_w0 262144 512 256 > gpurun_out/labpmc.log 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.Counter()
for p in glob.glob('gpurun_out/labpmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in acc: print(k, acc[k]/n[k], n[k])
PY
rm -rf gpurun_out/labpmc
