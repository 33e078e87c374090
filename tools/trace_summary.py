"""Per-step kernel summary of the timed region from a rocprofv3 kernel_trace.csv
(`--kernel-trace --output-format csv`).  usage: trace_summary.py <kernel_trace.csv> <timed steps>
The timed region = everything after the (2*steps+1)-th-from-last adamw launch (two per step)."""
import csv
import os
import re
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])))
rows.sort(key=lambda r: r[1])
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
sel = rows[ad[-(2 * steps + 1)] + 1:]
span = (sel[-1][2] - sel[0][1]) / 1e6


def short(n):
    n = re.sub(r'void at::native::', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    if n.startswith('Cijk_'):
        m = re.search(r'(Cijk_[A-Za-z]+_[A-Za-z]+)_.*?(MT\d+x\d+x\d+)', n)
        return 'library GEMM ' + (m.group(1) + ' ' + m.group(2) if m else '')
    if 'pdae::gemm' in n or 'pdae::rows::' in n or 'pdae::rows3::' in n or 'pdae::layernorm_bwd' in n or 'pdae::colsum2' in n or 'pdae::fps' in n or 'pdae::knn' in n:
        return re.sub(r'\(.*', '', n).replace('void ', '')
    return re.sub(r'[<(].*', '', n).replace('void ', '')[:80]


agg = {}
for n, s, e in sel:
    a = agg.setdefault(short(n), [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
lib = sum(v[1] for k, v in agg.items() if k.startswith('library GEMM'))
mine = sum(v[1] for k, v in agg.items() if k.startswith('pdae::'))
print('# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-also --probe-steps 0   (timed region: the last %d steps, hipGraph replays)' % steps)
print('# wall %.3f ms/step, GPU busy %.3f ms/step, %.1f kernels/step' % (span / steps, tot / 1e6 / steps, sum(v[0] for v in agg.values()) / steps))
print('# library GEMMs %.3f ms/step, hand-written pdae:: kernels %.3f ms/step, other (torch) %.3f ms/step'
      % (lib / 1e6 / steps, mine / 1e6 / steps, (tot - lib - mine) / 1e6 / steps))
print('%7s %12s %11s %11s  %s' % ('%time', 'us/step', 'calls/step', 'avg us', 'kernel'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[1] / tot < float(os.environ.get('TS_MIN', '0.0008')):
        continue
    print('%6.2f%% %12.1f %11.1f %11.1f  %s' % (100 * v[1] / tot, v[1] / 1e3 / steps, v[0] / steps, v[1] / 1e3 / v[0], k))
