"""Per-step kernel summary of the TIMED region of `rocprofv3 --kernel-trace -- python bench.py ...`
from the rocpd sqlite database rocprofv3 writes (bench_results.db).

usage: python tools/prof_db.py <results.db> <timed steps> [top N] [substring filter]
The timed region = everything after the (2*steps+1)-th-from-last adamw launch (two per step).
"""
import re
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
filt = sys.argv[4] if len(sys.argv) > 4 else None
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
sel = rows[ad[-(2 * steps + 1)] + 1:]
span = (sel[-1][2] - sel[0][1]) / 1e6


def short(n):
    n = re.sub(r'void at::native::', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    if n.startswith('Cijk_'):
        m = re.search(r'(Cijk_[A-Za-z]+_[A-Za-z]+)_.*?(MT\d+x\d+x\d+)', n)
        return 'library GEMM ' + (m.group(1) + ' ' + m.group(2) if m else '')
    if 'pdae::gemm' in n or 'pdae::rows::' in n:
        return re.sub(r'\(.*', '', n).replace('void ', '')
    return re.sub(r'[<(].*', '', n).replace('void ', '')[:80]


if filt == 'SEQ':      # the kernel sequence of the last step, non-library non-pdae kernels named in full
    one = rows[ad[-3] + 1:ad[-1] + 1]
    for n, s, e, gx, gy, gz in one:
        k = short(n)
        if not (k.startswith('library') or k.startswith('pdae::')):
            k = re.sub(r'void at::native::|\(anonymous namespace\)::|at::native::', '', n)[:150]
        print('%8.1f us  grid %9d  %s' % ((e - s) / 1e3, gx * gy * gz, k))
    sys.exit(0)
agg = {}
for n, s, e, gx, gy, gz in sel:
    k = short(n)
    if filt and filt not in n:
        continue
    if filt:
        k = '%s grid(%d,%d,%d)' % (k, gx, gy, gz)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
print('# timed region: %d steps, wall %.3f ms/step, GPU busy %.3f ms/step, %.1f kernels/step'
      % (steps, span / steps, tot / 1e6 / steps, sum(v[0] for v in agg.values()) / steps))
lib = sum(v[1] for k, v in agg.items() if k.startswith('library GEMM'))
mine = sum(v[1] for k, v in agg.items() if k.startswith('pdae::'))
print('# library GEMMs %.3f ms/step, pdae:: kernels %.3f ms/step, other (torch) %.3f ms/step'
      % (lib / 1e6 / steps, mine / 1e6 / steps, (tot - lib - mine) / 1e6 / steps))
print('%7s %12s %11s %11s  %s' % ('%time', 'us/step', 'calls/step', 'us/call', 'kernel'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%6.2f%% %12.1f %11.1f %11.1f  %s' % (100 * v[1] / tot, v[1] / 1e3 / steps, v[0] / steps, v[1] / 1e3 / v[0], k))
