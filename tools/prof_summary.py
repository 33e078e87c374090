"""Condense a rocprofv3 kernel_stats.csv: short names, per-step time, categories."""
import csv, re, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
def short(n):
    if n.startswith('Cijk_'):
        m = re.search(r'(Cijk_[A-Za-z]+_[A-Za-z]+)_.*?(MT\d+x\d+x\d+)', n)
        return 'rocblas/hipblaslt GEMM %s %s' % (m.group(1), m.group(2)) if m else 'GEMM'
    n = re.sub(r'void at::native::', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'<.*', '', n)
    return n[:70]
agg = {}
for r in rows:
    k = short(r['Name'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
tot = sum(v[1] for v in agg.values())
print('total kernel time %.3f ms over %.0f steps = %.3f ms/step' % (tot / 1e6, steps, tot / 1e6 / steps))
gemm = sum(v[1] for k, v in agg.items() if 'GEMM' in k)
print('GEMM share %.1f%% (%.3f ms/step)' % (100 * gemm / tot, gemm / 1e6 / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%6.2f%% %9.1f us/step %7.1f calls/step %9.1f us/call  %s' % (100 * v[1] / tot, v[1] / 1e3 / steps, v[0] / steps, v[1] / 1e3 / v[0], k))
