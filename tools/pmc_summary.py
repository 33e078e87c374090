"""HBM bytes per launch of every hand-written kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs: they do not fit one pass on gfx950).
usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [kernel_summary.txt]  -> JSON on stdout.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads ->
doubled.  Units: KiB.  With the kernel summary of the same command (tools/trace_summary.py: calls per step) the bytes of
the row-GEMM families (gemm3 / wgrad3t / wgrad3b / wgrad_reduce, or their fp32-input twins) are also summed per STEP:
`rows_families_hbm_bytes_per_step`, the `traffic` of bench.py's roofline."""
import csv
import glob
import json
import os
import re
import sys


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    return n.replace('pdae::', '')


def collect(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get('Counter_Name') != counter or 'pdae::' not in r['Kernel_Name']:
                    continue
                a = out.setdefault(short(r['Kernel_Name']), {})
                a[r['Dispatch_Id']] = a.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    return {k: (len(v), sum(v.values())) for k, v in out.items()}


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
res = {}
for key in sorted(fetch):
    n, s = fetch[key]
    wn, ws = write.get(key, (0, 0.0))
    fk, wk = s / n, (ws / wn if wn else 0.0)
    res[key] = {'FETCH_SIZE_KB_avg': fk, 'WRITE_SIZE_KB_avg': wk, 'dispatches': n, 'hbm_bytes_per_launch': (2 * fk + wk) * 1024}
res['_note'] = ('hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled (gfx950 reports half of wide coalesced '
                'reads, MI355X_MICROARCH.md HBM section); separate --pmc passes; fabric-side counters: Infinity-Cache hits included')
if len(sys.argv) > 3:
    calls = {}
    for ln in open(sys.argv[3]):
        m = re.match(r'\s*[\d.]+%\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)$', ln)
        if m:
            calls[short(m.group(4).strip())] = float(m.group(2))
    fam = [k for k in res if re.search(r'rows3::gemm3_kernel|rows3::wgrad3b_kernel|rows3::wgrad3t_kernel|rows::wgrad_reduce_kernel|rows::rows_gemm_kernel|rows::wgrad_kernel', k)]
    total, detail = 0.0, {}
    for k in fam:
        c = calls.get(k)
        if c is None:
            continue
        b = res[k]['hbm_bytes_per_launch'] * c
        total += b
        detail[k] = {'calls_per_step': c, 'bytes_per_step': b}
    res['rows_families_hbm_bytes_per_step'] = total
    res['rows_families_detail'] = detail
print(json.dumps(res, indent=1))
