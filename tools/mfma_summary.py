"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass
(--output-format csv).  util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the
formula of rocprofv3's derived MfmaUtil counter, with GRBM_GUI_ACTIVE divided by the 8 XCDs it is
reported summed over on this part (19.8 G counts per second of kernel time = 8 x 2.4 GHz; check: a
kernel of 16.8 M v_mfma_f32_32x32x2_f32 at 64 cycles each reports 1.07e9 busy cycles).
usage: mfma_summary.py <dir> -> JSON on stdout."""
import csv
import glob
import json
import os
import re
import sys

SIMDS = 256 * 4
XCDS = 8
acc = {}
for path in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r['Kernel_Name']
            if n.startswith('Cijk_'):
                key = 'library GEMMs (hipBLASLt/rocBLAS, all shapes)'
            elif 'pdae::' in n:
                key = re.sub(r'\(.*', '', n).replace('void ', '')
            else:
                continue
            a = acc.setdefault(key, {})
            d = a.setdefault(r['Dispatch_Id'], {})
            v = float(r['Counter_Value'])
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':      # one row per XCD: reduce(max), as MfmaUtil does
                d['GRBM_GUI_ACTIVE'] = max(d.get('GRBM_GUI_ACTIVE', 0.0), v)
            else:                                           # reduce(sum)
                d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + v
# Second denominator: the dispatch's own duration from the kernel trace of the same pass, at the 2.4 GHz peak clock.
# GRBM_GUI_ACTIVE / 8 reads HIGH on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back): a 40 us row
# GEMM reports ~113 k "cycles" = 2.9 GHz, which understates its utilisation by a third; the duration-based figure is
# busy cycles / (1024 SIMDs x duration x 2.4e9) = the fraction of the MFMA PEAK the dispatch reached.
PEAK_HZ = 2.4e9
dur = {}
for path in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9
out = {}
for key, disp in acc.items():
    busy = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for d in disp.values())
    act = sum(d.get('GRBM_GUI_ACTIVE', 0.0) for d in disp.values())
    if busy <= 0 or act <= 0:
        continue
    out[key] = {'dispatches': len(disp), 'SQ_VALU_MFMA_BUSY_CYCLES_avg': busy / len(disp),
                'GRBM_GUI_ACTIVE_avg': act / len(disp), 'mfma_util': busy / (act / XCDS * SIMDS)}
    secs = sum(dur.get(i, 0.0) for i in disp)
    if secs > 0 and all(i in dur for i in disp):
        out[key]['avg_us_in_this_pass'] = secs / len(disp) * 1e6
        out[key]['mfma_util_vs_peak_clock'] = busy / (SIMDS * secs * PEAK_HZ)
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]['mfma_util'])), indent=1))
