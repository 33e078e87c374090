"""Per-(kernel, grid) averages of every counter found in rocprofv3 --pmc --kernel-trace csv passes (several directories: one
per pass), for the large launches of the row-GEMM family, with a few derived ratios.
usage: pmc_kernel_counters.py <pass dir> [<pass dir> ...] -> JSON on stdout."""
import csv
import glob
import json
import os
import re
import sys

WANT = re.compile(r'rows3::wgrad3b_kernel|rows3::wgrad3t_kernel|rows3::gemm3_kernel<1, 2, 4, 2, 2, (true|false), \d, true, 0, true>|rows3::conv3_kernel')
acc, dur = {}, {}
for d in sys.argv[1:]:
    durs = {}
    for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                durs[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            per = {}
            for r in csv.DictReader(f):
                n = r['Kernel_Name']
                if not WANT.search(n):
                    continue
                us = durs.get(r['Dispatch_Id'], 0.0)
                if us < 100.0:                                   # the large launches only
                    continue
                key = re.sub(r'\(.*', '', n).replace('void ', '').replace('pdae::', '') + ' grid ' + r.get('Grid_Size', '?')
                per.setdefault((key, r['Dispatch_Id']), {}).setdefault(r['Counter_Name'], 0.0)
                per[(key, r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
                dur[(key, r['Dispatch_Id'], d)] = us
            for (key, disp), cs in per.items():
                a = acc.setdefault(key, {})
                for c, v in cs.items():
                    s = a.setdefault(c, [0.0, 0])
                    s[0] += v
                    s[1] += 1
                s = a.setdefault('us_in_' + os.path.basename(d.rstrip('/')), [0.0, 0])
                s[0] += dur[(key, disp, d)]
                s[1] += 1
out = {}
for key, a in sorted(acc.items()):
    o = {c: s[0] / s[1] for c, s in a.items()}
    g = lambda c: o.get(c, 0.0)
    if g('SQ_WAVE_CYCLES'):
        w = g('SQ_WAVE_CYCLES')
        o['frac_wait_any'] = g('SQ_WAIT_ANY') / w
        o['frac_wait_inst_any'] = g('SQ_WAIT_INST_ANY') / w
        o['frac_active_any'] = g('SQ_ACTIVE_INST_ANY') / w
        o['frac_active_valu'] = g('SQ_ACTIVE_INST_VALU') / w
        o['frac_active_lds'] = g('SQ_ACTIVE_INST_LDS') / w
        o['frac_active_vmem'] = g('SQ_ACTIVE_INST_VMEM') / w
    if g('TCP_TCC_READ_REQ_sum'):
        o['l1_to_l2_read_latency_cycles'] = g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum')
    if g('TCC_HIT_sum') + g('TCC_MISS_sum'):
        o['l2_hit_rate'] = g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))
    if g('SQ_LDS_IDX_ACTIVE'):
        o['lds_conflict_frac'] = g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')
    if g('SQ_INSTS_VMEM_RD'):
        o['vmem_rd_cycles_per_inst'] = g('SQ_INST_CYCLES_VMEM_RD') / g('SQ_INSTS_VMEM_RD')
    for c in [c for c in o if c.startswith('us_in_')]:
        us = o[c]
        if c == 'us_in_sq2' and g('SQ_VALU_MFMA_BUSY_CYCLES'):
            o['mfma_busy_vs_2p4ghz'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * us * 1e-6 * 2.4e9)
    out[key] = o
print(json.dumps(out, indent=1))
