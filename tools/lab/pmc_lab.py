"""per-kernel averages of every counter in rocprofv3 --pmc CSV output directories: pmc_lab.py <dir> [<dir> ...]"""
import csv
import glob
import os
import re
import sys

acc = {}
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
                if 'pdae' not in n:
                    continue
                a = acc.setdefault(n, {}).setdefault(r['Counter_Name'], {})
                a[r['Dispatch_Id']] = a.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
for n, cs in acc.items():
    print(n)
    for c, disp in sorted(cs.items()):
        print(f"   {c:32s} {sum(disp.values()) / len(disp):16.0f}   ({len(disp)} dispatches)")
