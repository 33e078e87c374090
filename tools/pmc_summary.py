"""HBM bytes per launch of the hand-written GEMM / attention kernels from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; separate runs: they do not fit one pass on gfx950).
usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>  -> JSON on stdout.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide
coalesced reads -> doubled.  Units: KiB."""
import csv
import glob
import json
import os
import re
import sys

NAMES = {
    r'gemm_nt_kernel<256, 256, 0, 3>': 'gemm_nt_kernel<256,256,NONE,GROUPBIAS_STATS>',
    r'gemm_nt_kernel<128, 384, 1, 4>': 'gemm_nt_kernel<128,384,BNRELU,GROUPMAX>',
    r'gemm_nt_kernel<256, 256, 1, 5>': 'gemm_nt_kernel<256,256,BNRELU,STORE_GROUPMAX>',
    r'gemm_tn_kernel<128, 128, 1>': 'gemm_tn_kernel<BNRELU>',
    r'gemm_tn_kernel<128, 128, 0>': 'gemm_tn_kernel<NONE>',
    r'rows_gemm_kernel<1, 1, 2, 2, false, 0>': 'rows_gemm_kernel<64x64,NT,STORE>',
    r'rows_gemm_kernel<1, 1, 2, 2, true, 0>': 'rows_gemm_kernel<64x64,KN,STORE>',
    r'rows_gemm_kernel<1, 1, 2, 2, false, 2>': 'rows_gemm_kernel<64x64,NT,GELU>',
    r'rows_gemm_kernel<1, 1, 2, 2, true, 3>': 'rows_gemm_kernel<64x64,KN,MUL>',
    r'rows::wgrad_kernel': 'wgrad_kernel',
    r'rows::wgrad_reduce_kernel': 'wgrad_reduce_kernel',
    r'fps_kernel': 'fps_kernel',
    r'knn_kernel': 'knn_kernel',
    r'chamfer_fwd_packed': 'chamfer_fwd_packed',
    r'chamfer_bwd_packed': 'chamfer_bwd_packed',
    r'add_layernorm_fwd_kernel': 'add_layernorm_fwd_kernel',
    r'attention_fwd_kernel': 'attention_fwd_kernel',
    r'attention_bwd_kernel': 'attention_bwd_kernel',
    r'layernorm_bwd_kernel': 'layernorm_bwd_kernel',
    r'bnrelu_backward_apply_kernel': 'bnrelu_backward_apply_kernel',
    r'bnrelu_backward_reduce_kernel': 'bnrelu_backward_reduce_kernel',
}


def collect(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get('Counter_Name') != counter:
                    continue
                for pat, key in NAMES.items():
                    if pat in r['Kernel_Name']:
                        a = out.setdefault(key, [0, 0.0])
                        a[0] += 1
                        a[1] += float(r['Counter_Value'])
                        break
    return out


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
res = {}
for key in fetch:
    n, s = fetch[key]
    wn, ws = write.get(key, [0, 0.0])
    fk, wk = s / n, (ws / wn if wn else 0.0)
    res[key] = {'FETCH_SIZE_KB_avg': fk, 'WRITE_SIZE_KB_avg': wk, 'dispatches': n,
                'hbm_bytes_per_launch': (2 * fk + wk) * 1024,
                'note': 'FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); separate --pmc passes'}
print(json.dumps(res, indent=1))
