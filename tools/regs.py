"""Register / spill / occupancy table of every kernel in a HIP source (compile only, no GPU):
    python tools/regs.py tools/lab/rows3_lab.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-c', src,
       '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.split('\n'):
    m = re.search(r'Function Name: (\S+)', ln)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {'name': name}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if 'error' in ln:
        print(ln)
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'vSpill':>6} {'sSpill':>6} {'occ':>4} {'scratch':>8}  kernel")
for r in rows:
    print(f"{r.get('VGPRs', -1):5d} {r.get('AGPRs', -1):5d} {r.get('TotalSGPRs', -1):5d} {r.get('VGPRs Spill', -1):6d} "
          f"{r.get('SGPRs Spill', -1):6d} {r.get('Occupancy', -1):4d} {r.get('ScratchSize', -1):8d}  "
          f"{re.sub(r'pdae::rows3?::', '', r['name'])[:110]}")
