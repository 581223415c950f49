"""(CPU) Register / scratch / LDS report of every kernel in one source, from hipcc's own remarks.
Usage: python tools/kres.py edge_bwd_f16.hip [extra hipcc flags...]"""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / 'pointvs_amd' / 'csrc'
src = sys.argv[1]
flags = sys.argv[2:]
if src == 'edge_mfma_fwd.hip':
    flags = ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1'] + flags
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-value', *flags,
                      '-Rpass-analysis=kernel-resource-usage', '-c', str(CSRC / src), '-o', '/dev/null'],
                     capture_output=True, text=True, cwd=CSRC)
if out.returncode:
    sys.exit(out.stderr[-3000:])
cur = None
rows = {}
for line in out.stderr.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = rows.setdefault(m.group(1), {})
        continue
    m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    short = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('void (anonymous namespace)::', '').split('(')[0][:60]
    print(f"{short:60s} VGPR {r.get('VGPRs', 0):3d} AGPR {r.get('AGPRs', 0):3d} spill {r.get('VGPRs Spill', 0):3d} "
          f"scratch {r.get('ScratchSize [bytes/lane]', 0):4d} SGPR {r.get('TotalSGPRs', r.get('SGPRs', 0)):3d} "
          f"sspill {r.get('SGPRs Spill', 0):2d} occ {r.get('Occupancy [waves/SIMD]', 0)} LDS {r.get('LDS Size [bytes/block]', 0)}")
