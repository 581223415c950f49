"""Between consecutive MFMAs of one kernel: how many VALU / transcendental / LDS / VMEM / SALU instructions.
Usage: isa_mfma_gaps.py <file.s> <kernel-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
i = m.start()
j = s.index('.end_amdhsa_kernel', i)
TRANS = ('v_exp', 'v_rcp', 'v_sqrt', 'v_rsq', 'v_log')
gap = dict(valu=0, trans=0, ds=0, vmem=0, salu=0, wait=0)
out = []
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith((';', '.')) or t.endswith(':'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'):
        out.append((op.replace('v_mfma_f32_', ''), dict(gap)))
        gap = dict.fromkeys(gap, 0)
    elif op.startswith(TRANS):
        gap['trans'] += 1
    elif op.startswith('v_'):
        gap['valu'] += 1
    elif op.startswith('ds_'):
        gap['ds'] += 1
    elif op.startswith(('global_', 'buffer_', 'scratch_')):
        gap['vmem'] += 1
    elif op == 's_waitcnt':
        gap['wait'] += 1
    elif op.startswith('s_'):
        gap['salu'] += 1
for k, (op, g) in enumerate(out):
    print(f'{k:3d} {op:14s} before: valu {g["valu"]:3d} trans {g["trans"]:2d} ds {g["ds"]:2d} vmem {g["vmem"]:2d} salu {g["salu"]:3d} wait {g["wait"]}')
