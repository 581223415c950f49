"""Condensed instruction timeline of the largest loop of one kernel in a hipcc -save-temps .s file: runs of
instruction classes (V = VALU, T = transcendental, M = MFMA, L = LDS, G = global/scratch memory, S = scalar,
W(...) = s_waitcnt, B = branch, | = basic-block label). Shows where the MFMAs sit between vector work and waits.
Usage: isa_timeline.py <file.s> <kernel-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(sub) + r'\S*):', s, re.M)
i = m.start()
j = s.index('.end_amdhsa_kernel', i)
lines = s[i:j].split('\n')
labels = {l.split(':')[0]: k for k, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
back = []
for k, l in enumerate(lines):
    mm = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        back.append((labels[mm.group(1)], k))
kk, k = max(back, key=lambda t: t[1] - t[0])
TRANS = ('v_exp', 'v_rcp', 'v_sqrt', 'v_rsq', 'v_log')
out, cur, n = [], None, 0


def flush():
    global cur, n
    if cur:
        out.append(f'{cur}{n}')
    cur, n = None, 0


for l in lines[kk:k + 1]:
    t = l.strip()
    if not t or t.startswith((';', '.')) and not re.match(r'^\.LBB', t):
        continue
    if re.match(r'^\.LBB\d+_\d+:', t):
        flush(); out.append('|'); continue
    op = t.split()[0]
    if op == 's_waitcnt':
        flush(); out.append('W(' + ' '.join(t.split()[1:]).replace('lgkmcnt', 'l').replace('vmcnt', 'v') + ')'); continue
    if op.startswith(('s_cbranch', 's_branch')):
        flush(); out.append('B'); continue
    c = ('M' if op.startswith('v_mfma') else 'T' if op.startswith(TRANS) else 'V' if op.startswith('v_') else
         'L' if op.startswith('ds_') else 'G' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'S')
    if c != cur:
        flush(); cur = c
    n += 1
flush()
print(f'loop of {k - kk} lines')
print(' '.join(out))
