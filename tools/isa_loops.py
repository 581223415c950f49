"""Instruction mix of the largest loops of one kernel in a hipcc -save-temps .s file.
Usage: isa_loops.py <file.s> <kernel-name-substring> [n_loops]"""
import collections
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
        back.append((labels[mm.group(1)], k, mm.group(1)))
back.sort(key=lambda t: -(t[1] - t[0]))
TRANS = ('v_exp', 'v_rcp', 'v_sqrt', 'v_rsq', 'v_log')
for kk, k, tgt in back[:int(sys.argv[3]) if len(sys.argv) > 3 else 3]:
    seg = [l.strip().split()[0] for l in lines[kk:k]
           if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    c = collections.Counter(seg)
    grp = lambda pred: sum(v for n, v in c.items() if pred(n))
    print(f'{tgt}: {len(seg)} instr | VALU {grp(lambda n: n.startswith("v_") and not n.startswith("v_mfma"))}'
          f' (trans {grp(lambda n: n.startswith(TRANS))}, pk {grp(lambda n: n.startswith("v_pk_"))})'
          f' | mfma {grp(lambda n: n.startswith("v_mfma"))} | ds {grp(lambda n: n.startswith("ds_"))}'
          f' | vmem {grp(lambda n: n.startswith(("global_", "buffer_", "scratch_")))} | salu {grp(lambda n: n.startswith("s_"))}')
    print('    ', ', '.join(f'{n} {v}' for n, v in c.most_common(28)))
