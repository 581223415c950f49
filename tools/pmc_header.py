#!/usr/bin/env python3
"""Prepend the derived fractions (VALU / matrix-pipe busy, per-wave issue / stall / wait shares, LDS) to a
tools/pmc_summary.py file.   usage: tools/pmc_header.py <file> <config-name>"""
import re
import sys

p, cfg = sys.argv[1], sys.argv[2]
txt = open(p).read()
if txt.startswith('SQ counters of'):
    txt = txt[txt.index('\n\n') + 2:]
blocks = re.split(r'\n(?=k_)', txt)
hdr = [f"SQ counters of the edge kernels and the column gather, {cfg} (tools/pmc_sq.sh: three 8-counter passes of bench.py --steps 1 --warmup 1;",
       "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (sum over 8 XCDs) cycles).",
       "Derived per kernel (1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8):"]
for b in blocks:
    m = re.match(r'(k_\S+.*?)\s+dispatches=(\d+)', b)
    if not m:
        continue
    vals = {k: float(v) for k, v in re.findall(r'(\w+)\s+total \S+\s+per-dispatch (\S+)', b)}
    cyc = vals.get('GRBM_GUI_ACTIVE', 0) / 8
    if not cyc:
        continue
    valu = vals.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / cyc
    mfma = vals.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc
    co = vals.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / max(vals.get('SQ_VALU_MFMA_BUSY_CYCLES', 1), 1)
    wc = vals.get('SQ_WAVE_CYCLES', 1)
    wait, stall, act = vals.get('SQ_WAIT_ANY', 0) / wc, vals.get('SQ_WAIT_INST_ANY', 0) / wc, vals.get('SQ_ACTIVE_INST_ANY', 0) / wc
    lds = vals.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / cyc
    conf = vals.get('SQ_LDS_BANK_CONFLICT', 0) / max(vals.get('SQ_LDS_IDX_ACTIVE', 1), 1)
    hdr.append(f"  {m.group(1)}: kernel {cyc:.3e} cycles per launch; VALU busy {valu:.0%} of SIMD time, matrix pipe {mfma:.0%} ({co:.0%} of it under VALU work); "
               f"per wave: issuing {act:.0%}, issue-stalled {stall:.0%}, in s_waitcnt {wait:.0%}; LDS busy {lds:.0%} of CU time ({conf:.0%} of it bank conflicts); "
               f"VALU instructions {vals.get('SQ_INSTS_VALU', 0):.3e}, MFMA {vals.get('SQ_INSTS_MFMA', 0):.3e}, LDS {vals.get('SQ_INSTS_LDS', 0):.3e} per launch")
open(p, 'w').write('\n'.join(hdr) + '\n\n' + txt)
print('\n'.join(hdr))
