"""(CPU) hipcc's hazard recognizer does not look inside inline asm. A vector instruction written in an asm statement
whose result an MFMA reads fewer than two instructions later is the "VALU write -> MFMA read" hazard unprotected: the
MFMA takes the register's OLD content (found in round 5: the 64-channel forward, right or wrong depending on where the
scheduler happened to put the last v_cvt_pk_f16_f32 of the operand split). This scan compiles the edge kernels as the
Makefile does and lists every MFMA with an asm-written source register at distance < MIN_GAP instructions.
Usage: python tools/asm_mfma_hazard_scan.py [file.hip ...]      exit code 1 if anything is found"""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / 'pointvs_amd' / 'csrc'
FWD = ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1']
FILES = ['edge_mfma_fwd.hip', 'edge_bwd_f16.hip', 'edge_bwd_wide.hip', 'edge_bwd_h64.hip', 'edge_mfma.hip', 'dense_ops.hip']
MIN_GAP = 2          # instructions that must lie between the asm write and the MFMA (LegacyVALUWritesVGPRWaitStates)


def regs(tok):
    tok = tok.strip()
    m = re.match(r'[va]\[(\d+):(\d+)\]', tok)
    if m:
        return {(tok[0], r) for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r'([va])(\d+)$', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def scan(src, extra=()):
    flags = FWD if src == 'edge_mfma_fwd.hip' else []
    out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-value',
                          *flags, *extra, '-S', '--cuda-device-only', str(CSRC / src), '-o', '-'],
                         capture_output=True, text=True, cwd=CSRC)
    if out.returncode:
        sys.exit(out.stderr[-2000:])
    found, kernel, ins, in_asm = [], None, [], False
    for line in out.stdout.splitlines():
        s = line.strip()
        m = re.match(r'^(_Z\w+):', s)
        if m:
            kernel, ins = m.group(1), []
            continue
        if s.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if s.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not s or s.startswith(('.', ';')) or s.endswith(':'):
            continue
        ops = s.split(None, 1)
        if ops[0].startswith('v_mfma') and len(ops) > 1:
            fields = ops[1].split(',')
            srcs = set().union(*[regs(t) for t in fields[1:4]])
            for dist, (asm, text, dst) in enumerate(reversed(ins[-MIN_GAP:])):
                if asm and dst & srcs:
                    between = ' ; '.join(t.split(None, 1)[0] for _, t, _ in ins[len(ins) - dist:]) or '-'
                    found.append((kernel, s[:80], text[:60], between))
        dst = regs(ops[1].split(',')[0]) if len(ops) > 1 and ops[0].startswith(('v_', 'ds_read', 'global_load')) else set()
        ins.append((in_asm, s, dst))
    return found


if __name__ == '__main__':
    files = sys.argv[1:] or FILES
    bad = 0
    for f in files:
        hits = scan(f)
        print(f'{f}: {len(hits)} MFMA(s) read an asm-written register fewer than {MIN_GAP} instructions later')
        for k, mf, asm, between in hits:
            print(f'    {k[:60]}  {mf}  <- {asm}  (between: {between})')
        bad += len(hits)
    sys.exit(1 if bad else 0)
