"""(CPU) Bank-conflict model of the LDS access patterns of the edge-backward kernels, by the lane-group rules of
MI355X_MICROARCH.md section LDS: a wave64 access is served in fixed lane groups, one LDS cycle per group when no two
DIFFERENT addresses of a group fall on one bank; every further distinct address on a bank adds a cycle.

    instruction            lane groups                                            banks (4-byte words)
    ds_read_b32            2 x 32                                                 32
    ds_read_b64, _tr_b16   2 x 32                                                 64
    ds_read_b128           4 x 16: {0-3,12-15,20-27} {4-11,16-19,28-31} (+32)      64
    ds_write_b32           2 x 32                                                 32
    ds_write_b64           4 x 16 contiguous                                      32
    ds_write_b128          8 x 8 contiguous                                       32

Usage: python tools/lds_conflicts.py            (prints cycles / ideal cycles per access pattern)
The patterns are restated from csrc/edge_bwd_f16.hip, csrc/edge_bwd_h64.hip and csrc/edge_mfma_common.h; a pattern's
address function takes the lane and returns a BYTE address."""
import sys


def groups(kind):
    if kind in ('r32', 'w32', 'r64', 'tr'):
        return [list(range(0, 32)), list(range(32, 64))]
    if kind == 'r128':
        g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
        g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
        return [g0, g1, [l + 32 for l in g0], [l + 32 for l in g1]]
    if kind == 'w64':
        return [list(range(16 * k, 16 * k + 16)) for k in range(4)]
    if kind == 'w128':
        return [list(range(8 * k, 8 * k + 8)) for k in range(8)]
    raise ValueError(kind)


WIDTH = {'r32': 4, 'w32': 4, 'r64': 8, 'tr': 8, 'r128': 16, 'w64': 8, 'w128': 16}
BANKS = {'r32': 32, 'w32': 32, 'r64': 64, 'tr': 64, 'r128': 64, 'w64': 32, 'w128': 32}


def cycles(kind, addr):
    """(LDS cycles, ideal cycles) of one wave instruction; addr(lane) -> byte address."""
    total = 0
    gs = groups(kind)
    for g in gs:
        per_bank = {}
        for lane in g:
            a = addr(lane)
            assert a % WIDTH[kind] == 0 or kind == 'tr', (kind, lane, a)
            for w in range(WIDTH[kind] // 4):
                word = a // 4 + w
                per_bank.setdefault(word % BANKS[kind], set()).add(word)
        total += max(len(v) for v in per_bank.values())
    return total, len(gs)


def img_off_r04(r, c, H=32):
    nch, rpc = H // 4, 128 // H
    return r * H + 4 * ((c >> 2) ^ ((r // rpc) & (nch - 1))) + (c & 3)      # in shorts


def img_off_r05(r, c, H=32):      # H = 32: swz(r) = bits (r2, r3, r1 ^ r4) of the row (edge_mfma_common.h img_off<1>)
    return r * H + 4 * ((c >> 2) ^ (((r >> 2) & 3) | ((((r >> 1) ^ (r >> 4)) & 1) << 2))) + (c & 3)


def img_off_paired(r, c, H=32):     # late round 5: 16-byte pairs of chunks (q, q + 2), permuted by row bits (r2, r1 ^ r3)
    q = c >> 2
    pair, t = ((q >> 2) << 1) | (q & 1), (q >> 1) & 1
    f = ((r >> 2) & 1) | ((((r >> 1) ^ (r >> 3)) & 1) << 1)
    return r * H + 8 * (pair ^ f) + 4 * t + (c & 3)


img_off = img_off_r04


def report(name, kind, addr, per_tile):
    c, ideal = cycles(kind, addr)
    print(f'{name:58s} {kind:5s} {c:3d} / {ideal} cycles   x{per_tile:3d} per tile  -> {c * per_tile:4d} ({(c - ideal) * per_tile:4d} in conflicts)')
    return c * per_tile, (c - ideal) * per_tile


def h32_backward(ts=36, swz_t1=False, paired=False):
    tot = conf = 0
    def add(r):
        nonlocal tot, conf
        tot += r[0]; conf += r[1]
    j = lambda l: l & 31
    hh = lambda l: l >> 5
    # d1b: lane-private float4 at (gq*64 + lane)*4 floats
    add(report('SiLU\'(z1) park, write', 'w128', lambda l: (0 * 64 + l) * 16, 4))
    add(report('SiLU\'(z1) park, read', 'r128', lambda l: (0 * 64 + l) * 16, 4))
    # image writes: part + img_off(j, 16s + 4hh) and (j, 16s + 8 + 4hh), 8 bytes each (paired layout: one 16-byte write)
    for s in (0, 1):
        if paired:
            add(report(f'image write, k-step {s}', 'w128', lambda l: 2 * img_off(j(l), 16 * s + 4 * hh(l)), 4 * 2 // 2))
            continue
        add(report(f'image write, k-step {s}, first half', 'w64', lambda l: 2 * img_off(j(l), 16 * s + 4 * hh(l)), 4 * 2 // 2))
        add(report(f'image write, k-step {s}, second half', 'w64', lambda l: 2 * img_off(j(l), 16 * s + 8 + 4 * hh(l)), 4 * 2 // 2))
    # weight row reads (W v): row j, columns 16s + 4hh (+8)
    for s in (0, 1):
        if paired:
            add(report(f'weight fragment rows, k-step {s}', 'r128', lambda l: 2 * img_off(j(l), 16 * s + 4 * hh(l)), 2 * 2 // 2))
            continue
        add(report(f'weight fragment rows, k-step {s}', 'r64', lambda l: 2 * img_off(j(l), 16 * s + 4 * hh(l)), 2 * 2 * 2 // 2))
    # transposed reads (W^T v and both weight-gradient operands)
    def tr_addr(s, second):
        def f(l):
            li, q, p = l & 15, (l & 15) >> 2, l & 3
            r0 = 16 * s + 4 * hh(l) + (8 if second else 0)
            col = 16 * ((l >> 4) & 1) + 4 * p
            return 2 * img_off(r0 + q, col)
        return f
    for s in (0, 1):
        for second in (False, True):
            add(report(f'transposed fragment, k-step {s}, rows +{8 if second else 0}', 'tr', tr_addr(s, second), (8 + 32) // 4))
    # per-channel tables: every lane of a half reads the same float4 (broadcast)
    add(report('per-channel table (broadcast)', 'r128', lambda l: 16 * hh(l), 28))
    # g_z1 tile: write float4 at T1 + j*TS + 8gq + 4hh (floats)
    if swz_t1:
        w = lambda gq: (lambda l: 4 * (j(l) * ts + 4 * (((2 * gq + hh(l)) ^ (j(l) & 7)))))
        rd = lambda k: (lambda l: 4 * ((k * 8 + l // 8) * ts + 4 * ((l % 8) ^ ((k * 8 + l // 8) & 7))))
    else:
        w = lambda gq: (lambda l: 4 * (j(l) * ts + 8 * gq + 4 * hh(l)))
        rd = lambda k: (lambda l: 4 * ((k * 8 + l // 8) * ts + 4 * (l % 8)))
    for gq in range(4):
        add(report(f'g_z1 tile write, quad {gq}', 'w128', w(gq), 1))
    for k in range(4):
        add(report(f'g_z1 tile read, rows {8 * k}..', 'r128', rd(k), 1))
    base_tx = 32 * ts * 4
    add(report('per-edge float4 (tx) write (32 lanes)', 'w128', lambda l: base_tx + 16 * j(l), 1))
    for k in range(4):
        add(report(f'per-edge float4 (tx) read, rows {8 * k}..', 'r128', lambda l: base_tx + 16 * (k * 8 + l // 8), 1))
    print(f'total {tot} LDS cycles per tile, {conf} of them conflicts ({100.0 * conf / tot:.0f} %)\n')
    return tot, conf


if __name__ == '__main__':
    print('== H = 32 backward (k_edge_bwd_f16) as of round 4: image swizzle (r / 4) & 7, g_z1 tile stride 36 floats ==')
    h32_backward(36)
    print('== round 5: image swizzle bits (r2, r3, r1 ^ r4), g_z1 tile unpadded with the quads of row r rotated by r & 7 ==')
    img_off = img_off_r05
    h32_backward(32, swz_t1=True)
    print('== late round 5: the two chunks of a fragment side by side (one 16-byte access), pairs permuted by (r2, r1 ^ r3) ==')
    img_off = img_off_paired
    h32_backward(32, swz_t1=True, paired=True)
