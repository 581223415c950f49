"""Lane-level numpy model of v_mfma_f32_32x32x2_f32 used to validate the register/LDS layouts of
the MFMA edge kernels before they are written in HIP (no GPU needed).

Maps (cdna_hip_programming.md §3): A operand lane l holds A[i=l&31][k=l>>5]; B operand lane l holds
B[k=l>>5][j=l&31]; C/D register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), column l&31.
"""
import numpy as np

L = np.arange(64)
J, HH = L & 31, L >> 5


def ch(t, hh):
    return (t & 3) + 8 * (t >> 2) + 4 * hh


def mfma(a, b, c):
    """a, b: [64] lane values; c: [16,64] accumulator registers. Returns d [16,64]."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[J, HH] = a
    B[HH, J] = b
    D = A @ B
    d = c.copy()
    for r in range(16):
        d[r] += D[ch(r, HH), J]
    return d


def to_x(V):
    """V [32 ch, 32 edges] -> X-layout registers [16,64]."""
    x = np.zeros((16, 64))
    for t in range(16):
        x[t] = V[ch(t, HH), J]
    return x


def from_x(x):
    V = np.zeros((32, 32))
    for t in range(16):
        V[ch(t, HH), J] = x[t]
    return V


rng = np.random.default_rng(0)
W = rng.normal(size=(32, 32)); V = rng.normal(size=(32, 32)); bias = rng.normal(size=32)

# forward-type GEMM Z = W V + b with V as B operand straight from X-layout registers
stage = np.zeros((16, 64))
for t in range(16):
    stage[t] = W[J, ch(t, HH)]            # Wlds[t][l] = W[out=l&31][in=ch(t, l>>5)]
acc = np.zeros((16, 64))
for r in range(16):
    acc[r] = bias[ch(r, HH)]
vx = to_x(V)
for t in range(16):
    acc = mfma(stage[t], vx[t], acc)
assert np.allclose(from_x(acc), W @ V + bias[:, None]), 'forward chain'

# dgrad G_in = W^T G_out : same staging with W^T
stage_t = np.zeros((16, 64))
for t in range(16):
    stage_t[t] = W.T[J, ch(t, HH)]         # = W[ch(t,hh)][l&31]
acc = np.zeros((16, 64))
for t in range(16):
    acc = mfma(stage_t[t], vx[t], acc)
assert np.allclose(from_x(acc), W.T @ V), 'dgrad chain'

# wgrad gW[c][k] = sum_e G[c][e] A1[k][e] from an LDS tile T[e][c] (row-major per edge):
# A operand lane l: G[c=l&31][e=2s+(l>>5)], B operand lane l: A1[k=l&31][e=2s+(l>>5)]
G = rng.normal(size=(32, 32)); A1 = rng.normal(size=(32, 32))
TG, TA = G.T.copy(), A1.T.copy()           # [e][c]
acc = np.zeros((16, 64))
for s in range(16):
    acc = mfma(TG[2 * s + HH, J], TA[2 * s + HH, J], acc)
gW = from_x(acc)                            # rows = c (ch(r,hh)), cols = k (lane&31)
assert np.allclose(gW, G @ A1.T), 'wgrad'

# X-layout global access pattern: lane reads 4 floats at channel 8g+4hh for g=0..3 => regs 4g..4g+3
for g in range(4):
    for q in range(4):
        assert np.all(ch(4 * g + q, HH) == 8 * g + 4 * HH + q)
print('mfma layout checks passed')
