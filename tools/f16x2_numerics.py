"""Emulation of the f16x2 product (edge_mfma_common.h) against fp64: error relative to sum|a||b| and to max|ref|,
next to a sequential fp32 FMA chain and the bf16x3 6-term product. CPU only; numpy."""
import numpy as np
rng=np.random.default_rng(0)
def split_f16(x, s):
    xs=(x.astype(np.float32)*np.float32(s)).astype(np.float32)
    hi=xs.astype(np.float16)
    lo=(xs-hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)
def scale_of(x):
    m=np.abs(x).max(); e=int(np.frexp(np.float32(m))[1])-1+127  # biased exponent
    eq=max(e,16); return 2.0**(140-eq)
def bf16_trunc(x):
    u=x.astype(np.float32).view(np.uint32)&0xffff0000
    return u.view(np.float32)
def split_bf16x3(x):
    x=x.astype(np.float32); h=bf16_trunc(x); r=(x-h).astype(np.float32); m=bf16_trunc(r); t=(r-m).astype(np.float32); l=bf16_trunc(t)
    return h.astype(np.float64),m.astype(np.float64),l.astype(np.float64)
for trial,(sa,sw) in enumerate([(1,0.18),(1e3,0.18),(1e-7,0.18),(1,1e-3)]):
    K=32
    W=(rng.uniform(-1,1,(32,K))*sw).astype(np.float32)
    # activations with wide dynamic range within the tile
    A=(rng.normal(size=(K,32))*sa*np.exp(rng.normal(size=(K,32))*2)).astype(np.float32)
    ref=W.astype(np.float64)@A.astype(np.float64)
    f32=(W@A)  # fp32 (blas, order differs)
    # sequential fp32 accumulate like fma chain
    acc=np.zeros((32,32),np.float32)
    for k in range(K): acc=(acc+W[:,k:k+1]*A[k:k+1,:]).astype(np.float32)
    sW=scale_of(W); sA=scale_of(A)
    wh,wl=split_f16(W,sW); ah,al=split_f16(A,sA)
    p=(wl@ah+wh@al+wh@ah)/(sW*sA)
    h,m,l=split_bf16x3(W); bh,bm,bl=split_bf16x3(A)
    p3=l@bh+h@bl+m@bm+m@bh+h@bm+h@bh
    den=np.abs(W).astype(np.float64)@np.abs(A).astype(np.float64)
    e=lambda x: (np.abs(x-ref)/den).max()
    em=lambda x: np.abs(x-ref).max()/np.abs(ref).max()
    print(f'trial {trial}: fp32 {e(acc):.2e} / {em(acc):.2e}   f16x2 {e(p):.2e} / {em(p):.2e}   bf16x3 {e(p3):.2e} / {em(p3):.2e}')
