#!/usr/bin/env python
"""Coefficients of ptc_gelu_cdf (pointcept_amd/csrc/ptc_common.h): 0.5 erfc(t) = exp2(t r(t) - 1) on t in [0, 3.95], r a polynomial fitted
(Lawson-reweighted least squares -> near-minimax) to log2(erfc(t)) / t with the weight that turns the residual into the ABSOLUTE error of
erf; prints, per degree, the fit error and the error of the fp32 Horner evaluation (one rounding per fused multiply-add, v_exp_f32 taken
as correctly rounded).  The kernel uses the 8-coefficient fit."""
import numpy as np
from scipy.special import erf, erfc
T=3.95
def fit(deg, iters=40):
    # r(t) of degree deg-1 ; q = t*r(t); minimise max |erfc(t)*(exp2(q_fit-q)-1)| ~ erfc*ln2*t*(r_fit-r)
    t=np.cos(np.pi*(np.arange(4000)+0.5)/4000)*T/2+T/2
    q=np.log2(erfc(t)); r=q/t
    w=erfc(t)*np.log(2)*t
    V=np.vander(t,deg,increasing=True)
    ww=w.copy()
    best=None
    for it in range(iters):
        coef,*_=np.linalg.lstsq(V*ww[:,None], r*ww, rcond=None)
        err=np.abs(w*(V@coef-r))
        m=err.max()
        if best is None or m<best[0]: best=(m,coef.copy())
        ww=ww*(1+2*err/m)   # Lawson-like reweighting toward minimax
        ww/=ww.max()
    return best
def eval32(coef, x):
    # emulate f32 Horner with fma (round once per fma) using float64
    t=np.minimum(np.abs(x.astype(np.float32)),np.float32(T)).astype(np.float64)
    c=[np.float64(np.float32(v)) for v in coef]
    acc=np.full_like(t,c[-1])
    for v in c[-2::-1]:
        acc=np.float64(np.float32(acc*t+v))
    qq=np.float64(np.float32(acc*t))
    e=np.float64(np.float32(np.exp2(qq)))   # v_exp_f32 ~1 ulp
    res=np.float64(np.float32(1.0-e))
    return np.sign(x)*res
x=np.linspace(-6,6,2000001)
for deg in (7,8,9,10,11):
    m,coef=fit(deg)
    e=np.abs(eval32(coef,x)-erf(x)).max()
    print(deg, "fit err",m,"f32 eval max abs err",e)
    if deg in (8,9,10): print(["%.9e"%v for v in coef])
