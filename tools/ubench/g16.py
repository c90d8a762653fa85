import numpy as np, math
from scipy.special import erf
C=[3.980601132e-01,-6.438287348e-02,8.499878459e-03,-7.195603685e-04,3.409395140e-05,-6.780236390e-07]
def bf16(x):
    x=np.asarray(x,np.float32); u=x.view(np.uint32); u=u+0x7fff+((u>>16)&1); return (u&0xffff0000).view(np.float32)
def gelu_ref(x): return 0.5*x*(1+erf(x/np.sqrt(2)))
def cur(x):   # fp32 deg5 -> bf16
    x=x.astype(np.float32); xc=np.clip(x,-3.5,3.5); u=xc*xc; q=np.float32(C[5])
    for c in C[4::-1]: q=(q*u+np.float32(c)).astype(np.float32)
    return bf16(x*(xc*q+np.float32(0.5)))
h=np.float16
def fma16(a,b,c): return (a.astype(np.float32)*b.astype(np.float32)+c.astype(np.float32)).astype(h)   # single rounding (f32 product of two f16 is exact)
def new(x, scale=4.0):  # x' = x/scale from the MFMA, fp16 math
    xp=(x/scale).astype(np.float32)
    # cvt rtz
    xh=xp.astype(h); 
    bad=np.abs(xh.astype(np.float32))>np.abs(xp); xh=np.where(bad, np.nextafter(xh, h(0)), xh).astype(h)
    u=(xh.astype(np.float32)*xh.astype(np.float32)).astype(h)
    umax=h((3.5/scale)**2); u=np.minimum(u,umax)
    ck=[h(c*scale*(scale*scale)**k) for k,c in enumerate(C)]
    q=np.full_like(u,ck[5])
    for c in ck[4::-1]: q=fma16(q,u,np.full_like(u,c))
    phi=np.clip(fma16(xh,q,np.full_like(u,h(0.5))).astype(np.float32),0,1).astype(h)
    y=(xh.astype(np.float32)*phi.astype(np.float32)).astype(h)
    return y.astype(np.float32)*scale, ck
rng=np.random.default_rng(0)
for sig in (0.3,1,2,4,8):
    x=rng.normal(0,sig,2_000_00).astype(np.float32)
    r=gelu_ref(x.astype(np.float64)); 
    a=cur(x); b,ck=new(x)
    rms=np.sqrt(np.mean(r*r))
    neg=x<-1
    print(f"sigma {sig}: cur rel {np.sqrt(np.mean((a-r)**2))/rms:.2e} max {np.abs(a-r).max():.2e} | f16 rel {np.sqrt(np.mean((b-r)**2))/rms:.2e} max {np.abs(b-r).max():.2e} | x<-1: cur rms {np.sqrt(np.mean((a-r)[neg]**2)):.2e} f16 rms {np.sqrt(np.mean((b-r)[neg]**2)):.2e}")
print([float(c) for c in ck])
