import numpy as np
from scipy.optimize import linprog
def fit(c, nterms, n=4001):
    x=np.linspace(1e-4,c,n); u=x*x
    A=np.stack([x*u**k for k in range(nterms)],1)   # x * sum c_k u^k
    y=np.tanh(x)
    # variables: coeffs (nterms), t
    rows=[];rhs=[]
    Aub=np.vstack([np.hstack([A,-np.ones((n,1))]), np.hstack([-A,-np.ones((n,1))])])
    bub=np.concatenate([y,-y])
    # tail: |1 - p(c)| <= t  (p(c) is last row of A)
    a=A[-1]
    Aub=np.vstack([Aub, np.hstack([-a,[-1]]), np.hstack([a,[-1]])])
    bub=np.concatenate([bub,[-1.0],[1.0]])
    cost=np.zeros(nterms+1); cost[-1]=1
    r=linprog(cost,A_ub=Aub,b_ub=bub,bounds=[(None,None)]*nterms+[(0,None)],method='highs')
    return r.x[:-1], r.x[-1]
def eval16(co,c,x):
    x=np.clip(x,-c,c).astype(np.float16)
    u=(x*x).astype(np.float16)
    p=np.float16(co[-1])*np.ones_like(x)
    for k in range(len(co)-2,-1,-1):
        p=(p.astype(np.float32)*u.astype(np.float32)+np.float32(np.float16(co[k]))).astype(np.float16)   # fma: one rounding
    return (p.astype(np.float32)*x.astype(np.float32)).astype(np.float16).astype(np.float64)
best=None
for nt in (5,6,7):
  for c in np.arange(2.8,4.01,0.1):
    co,t=fit(c,nt)
    xs=np.linspace(-6,6,200001)
    e16=np.abs(eval16(co,c,xs)-np.tanh(xs)).max()
    print(nt, round(c,2), 'minimax %.2e'%t, 'f16 eval max err %.2e'%e16, np.array2string(co,precision=6))
print("---- fp32 evaluation")
def eval32(co,c,x):
    x=np.clip(x,-c,c).astype(np.float32); u=x*x
    p=np.float32(co[-1])*np.ones_like(x)
    for k in range(len(co)-2,-1,-1):
        p=(p.astype(np.float64)*u+np.float64(np.float32(co[k]))).astype(np.float32)
    return (p*x).astype(np.float64)
for nt,c in ((6,3.3),(7,3.3),(7,3.2),(8,3.6),(8,3.8)):
    co,t=fit(c,nt); xs=np.linspace(-8,8,400001)
    print(nt,c,'minimax %.3e'%t,'fp32 err %.3e'%np.abs(eval32(co,c,xs)-np.tanh(xs)).max(), [float(np.float32(v)) for v in co])
# the rsq form: tanh x = x rsq(g(x^2)), g(u) = u / tanh^2(sqrt u)
for deg in (2,3,4):
  for c in (3.5,4.0,5.0):
    x=np.linspace(1e-3,c,4001); u=x*x; g=u/np.tanh(x)**2
    A=np.stack([u**k for k in range(deg+1)],1)
    # relative minimax fit of g (error in tanh = 0.5 * relative error of g * tanh)
    W=1.0/g
    n=len(x)
    Aub=np.vstack([np.hstack([A*W[:,None],-np.ones((n,1))]), np.hstack([-A*W[:,None],-np.ones((n,1))])]); bub=np.concatenate([np.ones(n),-np.ones(n)])
    cost=np.zeros(deg+2); cost[-1]=1
    r=linprog(cost,A_ub=Aub,b_ub=bub,bounds=[(None,None)]*(deg+1)+[(0,None)],method='highs')
    co=r.x[:-1]; xs=np.linspace(-8,8,400001); xc=np.clip(xs,-c,c); gg=sum(co[k]*(xc*xc)**k for k in range(deg+1))
    print('rsq deg',deg,'c',c,'err %.2e'%np.abs(xc/np.sqrt(gg)-np.tanh(xs)).max())
