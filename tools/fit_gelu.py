import numpy as np
from scipy.optimize import linprog
from scipy.special import erf
def fit(deg, L, xmax=12.0, n=4000):
    u = np.linspace(0, L, n)
    # unknown c[0..deg], E.  approx Phi(u)-0.5 = u*q(u^2), q = sum c_k t^k
    t = u*u
    A = np.stack([u * t**k for k in range(deg+1)], 1)          # [n, deg+1]
    f = 0.5*erf(u/np.sqrt(2))
    # gelu error = x * (A c - f), x=u
    W = u[:,None]
    rows=[]; rhs=[]
    G = W*A; g = u*f
    # tail: x in (L, xmax]: x*(A_L c + 0.5 - Phi(x))
    xt = np.linspace(L, xmax, 800)
    AL = A[-1]
    Gt = xt[:,None]*AL[None,:]; gt = xt*(0.5*erf(xt/np.sqrt(2)))
    Gall = np.concatenate([G,Gt]); gall=np.concatenate([g,gt])
    m = Gall.shape[0]
    # minimize E s.t. Gall c - gall <= E, -(Gall c - gall) <= E
    Aub = np.concatenate([np.concatenate([Gall, -np.ones((m,1))],1), np.concatenate([-Gall, -np.ones((m,1))],1)])
    bub = np.concatenate([gall, -gall])
    cost = np.zeros(deg+2); cost[-1]=1
    # scale columns for conditioning
    sc = np.array([L**(2*k+1) for k in range(deg+1)]+[1.0])
    r = linprog(cost, A_ub=Aub/ sc[None,:]*1.0, b_ub=bub, bounds=[(None,None)]*(deg+2), method="highs")
    c = r.x[:-1]/sc[:-1]
    return c, r.x[-1]
def check(c, L):
    x = np.linspace(-12,12,200001).astype(np.float32)
    u = np.clip(x,-L,L).astype(np.float32); t=(u*u).astype(np.float32)
    p = np.float32(c[-1])*t + np.float32(c[-2])
    for k in range(len(c)-3,-1,-1): p = p*t+np.float32(c[k])
    y = x*(u*p+np.float32(0.5))
    ex = 0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
    return np.abs(y-ex).max()
for deg,Ls in ((6,(3.9,)),(7,(3.9,4.0,4.05,4.1,4.2)),(8,(4.2,))):
    for L in Ls:
        c,E = fit(deg,L)
        print(deg,L,"minimax E",E,"fp32 check",check(c,L))
        print("   ", ", ".join(f"{v:.9e}f" for v in c))
c6=[3.986083969e-01, -6.556460269e-02, 9.218763890e-03, -9.056357457e-04, 5.740218682e-05, -2.075315505e-06, 3.214920233e-08]
c8=[3.989074382e-01, -6.636037144e-02, 9.830130026e-03, -1.114147779e-03, 9.457434347e-05, -5.760762241e-06, 2.343669162e-07, -5.633299462e-09, 5.998041464e-11]
print("existing c6", check(c6,3.9), "c8", check(c8,4.2))
