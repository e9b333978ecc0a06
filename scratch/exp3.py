import numpy as np, sys
from exp2 import *

class Adaptive:
    def __init__(s, H, A, rhos, sigma=1e-6, alpha=1.6, dscale=1.0):
        n = H.shape[0]; s.n = n; s.H = H
        s.A = A.copy(); s.A[n:] *= dscale; s.dscale = dscale
        s.rhos = rhos; s.sigma, s.alpha = sigma, alpha
        s.Ms = [np.linalg.inv(H + sigma*np.eye(n) + r*s.A.T@s.A) for r in rhos]
        s.Hinv = np.linalg.inv(H); s.G0 = s.A@s.Hinv@s.A.T; s.HiAt = s.Hinv@s.A.T
        s.x = np.zeros(n); s.y = np.zeros(A.shape[0]); s.ir = len(rhos)//2
    def solve(s, g, lo, up, stable=2, maxit=1000, adapt_every=10):
        A = s.A; n = s.n; lo = lo.copy(); up = up.copy(); lo[n:] *= s.dscale; up[n:] *= s.dscale
        sg, al = s.sigma, s.alpha
        Uunc = -s.Hinv@g; v = A@Uunc
        if (v <= up).all() and (v >= lo).all():
            s.x = Uunc; s.y[:] = 0; return Uunc, 0, 0, True
        x = s.x; y = s.y; z = np.clip(A@x, lo, up); ir = s.ir
        last=None; cnt=0; npol=0
        for it in range(1, maxit+1):
            rho = s.rhos[ir]; M = s.Ms[ir]
            xt = M@(sg*x - g + A.T@(rho*z - y)); zt = A@xt
            x = al*xt + (1-al)*x; zh = al*zt + (1-al)*z
            zn = np.clip(zh + y/rho, lo, up); y = y + rho*(zh - zn); z = zn
            act = np.where((z>=up)&(y>0),1,np.where((z<=lo)&(y<0),-1,0))
            if last is not None and (act==last).all(): cnt+=1
            else: cnt=0
            last=act
            if cnt>=stable:
                u,lam,ok = polish(s.H,s.Hinv,s.G0,s.HiAt,g,A,lo,up,act); npol+=1; cnt=0
                if ok:
                    s.x=u; s.y=lam; s.ir=ir; return u,it,npol,True
            if it % adapt_every == 0:
                Ax = A@x
                rp = np.abs(Ax - z).max(); rd = np.abs(s.H@x + g + A.T@y).max()
                np_ = max(np.abs(Ax).max(), np.abs(z).max(), 1e-12); nd = max(np.abs(s.H@x).max(), np.abs(A.T@y).max(), np.abs(g).max(), 1e-12)
                ratio = np.sqrt((rp/np_)/max(rd/nd,1e-30))
                tgt = rho*ratio
                # pick nearest in log
                j = int(np.argmin(np.abs(np.log(s.rhos) - np.log(max(tgt,1e-30)))))
                if abs(j-ir) >= 1: ir = j
        s.x=x; s.y=y; s.ir=ir
        return x,it,npol,False
    def shift(s):
        n=s.n; s.x=np.concatenate([s.x[1:],s.x[-1:]]); yu=s.y[:n]; yd=s.y[n:]
        s.y=np.concatenate([yu[1:],yu[-1:],yd[1:],yd[-1:]])

if __name__=='__main__':
    N,dt,vx = 50,0.01,8.0
    md = build(N,dt,vx); H = md['H']; A,_,_ = cons(md,0.0)
    rhos = 10.0**np.arange(-2,5.01,float(sys.argv[3]) if len(sys.argv)>3 else 0.5)
    dscale = float(sys.argv[4]) if len(sys.argv)>4 else 1.0
    hy = Adaptive(H,A,rhos,dscale=dscale)
    rng = np.random.default_rng(0); nsteps=int(sys.argv[1]); nsc=int(sys.argv[2])
    its=[];pols=[];fails=0
    for scen in range(nsc):
        psi = rng.uniform(-0.1,0.1)
        x = np.array([15.0, rng.uniform(-0.3,0.3), np.cos(psi), np.sin(psi), vx, rng.uniform(-0.2,0.2), rng.uniform(-0.2,0.2), rng.uniform(-0.05,0.05)])
        if scen==0: x = np.array([15.0,0.1,1,0,8.0,0,0,0])
        if scen==1: x = np.array([15.0,-0.3,1,0,8.0,0,0,0.28])
        if scen==2: x = np.array([15.0,0.3,1,0,8.0,0,0,0.28])
        lap=0; pl=Planner(vx,N,dt); hy.x[:]=0; hy.y[:]=0; hy.ir=len(rhos)//2
        for t in range(nsteps):
            hd=np.arctan2(x[3],x[2]); w=pl.wp(x[0],x[1],hd,lap)
            yref=np.zeros((N+1,6)); yref[:,0]=w[:,0]; yref[:,1]=w[:,1]; yref[:,2]=w[:,3]
            xhat=np.array([0,0,0,x[5],x[6],x[7]])
            g,xf=grad(md,xhat,yref); _,lo,up=cons(md,x[7])
            u,it,npol,ok = hy.solve(g,lo,up)
            its.append(it); pols.append(npol); fails += (not ok)
            if (not ok) or it>60 or (t<30 and t%5==0): print(scen,t,'it',it,'npol',npol,'ok',ok,'rho',rhos[hy.ir],'u0',u[0], 'nact', int((np.abs(hy.y)>1e-10).sum()), 'nd', int((np.abs(hy.y[N:])>1e-10).sum()))
            hy.shift()
            xn=rk4(x,u[0],dt)
            if xn[0]>center and x[0]<=center: lap+=1
            x=xn
    its=np.array(its); print('steps',len(its),'mean',its.mean(),'p50',np.median(its),'p99',np.percentile(its,99),'max',its.max(),'polmean',np.mean(pols),'fails',fails,'constr frac',(its>0).mean())
