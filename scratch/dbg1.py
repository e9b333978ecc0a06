import numpy as np
from exp2 import *
N, dt, vx = 50, 0.01, 8.0
md = build(N, dt, vx); H = md['H']; A,_,_ = cons(md,0.0)
x = np.array([15.0, -0.3, 1, 0, 8.0, 0, 0, 0.28])
pl = Planner(vx,N,dt); w = pl.wp(x[0],x[1],0.0,0)
yref = np.zeros((N+1,6)); yref[:,0]=w[:,0]; yref[:,1]=w[:,1]; yref[:,2]=w[:,3]
xhat = np.array([0,0,0,x[5],x[6],x[7]])
g,xf = grad(md,xhat,yref); _,lo,up = cons(md,x[7])
print(lo[N:N+3], up[N:N+3])
for rd in (1.0, 100.0, 1e4):
    rv = np.concatenate([np.ones(N), np.full(N-1, rd)])
    xa,za,ya,hist = admm(H,g,A,lo,up,rv,iters=3000)
    v = A@xa
    print('rho_d',rd,'prim viol', max((v-up).max(), (lo-v).max()), 'stat res', np.abs(H@xa+g+A.T@ya).max(), 'u[:6]', xa[:6], 'delta', (0.28+np.cumsum(xa)*dt)[:8])
    print(' act u', hist[-1][:N], ' act d', hist[-1][N:])
