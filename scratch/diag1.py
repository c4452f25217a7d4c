import sys, warnings, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden
from helpers import T, lowrank_operator
import pytorchhessianfree_amd as product
DEV='cuda'
g=load_golden('cg_lowrank.npz')
def rel(a,b): return float(np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30))
for key in [str(k) for k in g['index']]:
    A,B,damping=lowrank_operator(g,key,DEV)
    b=T(g[key+'/b'],DEV); x0=T(g[key+'/x0'],DEV) if key+'/x0' in g else None
    M=product.DiagonalPreconditioner(T(g[key+'/diag'],DEV),damping) if int(g[key+'/precond']) else None
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        gx,gm,reason=product.cg(product.DampedCurvature(B,damping),b,x0=x0,M=M,max_iter=250,martens_conv_crit=True,store_x_at_iters=None)
    X,X64=g[key+'/X'],g[key+'/X64']; m,m64=g[key+'/m'],g[key+'/m64']
    print(key,'gpu iters',len(gx)-1,'ref32',X.shape[0]-1,'ref64',X64.shape[0]-1,reason)
    for i in range(min(len(gx),X.shape[0],X64.shape[0])):
        if gx[i] is not None and not np.isnan(X[i]).any() and not np.isnan(X64[i]).any():
            xg=gx[i].cpu().numpy().astype(np.float64)
            print('  it %3d e_gpu64 %.2e e_ref64 %.2e e_gpuref %.2e | m: gpu-64 %.2e ref-64 %.2e rel m64 %.3e'%(i,rel(xg,X64[i]),rel(X[i].astype(np.float64),X64[i]),rel(xg,X[i].astype(np.float64)),abs(float(gm[i])-m64[i]),abs(m[i]-m64[i]),abs(m64[i])))
g=load_golden('cg_f64.npz')
for key in [str(k) for k in g['index']]:
    A,b=T(g[key+'/A'],DEV),T(g[key+'/b'],DEV); dim=A.shape[0]
    gx,_,reason=product.cg(lambda v:A@v,b,max_iter=10*dim,tol=1e-5,atol=1e-6,store_x_at_iters=list(range(10*dim)))
    X=g[key+'/X']
    k=min(len(gx),X.shape[0])
    print(key,'f64 iters',len(gx)-1,X.shape[0]-1,reason, 'max rel', max(rel(gx[i].cpu().numpy(),X[i]) for i in range(k)))
