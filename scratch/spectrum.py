import torch, time, sys, warnings
from pytorchhessianfree_amd import testproblems as tp, curvature as cv
from oracle import pcg as oracle
torch.manual_seed(0)
m,(x,t),lf=tp.resnet18_mnist(data_seed=1000)
params=[p for p in m.parameters()]
out=m(x); loss=lf(out,t); print('loss',float(loss), 'out abs max', float(out.abs().max()))
G=cv.GGNOperator(loss,out,params)
g=cv.flatten_into(torch.autograd.grad(loss,params,retain_graph=True),params)
print('grad norm',float(g.norm()))
v=torch.randn(G.n); v/=v.norm()
for i in range(8):
    w=G(v); lam=float(v@w); v=w/w.norm()
print('lambda_max ~',lam)
for damping in [float(a) for a in sys.argv[1:]]:
    tr=oracle._Trace()
    t0=time.time()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        xs,ms,reason=oracle.pcg(lambda u:G(u)+damping*u,-g,max_iter=250,tol=0.0,martens_conv_crit=False,store_x_at_iters=[],trace=tr)
    print('damping',damping,'iters',len(xs)-1,reason,'time',time.time()-t0,'res', ['%.1e'%r for r in tr.res_norm[::25]])
