"""Persistent engine session against the generic path: own forward / gradient parity, step traces,
step wall time (ResNet-18 workload, default optimizer settings)."""
import os, sys, time, warnings
sys.path.insert(0, os.getcwd())
import torch
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import testproblems as tp, modelprep, curvature
from pytorchhessianfree_amd.session import EngineSession

hf.configure()
dev = "cuda"
seeds = tp.RESNET18_B32_SEPARATED_SEEDS


def make(batch_seed=0):
    model, (x, t), lossf = tp.resnet18_mnist(batch_size=32, device=dev, data_seed=seeds[batch_seed])
    modelprep.prepare_model(model, channels_last=True)
    return model, x, t, lossf


# ---- 1. engine forward / gradient parity -------------------------------------------------
model, x, t, lossf = make()
params = [p for p in model.parameters() if p.requires_grad]
out = model(x)
loss = lossf(out, t)
g_ref = curvature.flatten_into(torch.autograd.grad(loss, params, retain_graph=True), params)
sess = EngineSession.try_create(loss, out, params)
assert sess is not None, "no session"
eng = sess.engine
print("logits err", float((eng.logits - out.detach()).abs().max() / out.detach().abs().max()))
print("loss", float(loss), float(eng.loss_buf))
g = sess.gradient().clone()
print("grad err (max-norm rel)", float((g - g_ref).abs().max() / g_ref.abs().max()))
v = torch.randn(eng.n, device=dev)
op = curvature.GGNOperator(loss, out, params)
want = op(v).clone()
got = sess(v).clone()
print("product err", float((got - want).abs().max() / want.abs().max()))
print("product repeat bitwise", bool(torch.equal(sess(v), got)))
del sess, eng, op, out, loss

# ---- 2. step traces: session vs generic ------------------------------------------------------
def run(session, steps=6):
    os.environ["HF_SESSION"] = "1" if session else "0"
    model, x, t, lossf = make()
    data = [tp.resnet18_mnist(batch_size=32, device=dev, data_seed=seeds[i % len(seeds)])[1] for i in range(steps)]
    opt = hf.HessianFree(model.parameters(), graph_matvec=True)
    times = []
    for i in range(steps):
        xi, ti = data[i]
        def forward():
            o = model(xi); return lossf(o, ti), o
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fl = opt.step(forward)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    st = opt.state
    print(("session" if session else "generic"), "times ms", ["%.1f" % (1e3 * a) for a in times])
    print("  iters", st["num_cg_iters"], "best", st["best_cg_iters"], "lr", st["learning_rates"])
    print("  init", ["%.5f" % a for a in st["init_losses"]], "damp", ["%.4f" % a for a in st["dampings"]], "final %.5f" % fl)
    print("  session active:", opt._session is not None)
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone(), st

pa, sa = run(True)
pb, sb = run(False)
print("param diff after steps (rel max)", float((pa - pb).abs().max() / pb.abs().max()))
