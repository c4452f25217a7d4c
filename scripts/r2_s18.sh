cd "$GRAFT_REPO_ROOT" || exit 1
echo "auto F:0"; HF_CONV_AUTO="F:0" timeout 600 python scripts/experiments/grad_check.py 2>&1 | grep -v amdgpu.ids | tail -3
python - <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
import pytorchhessianfree_amd as hf
from pytorchhessianfree_amd import modelprep, testproblems as tp
hf.configure()
dev="cuda"
# stem only: conv1 -> bn1 -> relu -> maxpool, NHWC own forward vs stock fp64, forward AND backward
model, (x, t), lossf = tp.resnet18_mnist(32, device=dev, data_seed=1000)
ref, _, _ = tp.resnet18_mnist(32, device=dev, data_seed=1000); ref = ref.double()
modelprep.prepare_model(model, channels_last=True)
def stem(m, xx): return m.maxpool(m.relu(m.bn1(m.conv1(xx))))
y = stem(model, x); yr = stem(ref, x.double())
print("stem fwd err", float((y.double()-yr).abs().max()/yr.abs().max()), y.is_contiguous(memory_format=torch.channels_last))
g = torch.randn_like(yr)
gs = torch.autograd.grad(y, [model.conv1.weight, model.bn1.weight, model.bn1.bias], g.float().contiguous(memory_format=torch.channels_last))
gr = torch.autograd.grad(yr, [ref.conv1.weight, ref.bn1.weight, ref.bn1.bias], g)
for a,b,n in zip(gs, gr, ["conv1.w","bn1.w","bn1.b"]): print(n, float((a.double()-b).abs().max()/b.abs().max()))
# block 0 alone on the pooled input
b0, r0 = model.layers[0], ref.layers[0]
xin = y.detach().requires_grad_(True); xr = yr.detach().requires_grad_(True)
o = b0(xin); orr = r0(xr)
print("block0 fwd err", float((o.double()-orr).abs().max()/orr.abs().max()))
g = torch.randn_like(orr)
ps = [b0.conv1.weight, b0.bn1.bias, b0.conv2.weight, b0.bn2.bias, xin]; pr = [r0.conv1.weight, r0.bn1.bias, r0.conv2.weight, r0.bn2.bias, xr]
gs = torch.autograd.grad(o, ps, g.float().contiguous(memory_format=torch.channels_last)); gr = torch.autograd.grad(orr, pr, g)
for a,b,n in zip(gs, gr, ["c1.w","bn1.b","c2.w","bn2.b","x"]): print(n, float((a.double()-b).abs().max()/b.abs().max()))
PY
