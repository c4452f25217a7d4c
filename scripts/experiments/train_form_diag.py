"""Round-5 diagnosis of GPUTEST_r04's red test (train-mode BatchNorm finalisation forms: `barrier` differed from
`separate` by 3.65e-6 on the driver's box, <= 1e-6 on the builder's).  Per form, on fresh models with the same seeds:
checksums of what the engine linearises at (recorded activations, batch statistics, ReLU masks), the product's
delta to the first `separate` product, and the same form built TWICE (is the input side reproducible between model
instances of one process?).  Prints one JSON line per engine; no assertions."""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytorchhessianfree_amd as hf  # noqa: E402
from pytorchhessianfree_amd import curvature, modelprep, testproblems as tp  # noqa: E402

hf.configure()
DEV = "cuda:0"


def h(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:10]


def main():
    props = torch.cuda.get_device_properties(0)
    print(json.dumps({"device": props.name, "cus": props.multi_processor_count,
                      "threads": torch.get_num_threads()}), flush=True)
    import platform
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    cm, (cx, ct), _ = tp.resnet18_mnist(batch_size=16, device="cpu", data_seed=tp.RESNET18_B32_SEPARATED_SEEDS[2])
    print(json.dumps({"cpu": cpu[0], "ncpu": len(cpu), "cpu_caps": torch.backends.cpu.get_cpu_capability(),
                      "init_weights_cpu": h(torch.cat([p.reshape(-1) for p in cm.parameters()])),
                      "inputs_cpu": h(cx), "randn_cpu": h(torch.randn(100000, generator=torch.Generator().manual_seed(3))),
                      "randn_gpu": h(torch.randn(100000, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)))}),
          flush=True)
    seeds = [tp.RESNET18_B32_SEPARATED_SEEDS[i] for i in (2, 0, 1, 3, 4, 5)] + [7, 8, 9, 10]
    for seed in seeds:
        one_seed(seed, ["separate", "barrier", "tail", "prologue"] + (["separate", "barrier"] if seed == seeds[0] else []))


def one_seed(seed, order):
    v = None
    ref = None
    first = {}
    for form in order:
        os.environ["HF_BN_TRAIN_FORM"] = form
        model, (x, t), lossf = tp.resnet18_mnist(batch_size=16, device=DEV, data_seed=seed)
        model.train()
        modelprep.prepare_model(model, channels_last=True)
        params = [p for p in model.parameters() if p.requires_grad]
        out = model(x)
        op = curvature.ggn_operator(lossf(out, t), out, params)
        if v is None:
            v = torch.randn(op.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(41))
        got = op(v).clone()
        rep = all(torch.equal(op(v), got) for _ in range(5))
        rec = {
            "seed": seed, "form": form, "logits": h(out), "x": h(x),
            "mean_t": h(torch.cat([u.mean_t for u in op.units])),
            "rstd": h(torch.cat([u.rstd for u in op.units])),
            "masks": h(torch.cat([(u.y > 0).reshape(-1) for u in op.units if u.relu])),
            "a": h(torch.cat([u.a.reshape(-1) for u in op.units])),
            "rb": [u.rb for u in op.units], "fuse": sum(bool(u.fuse) for u in op.units),
            "product": h(got), "repeatable": rep,
        }
        if ref is None:
            ref = got
        rec["delta_to_first_separate"] = float((got - ref).abs().max() / ref.abs().max())
        if form in first:
            rec["delta_to_same_form_before"] = float((got - first[form]).abs().max() / ref.abs().max())
        else:
            first[form] = got
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
