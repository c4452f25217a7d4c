"""Worker for tests/test_distributed_gpu.py: ONE rank of a 2-rank data-parallel run,
both ranks on ``cuda:0``, talking over gloo (which accepts device tensors; RCCL wants
one device per rank and a 1-GPU box has one).

What it exercises with more than one rank, on the GPU, for the first time:
``HessianFree.step`` / ``acc_step`` with the HIP ``cg()`` -- the lockstep stop rule
(every rank leaves the loop at the same iteration although each decides from its own
device flag), the weighted ``hf_pack`` gather, ONE all-reduce per product, eager and
hipGraph-replayed products (product graph -> all-reduce -> K1-K3 graph).

The problem is the reference's own statement of shard additivity,
``/root/reference/tests/test_optimizer_acc.py:116-175``: chunks of 7 and 8 samples
equal one batch of 15 (golden traces of the real reference: tests/golden/acc_step.npz).
Writes ``<outdir>/rank<r>.npz``.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python dp_two_ranks.py <outdir>
"""

import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import pytorchhessianfree_amd as hf  # noqa: E402
from conftest import load_golden  # noqa: E402
from helpers import T, small_nn, trainable_vec  # noqa: E402

hf.configure()
DEV = "cuda:0"


def main(outdir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.group.WORLD
    g = load_golden("acc_step.npz")
    sizes = [7, 8]
    out = {}
    try:
        for curv, reduction in [("ggn", "mean"), ("ggn", "sum"), ("hessian", "mean")]:
            key = f"{curv}_{reduction}"
            lossf = torch.nn.MSELoss(reduction=reduction)
            weight = sizes[rank] / sum(sizes) if reduction == "mean" else 1.0
            for mode in ("step", "step_graph", "acc_step"):
                model = small_nn(g, key, DEV)
                opt = hf.HessianFree(model.parameters(), curvature_opt=curv, cg_max_iter=4,
                                     process_group=group, shard_weight=weight,
                                     graph_matvec=(mode == "step_graph"))
                params = []
                for s in range(3):
                    inputs = T(g[f"{key}/inputs/{s}/{rank}"], DEV)
                    targets = T(g[f"{key}/targets/{s}/{rank}"], DEV)

                    def forward():
                        o = model(inputs)
                        return lossf(o, targets), o

                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        if mode == "acc_step":
                            opt.acc_step(model, lossf, [(inputs, targets)], reduction=reduction)
                        else:
                            opt.step(forward)
                    params.append(trainable_vec(model).cpu().numpy().copy())
                tag = f"{key}/{mode}/"
                out[tag + "params"] = np.stack(params)
                out[tag + "init_losses"] = np.array(opt.state["init_losses"])
                out[tag + "num_cg_iters"] = np.array(opt.state["num_cg_iters"])
                out[tag + "dampings"] = np.array(opt.state["dampings"])
                out[tag + "learning_rates"] = np.array(opt.state["learning_rates"])
                out[tag + "reasons"] = np.array([str(r) for r in opt.state["cg_reasons"]])

        # ---- the solver alone, long enough for the lockstep rule to matter ------------
        # 2-rank damped low-rank system: rank k holds half of the low-rank factor; the
        # operator ends in an all-reduce (`.group`), so cg() must use its lockstep rule
        # and both ranks must report the same iterate, bitwise.
        gen = torch.Generator().manual_seed(5)
        n, r = 50021, 12
        U = torch.randn(n, r, generator=gen) / r**0.5
        d = torch.rand(n, generator=gen)
        b = torch.randn(n, generator=gen).to(DEV)
        Uk = U[:, rank::world].to(DEV)
        dk = (d / world).to(DEV)

        class Shard:
            group = dist.group.WORLD
            calls = 0

            def __call__(self, v):
                self.calls += 1
                y = dk * v + Uk @ (Uk.T @ v)
                dist.all_reduce(y, group=self.group)
                return y

        op = Shard()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, ms, reason = hf.cg(hf.DampedCurvature(op, 0.05), b, max_iter=250,
                                   martens_conv_crit=True, store_x_at_iters=None)
        out["solver/x"] = xs[-1].cpu().numpy()
        out["solver/n_iters"] = np.array([len(xs) - 1])
        out["solver/calls"] = np.array([op.calls])
        out["solver/reason"] = np.array([reason])
        out["solver/m"] = np.array([float(m) for m in ms])
        # single-process reference of the same system on this GPU
        Ud, dd = U.to(DEV), d.to(DEV)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs1, _, reason1 = hf.cg(hf.DampedCurvature(lambda v: dd * v + Ud @ (Ud.T @ v), 0.05), b,
                                    max_iter=250, martens_conv_crit=True, store_x_at_iters=None)
        out["solver/x_single"] = xs1[-1].cpu().numpy()
        out["solver/n_iters_single"] = np.array([len(xs1) - 1])
        out["solver/reason_single"] = np.array([reason1])
        # ---- the fused engine's product over two ranks: only the entries that can be non-zero
        # travel (compact all-reduce); must equal the plain all-reduce of the local products
        from pytorchhessianfree_amd import curvature, modelprep
        from pytorchhessianfree_amd import testproblems as tp
        from pytorchhessianfree_amd.engine import FusedGGNEngine

        net, (xb, tb), ce = tp.resnet18_mnist(batch_size=6, device=DEV, data_seed=40 + rank)
        modelprep.prepare_model(net, channels_last=True)
        ps = [p for p in net.parameters() if p.requires_grad]
        o = net(xb)
        eng = curvature.ggn_operator(ce(o, tb), o, ps, weight=0.5, group=group)
        assert isinstance(eng, FusedGGNEngine)
        v = torch.randn(eng.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
        plain = eng.local(v).clone()
        dist.all_reduce(plain, group=group)
        got = eng(v).clone()
        out["engine/equal_plain_allreduce"] = np.array([bool(torch.equal(got, plain))])
        out["engine/reduce_bytes"] = np.array([eng.reduce_bytes, 4 * eng.n])
        out["engine/checksum"] = np.array([float(got.double().sum()), float(got.double().abs().max())])
        # ---- the same product with the all-reduce chunked by stage and overlapped (two hipGraphs,
        # session.ChunkedEngineOperator): bitwise the single compact all-reduce, and a solve through it
        from pytorchhessianfree_amd.session import ChunkedEngineOperator

        def builder():
            oo = net(xb)
            return curvature.ggn_operator(ce(oo, tb), oo, ps, weight=0.5, group=group)

        chunked = ChunkedEngineOperator(builder, params=ps)
        got2 = chunked(v).clone()
        out["chunked/equal_plain_allreduce"] = np.array([bool(torch.equal(got2, plain))])
        eng = chunked.engine
        tail = sum(t.numel() for t in eng._reduce_pieces(chunked.output_buffer, "tail"))
        out["chunked/tail_share"] = np.array([4.0 * tail / eng.reduce_bytes])
        out["chunked/pieces"] = np.array([len(eng._reduce_pieces(chunked.output_buffer, "head")),
                                          len(eng._reduce_pieces(chunked.output_buffer, "tail"))])
        grad = torch.randn(eng.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xa, _, ra = hf.cg(hf.DampedCurvature(chunked, 0.1), grad, max_iter=12, tol=0.0, store_x_at_iters=[0])
            xb_, _, rb = hf.cg(hf.DampedCurvature(eng, 0.1), grad, max_iter=12, tol=0.0, store_x_at_iters=[0])
        out["chunked/solve_equal"] = np.array([bool(torch.equal(xa[-1], xb_[-1])) and ra == rb])
        out["chunked/solve_x"] = xa[-1].cpu().numpy()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
