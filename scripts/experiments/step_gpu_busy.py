"""How much of a default HessianFree.step() the GPU is busy: run under `rocprofv3 --kernel-trace --output-format csv`,
then call with the trace directory to get  sum of kernel durations / wall time  over the steady steps.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python3 scripts/experiments/step_gpu_busy.py run
    python3 scripts/experiments/step_gpu_busy.py report /tmp/st
"""
import csv
import glob
import os
import sys
import time
import warnings

if sys.argv[1] == "run":
    sys.path.insert(0, os.getcwd())
    import torch

    import pytorchhessianfree_amd as hf
    from pytorchhessianfree_amd import modelprep
    from pytorchhessianfree_amd import testproblems as tp

    seeds = tp.RESNET18_B32_SEPARATED_SEEDS
    model, _, lossf = tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=seeds[0])
    modelprep.prepare_model(model, channels_last=True)
    data = [tp.resnet18_mnist(batch_size=32, device="cuda", data_seed=s)[1] for s in seeds]
    opt = hf.HessianFree(model.parameters(), graph_matvec=True)
    marks = []
    for i in range(10):
        x, t = data[i % len(data)]

        def forward():
            o = model(x)
            return lossf(o, t), o

        torch.cuda.synchronize()
        t0 = time.time_ns()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opt.step(forward)
        torch.cuda.synchronize()
        marks.append((t0, time.time_ns()))
    print("STEPS", marks[2:], opt.state["num_cg_iters"][2:], flush=True)
else:
    root = sys.argv[2]
    path = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    # steps are separated by the synchronize() gaps: cluster kernels by gaps > 200 us?  Use busy / span per cluster of
    # kernels between host syncs instead: a step = a maximal run of kernels with gaps < 1 ms, longer than 5 ms
    runs, cur = [], [rows[0]]
    for r in rows[1:]:
        if int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 1_000_000:
            runs.append(cur)
            cur = []
        cur.append(r)
    runs.append(cur)
    for run in runs[-8:]:
        span = (int(run[-1]["End_Timestamp"]) - int(run[0]["Start_Timestamp"])) / 1e6
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in run) / 1e6
        gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"][:40]) for a, b in zip(run, run[1:]))[-6:]
        print(f"span {span:6.2f} ms  busy {busy:6.2f} ms  idle {span - busy:5.2f} ms  kernels {len(run)}  largest gaps (us): {[(round(g), n) for g, n in gaps]}")
