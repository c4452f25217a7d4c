#!/bin/bash
# where does an iteration's wall time go: kernel busy time vs gaps (graph start, PCG launches)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/gap.log 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob('gpurun_out/gap/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
k2 = [i for i, r in enumerate(rows) if 'k_update_xr' in r['Kernel_Name']]
# take iterations 300..400 of the trace (inside the timed solve)
sel = k2[350:450]
busy = gaps_pre_k1 = gaps_post_k3 = gaps_inner = 0.0
n = 0
for a, b in zip(sel[:-1], sel[1:]):
    it = rows[a:b]  # from K2 of iteration i to K2 of i+1 (exclusive)
    t0, t1 = int(it[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    bz = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in it)
    busy += bz
    # gap after K3 (index 1 of it) before first graph kernel (index 2)
    gaps_post_k3 += int(it[2]['Start_Timestamp']) - int(it[1]['End_Timestamp'])
    # gap between last graph kernel (k_pack) and K1
    kp = [j for j, r in enumerate(it) if 'k_pack' in r['Kernel_Name']][-1]
    gaps_pre_k1 += int(it[kp + 1]['Start_Timestamp']) - int(it[kp]['End_Timestamp'])
    gaps_inner += (t1 - t0) - bz
    n += 1
print("per iteration (us): wall %.1f busy %.1f all gaps %.1f | K3->graph first kernel %.1f | k_pack->K1 %.1f | kernels %.1f" % (
    (int(rows[sel[-1]]['Start_Timestamp']) - int(rows[sel[0]]['Start_Timestamp'])) / n / 1e3, busy / n / 1e3, gaps_inner / n / 1e3,
    gaps_post_k3 / n / 1e3, gaps_pre_k1 / n / 1e3, (sel[-1] - sel[0]) / n))
P
rm -rf gpurun_out/gap
