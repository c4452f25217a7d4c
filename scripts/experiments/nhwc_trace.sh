#!/bin/bash
# ordered kernel list of ONE replayed NHWC product (immediate mode, warm db)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DB=$GRAFT_REPO_ROOT/gpurun_out/trace_db; rm -rf $DB; mkdir -p $DB; cp pytorchhessianfree_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
rm -rf gpurun_out/trace_cl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_cl -- python3 scratch/nhwc_mode.py 0 > gpurun_out/trace_cl.log 2>&1
grep RESULT gpurun_out/trace_cl.log
python3 - <<'P'
import csv, glob
f = glob.glob('gpurun_out/trace_cl/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last product = kernels after the second-to-last k_pack
idx = [i for i, r in enumerate(rows) if 'k_pack' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
out = open('gpurun_out/trace_cl_one_product.txt', 'w')
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write("%9.2f %7.2f  g=%s w=%s  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'), r['Kernel_Name'][:150]))
out.close()
print("kernels in one product:", b - a, "span us:", (int(rows[b-1]['End_Timestamp']) - t0) / 1e3)
P
find gpurun_out/trace_cl -name "*kernel_trace.csv" -delete
