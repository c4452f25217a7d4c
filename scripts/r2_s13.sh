cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python3 $R/scripts/experiments/engine_check.py > /tmp/e.out 2>&1)
tail -2 /tmp/e.out
f=$(find /tmp/prof_e -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
names=[(r["Kernel_Name"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r.get("Grid_Size") or r.get("Grid_Size_X")) for r in rows]
names.sort(key=lambda t:t[2])
idx=[i for i,t in enumerate(names) if "k_unpack_tangent" in t[0]]
a,b=idx[-2],idx[-1]
tot=0
def short(n):
    n=re.sub(r"\(anonymous namespace\)::","",n); n=n.replace("void ","")
    m=re.match(r"([A-Za-z_0-9:]+)",n); return (m.group(1) if m else n)[:34]
agg={}
for i in range(a,b):
    n,d,s,g=names[i]; tot+=d
    k=short(n); agg.setdefault(k,[0,0.0]); agg[k][0]+=1; agg[k][1]+=d/1e3
    print(f"{k:36s} grid={g:>8} dur={d/1e3:6.2f}")
print("kernel time in one product (us):", tot/1e3, " wall:", (names[b][2]-names[a][2])/1e3, "launches", b-a)
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]): print(f"{k:36s} x{c:3d} {t:7.1f} us")
PY
