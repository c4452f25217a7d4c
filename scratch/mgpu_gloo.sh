#!/bin/bash
# 2 ranks sharing the one GPU, gloo backend: exercises bench.py's multi-rank logic
cd $GRAFT_REPO_ROOT
export LOCAL_RANK_OVERRIDE=0
python - <<'PY'
import os, subprocess, sys
procs=[]
for r in range(2):
    env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    procs.append(subprocess.Popen([sys.executable,"bench.py","--gpus","2","--steps","1","--warmup","1","--iters","30","--backend","gloo"]+sys.argv[1:],env=env,stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True))
for r,p in enumerate(procs):
    out,_=p.communicate(timeout=600)
    print("rank",r,"rc",p.returncode); print(out[-1500:])
PY
