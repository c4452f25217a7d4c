cd "$GRAFT_REPO_ROOT" || exit 1
for b in 0 256 512 1024; do echo "== no-reduce blocks=$b"; timeout 300 python scripts/conv_kernel_bench.py --no-reduce 1 --blocks $b 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['layer'], 'T',d['T_own'],'D',d['D_own'],'W',d['W_own'],'| miopen',d['T_miopen'],d['D_miopen'],d['W_miopen'])"; done
