cd "$GRAFT_REPO_ROOT" || exit 1
for f in "" "F:0"; do for s in 1000 0 1; do
echo "== HF_CONV_AUTO='$f' seed $s"
HF_CONV_AUTO="$f" DATA_SEED=$s timeout 300 python scripts/experiments/mask_check.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn"
done; done
