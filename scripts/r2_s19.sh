cd "$GRAFT_REPO_ROOT" || exit 1
for f in "" "F:0"; do for i in 1 0; do
echo "== HF_CONV_AUTO='$f' NHWC_IN=$i"
HF_CONV_AUTO="$f" NHWC_IN=$i timeout 300 python scripts/experiments/block_check.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn"
done; done
