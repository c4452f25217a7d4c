cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python scripts/experiments/engine_check.py 2>&1 | grep -v amdgpu.ids | tail -16
