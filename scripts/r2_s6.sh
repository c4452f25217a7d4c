cd "$GRAFT_REPO_ROOT" || exit 1
python scripts/experiments/determinism_probe.py 2>&1 | grep -v amdgpu.ids | tail -40
