cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s9; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_optimizer_gpu.py -m gpu -q -k "config4 or resnet18_newton or deterministic_mode or distinct" > $O/pytest.log 2>&1
grep -v "^  File\|amdgpu.ids" $O/pytest.log | tail -60
