cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s21
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/s21/pytest.txt
tail -40 gpurun_out/s21/pytest.txt
