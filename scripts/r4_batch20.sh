#!/bin/bash
# Round-4 GPU batch 20: the suite's heaviest CPU references with 16 torch threads against torch's default.
O=gpurun_out/r4w; mkdir -p $O
SEL="newton_solve_matches_reference_cpu_path or train_mode_batchnorm_product_and_solve or bottleneck_net or session_steps_match_reference"
( time HF_TEST_CPU_THREADS=16 python -m pytest tests/test_optimizer_gpu.py tests/test_session_gpu.py -q -m gpu -k "$SEL" --durations=6 ) > $O/threads16.log 2>&1
( time HF_TEST_CPU_THREADS=0 python -m pytest tests/test_optimizer_gpu.py tests/test_session_gpu.py -q -m gpu -k "$SEL" --durations=6 ) > $O/threads_default.log 2>&1
tail -14 $O/threads16.log; tail -14 $O/threads_default.log; nproc; python -c "import torch; print(torch.get_num_threads())"
