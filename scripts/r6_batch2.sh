#!/bin/bash
# Round-6 batch 2 (one box): the tests added since batch 1 + whatever failed there, then the full suite.
OUT=${1:-gpurun_out/r6c}; mkdir -p $OUT
python -m pytest tests/test_session_gpu.py tests/test_conv_gpu.py tests/test_distributed_gpu.py -m gpu -q --durations=10 -p no:cacheprovider -k "frozen or mse or zero_padded or ladder or torchrun or refused" > $OUT/new_tests.log 2>&1
tail -4 $OUT/new_tests.log
HF_TOL_LOG=$OUT/tol.jsonl python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $OUT/suite.log 2>&1
tail -4 $OUT/suite.log
