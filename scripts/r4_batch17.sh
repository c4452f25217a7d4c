#!/bin/bash
# Round-4 GPU batch 17: the new unit tests of the train-mode launches.
O=gpurun_out/r4r; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_conv_gpu.py -q -m gpu -k "float64_formulas or partial_sums_in_its_epilogue" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
tail -40 $O/tests.log
