#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
timeout 900 python -m pytest tests/test_acc_session_gpu.py -q -m gpu -k "train_mode" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/rc.log
tail -40 $O/tests.log
