#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python -m pytest tests/test_torch_library_ops.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
