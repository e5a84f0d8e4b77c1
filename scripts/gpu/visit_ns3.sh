#!/bin/bash
mkdir -p gpurun_out/r6_ns
timeout 900 python -m pytest tests/test_native_step_gpu.py -q > gpurun_out/r6_ns/5_test.txt 2>&1
tail -30 gpurun_out/r6_ns/5_test.txt | cut -c1-250
