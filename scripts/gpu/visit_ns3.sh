#!/bin/bash
mkdir -p gpurun_out/r6_ns
timeout 900 python -m pytest tests/test_native_step_gpu.py tests/test_baseline_shapes_gpu.py -q -k "native or compiled or C2_shape" -s > gpurun_out/r6_ns/7_test.txt 2>&1
grep -v "Warn\|warn" gpurun_out/r6_ns/7_test.txt | tail -30 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_ns/8_smoke.txt 2>&1; tail -5 gpurun_out/r6_ns/8_smoke.txt | cut -c1-250
