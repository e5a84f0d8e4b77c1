#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_transformer_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_tr.log; tail -60 gpurun_out/pytest_tr.log | cut -c1-200
