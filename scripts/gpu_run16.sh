#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "mha or hstu" 2>&1 | tail -3 | cut -c1-220
RT_ATTN_IMPL=stream timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "mha or hstu" 2>&1 | tail -2 | cut -c1-220
bash scripts/gpu_run15.sh new
