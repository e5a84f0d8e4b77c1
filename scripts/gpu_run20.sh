#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "mha or hstu" 2>&1 | tail -2 | cut -c1-220
TAG=coop timeout 120 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids
TAG=nocoop RT_ATTN_COOP=0 timeout 120 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids
