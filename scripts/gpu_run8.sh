#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2 3; do timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "embed" 2>&1 | tail -3 | cut -c1-150; done
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "layernorm or embed" 2>&1 | tail -3 | cut -c1-150
