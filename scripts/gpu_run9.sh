#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_models_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -60 | cut -c1-220
