#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for impl in 2 3; do RT_GEMM_IMPL=$impl timeout 300 python scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; done
