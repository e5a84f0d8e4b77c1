#!/bin/bash
# Round-2 visit A: smoke, the whole GPU suite, then the evidence collection of scripts/gpu_profile_r2.sh.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -3
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-220 | tee gpurun_out/r2/pytest_gpu_tail.txt
bash scripts/gpu_profile_r2.sh 2>&1 | tee gpurun_out/r2/visit.log | cut -c1-220
