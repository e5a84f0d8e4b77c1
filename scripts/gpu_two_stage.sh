#!/bin/bash
# Validation visit for the opt-in two-stage top-k: its own tests, then the 5M x 512 bench in both modes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
RT_TEST_TWO_STAGE=1 timeout 300 python -m pytest tests/test_rank_two_stage_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -25 | cut -c1-220
timeout 200 python scripts/two_stage_bench.py 2>&1 | grep -v Warn | tail -4 | cut -c1-250
