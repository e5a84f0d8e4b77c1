#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for coop in 0 1; do for dbg in 0 1 2 3 4 5 6 7; do
TAG="coop=$coop dbg=$dbg" RT_ATTN_COOP=$coop RT_ATTN_DBG=$dbg timeout 100 python scripts/attn_fwd_bench.py 2>&1 | grep -v amdgpu.ids
done; done
