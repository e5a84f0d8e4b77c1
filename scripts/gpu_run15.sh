#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-default} timeout 120 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids
TAG=${1:-default}-dkv4 RT_ATTN_DKV_NW=4 timeout 120 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids
