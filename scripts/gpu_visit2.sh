#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -2; done
timeout 600 python scripts/smoke_stress.py 40 2>&1 | grep -v "OK$" | tail -30
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -40 | cut -c1-250
