#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/gpu_run11.sh 2>&1 | grep -v "Warning\|warnings\|detach\|^$\|Docs:\|assert abs\|loss=float"
bash scripts/gpu_prof_train.sh | head -34
