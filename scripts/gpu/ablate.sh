#!/bin/bash
# Kernel time of the 4,096-user coarse pass with parts of its selection switched off (diagnostic build -DRT_ABLATION_BUILD of rt_topk.hip,
# loaded through RT_LIB_PATH): bash scripts/gpu/ablate.sh <out dir> <RT_TOPK_DEBUG value> [extra env ...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$1; DBG=$2; shift 2
mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && env "$@" RT_TOPK_DEBUG=$DBG rocprofv3 --kernel-trace -d $R/$O/prof -o p -- python $R/bench.py --workload topk5m --users-per-step 4096 --topk-steps 2 --no-cpu-baseline > $R/$O/bench.json 2> $R/$O/bench.err)
python scripts/prof_summary.py $(find $O/prof -name "*.db" | head -1) 8 2>&1 | head -14 | cut -c1-170
find $O -name "*.db" -delete
