#!/bin/bash
# rocprofv3 kernel trace of the training bench -> per-kernel table (all kernels, 45 rows)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_train
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_train -o p -- python $R/bench.py --no-cpu-baseline --workload train --steps 10 --warmup 3 > $R/gpurun_out/prof_train.log 2>&1)
python scripts/prof_summary.py $(find gpurun_out/prof_train -name "*.db" | head -1) 45 > gpurun_out/prof_train.md
cat gpurun_out/prof_train.md | cut -c1-175
find gpurun_out -name "*.db" -size +20M -delete
