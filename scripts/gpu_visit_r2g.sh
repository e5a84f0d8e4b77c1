#!/bin/bash
# Round-2 visit G (after the GEMM inner loop moved to the bf16x6 split): smoke, the whole GPU suite, the default bench line, the
# train kernel traces + HBM-traffic PMC passes and the family lines (R2_LEAN part of gpu_profile_r2.sh), and the same train
# line with RT_GEMM_SPLIT=exact beside it.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export R2_OUT=r2g R2_LEAN=1 TMPDIR=/tmp; mkdir -p gpurun_out/r2g
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -3
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-220 | tee gpurun_out/r2g/pytest_gpu_tail.txt
RT_GEMM_SPLIT=exact timeout 200 python bench.py --workload train --steps 100 --no-cpu-baseline > gpurun_out/r2g/bench_train_exact_gemm.json 2> gpurun_out/r2g/bench_train_exact_gemm.err
bash scripts/gpu_profile_r2.sh 2>&1 | tee gpurun_out/r2g/visit.log | cut -c1-220
