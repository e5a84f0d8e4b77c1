#!/bin/bash
# wgrad split-K cap under the bf16x6 loop: train line with RT_WGRAD_SPLITS = 32 / 64 / 128 (+ 48, 96)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2k; mkdir -p $O; export TMPDIR=/tmp
for sp in 64 32 128 48 96; do
  RT_WGRAD_SPLITS=$sp timeout 120 python bench.py --workload train --steps 100 --no-cpu-baseline > $O/bench_train_sp$sp.json 2> $O/bench_train_sp$sp.err
  python - $O/bench_train_sp$sp.json $sp <<'P'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); kb = j["kernel_breakdown"]
print("splits cap", sys.argv[2], j["value"], "seqs/s", j["ms_per_step"], "ms/step", {k: kb[k] for k in ("rt_gemm", "rt_gemm_grouped") if k in kb})
P
done
