#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_baseline_shapes_gpu.py tests/test_trajectory_gpu.py tests/test_models_gpu.py tests/test_validation_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -6 | cut -c1-250
show() { python - $1 <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], {k:j[k] for k in ("value","ms_per_step","final_loss","steps")})
    for k,v in list(j["kernel_breakdown"].items())[:6]: print(f"  {k:28s} {v}")
except Exception as e: print("parse failed", e)
PY
}
timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train.json 2> gpurun_out/b_train.err; show b_train
RT_LOSS_FAST=0 timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train_slowloss.json 2> gpurun_out/b_train_slowloss.err; show b_train_slowloss
timeout 600 python bench.py --workload hstu --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/f_hstu.json 2> gpurun_out/f_hstu.err; show f_hstu
