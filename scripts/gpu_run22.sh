#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-220
RT_ATTN_IMPL=stream timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "mha or hstu" 2>&1 | tail -2 | cut -c1-220
TAG=new timeout 120 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; python - <<PY
import json
j=json.loads(open("gpurun_out/bench_train.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","final_loss")})
PY
