#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-200
timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; tail -2 gpurun_out/bench_train.err; python - <<PY
import json
j=json.loads(open("gpurun_out/bench_train.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","final_loss")})
print(j["roofline"])
for k,v in list(j["kernel_breakdown"].items()): print(f"  {k:24s} {v}")
PY
