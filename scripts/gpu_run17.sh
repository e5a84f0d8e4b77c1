#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | cut -c1-200
echo "--- two-kernel sampled path forced"
RT_SAMPLED_SPLIT=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py -m gpu -q -p no:cacheprovider -k "sampled or golden or random" 2>&1 | tail -3 | cut -c1-200
for v in fused split; do
if [ $v = split ]; then export RT_SAMPLED_SPLIT=1; fi
timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; python - <<PY
import json
j=json.loads(open("gpurun_out/bench_train.json").read().strip().splitlines()[-1])
print("$v", {k:j[k] for k in ("value","ms_per_step","final_loss")})
for k,v in list(j["kernel_breakdown"].items())[:5]: print(f"  {k:24s} {v}")
PY
done
