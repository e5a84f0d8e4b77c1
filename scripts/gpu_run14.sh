#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -12 | cut -c1-220
echo "--- streaming family forced"
RT_ATTN_IMPL=stream timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "mha or hstu" 2>&1 | tail -4 | cut -c1-220
for nw in 8 4; do
RT_ATTN_DKV_NW=$nw timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; python - <<PY
import json
j=json.loads(open("gpurun_out/bench_train.json").read().strip().splitlines()[-1])
print("dkv NW=$nw", {k:j[k] for k in ("value","ms_per_step","final_loss")})
for k,v in list(j["kernel_breakdown"].items())[:6]: print(f"  {k:24s} {v}")
PY
done
bash scripts/gpu_prof_train.sh | head -16
