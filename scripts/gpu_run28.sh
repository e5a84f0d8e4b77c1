#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_rank_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/rank.log 2>&1; tail -3 gpurun_out/rank.log | cut -c1-200
run() { tag="$1"; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
env "${envs[@]}" timeout 300 python bench.py --workload recommend --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/b.json 2>gpurun_out/b.err; python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("$tag", round(j["value"]), j["unit"], "ms/step", j["ms_per_step"], j["roofline"]["mfma_f32_TFLOPs"], "TF")
except Exception as e:
    print("$tag failed", e, open("gpurun_out/b.err").read()[-300:])
PY
}
run "hash filter  " A=1
run "hash, upp128 " A=1 --users-per-pass 128
