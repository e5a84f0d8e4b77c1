#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
run() { tag="$1"; shift; envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
env "${envs[@]}" timeout 300 python bench.py --workload recommend --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/b.json 2>gpurun_out/b.err; python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("$tag", round(j["value"]), j["unit"], "ms/step", j["ms_per_step"], j["roofline"]["mfma_f32_TFLOPs"], "TF", j["config"]["workload"][:60])
except Exception as e:
    print("$tag failed", e, open("gpurun_out/b.err").read()[-300:])
PY
}
run "auto            " A=1
run "upp128          " A=1 --users-per-pass 128
run "upp128 nolds    " RT_TOPK_LDS_LISTS=0 --users-per-pass 128
run "upp64 loaders0  " RT_TOPK_LOADERS=0
run "upp64 wg2       " RT_TOPK_WG_PER_CU=2 RT_TOPK_STAGES=3
run "upp32           " A=1 --users-per-pass 32
run "users 65536     " A=1 --users-per-step 65536
