#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
run() { tag="$1"; shift
env "$@" timeout 300 python bench.py --workload topk5m --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print("$tag", "ms/step", j["ms_per_step"], "kernel ms", j["roofline"].get("avg_launch_ms"))
PY
}
run "full        " RT_TOPK_LOADERS=2
run "no select   " RT_TOPK_LOADERS=2 RT_TOPK_DEBUG=1
run "no mfma     " RT_TOPK_LOADERS=2 RT_TOPK_DEBUG=3
run "no dma      " RT_TOPK_LOADERS=2 RT_TOPK_DEBUG=5
run "no dma wg2  " RT_TOPK_LOADERS=2 RT_TOPK_DEBUG=5 RT_TOPK_WG_PER_CU=2 RT_TOPK_STAGES=4
run "barriers only" RT_TOPK_LOADERS=2 RT_TOPK_DEBUG=7
