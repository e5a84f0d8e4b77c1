#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
run() { tag="$1"; shift
env "$@" timeout 300 python bench.py --workload topk5m --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print("$tag", "ms/step", j["ms_per_step"], j["roofline"]["achieved"], j["roofline"]["unit"], "kernel ms", j["roofline"].get("avg_launch_ms"))
PY
}
run "ld2 wg1 ns6" RT_TOPK_LOADERS=2
run "ld2 wg2 ns4" RT_TOPK_LOADERS=2 RT_TOPK_WG_PER_CU=2 RT_TOPK_STAGES=4
run "ld2 wg2 ns3" RT_TOPK_LOADERS=2 RT_TOPK_WG_PER_CU=2 RT_TOPK_STAGES=3
run "ld0 wg2 ns4" RT_TOPK_LOADERS=0 RT_TOPK_WG_PER_CU=2 RT_TOPK_STAGES=4
run "ld2 wg3 ns3" RT_TOPK_LOADERS=2 RT_TOPK_WG_PER_CU=3 RT_TOPK_STAGES=3
run "ld0 wg3 ns3" RT_TOPK_LOADERS=0 RT_TOPK_WG_PER_CU=3 RT_TOPK_STAGES=3
