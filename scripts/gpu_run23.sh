#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
for ld in 0 2; do
RT_TOPK_LOADERS=$ld timeout 300 python bench.py --workload topk5m --no-cpu-baseline > gpurun_out/b.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print("loaders=$ld", j["value"], j["unit"], "ms/step", j["ms_per_step"], j["roofline"]["achieved"], j["roofline"]["unit"], j["roofline"].get("avg_launch_ms"))
PY
done
RT_TOPK_LOADERS=2 timeout 300 python bench.py --workload recommend --no-cpu-baseline > gpurun_out/b.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print("recommend loaders=2", j["value"], j["unit"], "ms/step", j["ms_per_step"])
PY
RT_TOPK_LOADERS=0 timeout 300 python bench.py --workload recommend --no-cpu-baseline > gpurun_out/b.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print("recommend loaders=0", j["value"], j["unit"], "ms/step", j["ms_per_step"])
PY
