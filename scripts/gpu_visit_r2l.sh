#!/bin/bash
# recommend() glue on the per-Dataset session index: the glue / contract tests, then the auto line (short legs) for recommend_e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2l; mkdir -p $O; export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_models_gpu.py tests/test_checkpoint.py -q -m gpu -x -k "recommend or contract or checkpoint" -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest.txt
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rec-steps 3 --topk-steps 1 > $O/bench_auto_short.json 2> $O/bench_auto_short.err
python - <<'P'
import json
j = json.loads(open("gpurun_out/r2l/bench_auto_short.json").read().strip().splitlines()[-1])
print("train", j["value"], "e2e", {k: v for k, v in j["recommend_e2e"].items() if k != "what"})
P
