#!/bin/bash
# One GPU visit: smoke, the whole GPU test suite, the default (three-leg) bench line.  Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -3
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x 2>&1 | tail -25 | cut -c1-250
timeout 900 python bench.py "$@" > gpurun_out/bench_auto.json 2> gpurun_out/bench_auto.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_auto.err | cut -c1-300
python - <<'PY'
import json
try:
    j=json.loads(open("gpurun_out/bench_auto.json").read().strip().splitlines()[-1])
    print({k:j[k] for k in ("value","ms_per_step","final_loss","steps")}); print(j["roofline"]); print("cpu", j["cpu_baseline"])
    for k,v in list(j["kernel_breakdown"].items()): print(f"  {k:28s} {v}")
    for leg in ("recommend_e2e","recommend","topk5m"):
        r=j.get(leg); 
        if r: print(leg, {k:r[k] for k in r if k not in ("config","metric")})
    print("env", j.get("env"))
except Exception as e: print("bench parse failed", e)
PY
