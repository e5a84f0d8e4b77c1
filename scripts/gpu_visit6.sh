#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -15 | cut -c1-250
show() { python - $1 <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], {k:j[k] for k in ("value","ms_per_step","final_loss","steps")}); r=j["roofline"]; print("  roof", r.get("achieved"), r.get("frac"), r.get("avg_launch_ms"), r.get("single_stream"), "kernel_ms", r.get("kernel_ms_per_step"))
    for k,v in list(j["kernel_breakdown"].items())[:12]: print(f"  {k:28s} {v}")
except Exception as e: print("parse failed", e)
PY
}
timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train.json 2> gpurun_out/b_train.err; show b_train
RT_SIDE_STREAM=0 timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train_1s.json 2> gpurun_out/b_train_1s.err; show b_train_1s
timeout 600 python bench.py --workload bert4rec --steps 20 --warmup 4 > gpurun_out/fam_bert4rec.json 2> gpurun_out/fam_bert4rec.err; show fam_bert4rec
