#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -k "linear_fwd_bwd and 25600-256" -x 2>&1 | grep -E "^E|Error|assert|passed|failed" | head -20 | cut -c1-300
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -k "gemm_224" 2>&1 | grep -E "FAILED|passed|failed|Error|^E  " | head -20 | cut -c1-300
show() { python - $1 <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], {k:j[k] for k in ("value","ms_per_step","final_loss","steps")}); r=j["roofline"]; print("  roof", r.get("achieved"), r.get("frac"), r.get("avg_launch_ms"), r.get("single_stream"))
    for k,v in list(j["kernel_breakdown"].items())[:3]: print(f"  {k:28s} {v}")
except Exception as e: print("parse failed", e)
PY
}
timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train.json 2> gpurun_out/b_train.err; show b_train
RT_GEMM_BM224=0 timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train_no224.json 2> gpurun_out/b_train_no224.err; show b_train_no224
