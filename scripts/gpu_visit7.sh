#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_baseline_shapes_gpu.py tests/test_checkpoint.py -m gpu -q -p no:cacheprovider --timeout=600 -k "hstu or stu or ckpt or checkpoint" 2>&1 | tail -6 | cut -c1-250
timeout 600 python bench.py --workload hstu --steps 20 --warmup 4 > gpurun_out/fam_hstu.json 2> gpurun_out/fam_hstu.err
python - <<'PY'
import json
try:
    j=json.loads(open("gpurun_out/fam_hstu.json").read().strip().splitlines()[-1])
    print({k:j[k] for k in ("value","ms_per_step","final_loss","steps")})
    for k,v in list(j["kernel_breakdown"].items())[:10]: print(f"  {k:28s} {v}")
except Exception as e: print("parse failed", e)
PY
