#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -k "linear or mha" 2>&1 | tail -3 | cut -c1-200
TAG=nodma timeout 120 python scripts/attn_bench.py 2>&1 | tail -2
TAG=dma RT_ATTN_DMA=1 timeout 120 python scripts/attn_bench.py 2>&1 | tail -2
show() { python - $1 <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], {k:j[k] for k in ("value","ms_per_step","final_loss","steps")})
    for k,v in list(j["kernel_breakdown"].items())[:6]: print(f"  {k:28s} {v}")
except Exception as e: print("parse failed", e)
PY
}
timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train.json 2> gpurun_out/b_train.err; show b_train
RT_ATTN_DMA=1 timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train_dma.json 2> gpurun_out/b_train_dma.err; show b_train_dma
