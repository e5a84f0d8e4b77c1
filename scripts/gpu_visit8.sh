#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_baseline_shapes_gpu.py tests/test_trajectory_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 -x 2>&1 | tail -12 | cut -c1-250
TAG=dma timeout 120 python scripts/attn_bench.py 2>&1 | tail -2
TAG=nodma RT_ATTN_DMA=0 timeout 120 python scripts/attn_bench.py 2>&1 | tail -2
timeout 600 python bench.py --workload train --no-cpu-baseline > gpurun_out/b_train.json 2> gpurun_out/b_train.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/b_train.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","final_loss")})
for k,v in list(j["kernel_breakdown"].items())[:8]: print(f"  {k:28s} {v}")
PY
bash scripts/gpu_pmc_attn.sh dma 2>&1 | tail -8 | cut -c1-250
