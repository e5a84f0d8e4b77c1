#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for impl in 0 2 3 4; do RT_GEMM_IMPL=$impl timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tail -7; done
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 | cut -c1-200
for impl in 2 3; do
RT_GEMM_IMPL=$impl timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_train_$impl.json 2> gpurun_out/bench_train.err; tail -2 gpurun_out/bench_train.err; python - <<PY
import json
j=json.loads(open("gpurun_out/bench_train_$impl.json").read().strip().splitlines()[-1])
print($impl, {k:j[k] for k in ("value","ms_per_step","final_loss")})
print(j["roofline"])
for k,v in list(j["kernel_breakdown"].items())[:8]: print(f"  {k:24s} {v}")
PY
done
