#!/bin/bash
# Round-2 visit C: attention after the cheaper dropout hash + hoisted row-fragment loads; BERT4Rec with the split-K dS product.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r2c; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_baseline_shapes_gpu.py tests/test_models_gpu.py tests/test_checkpoint.py tests/test_trajectory_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200
for impl in ring stream; do
  RT_ATTN_IMPL=$impl timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_baseline_shapes_gpu.py tests/test_transformer_gpu.py -m gpu -q -p no:cacheprovider -k "mha or hstu or dropout or stu" 2>&1 | tail -1 | cut -c1-200
done
for impl in res ring; do RT_ATTN_IMPL=$impl timeout 120 python scripts/attn_bench.py 2>&1 | grep "^\[" ; done | tee $O/attn_bench_c2.txt
for impl in ring stream; do RT_ATTN_IMPL=$impl timeout 120 python scripts/attn_bench.py --L 512 --hstu 2>&1 | grep "^\[" ; done | tee $O/attn_bench_c4_hstu.txt
for w in train bert4rec hstu; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
j=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1]); print("$w", j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in list(j["kernel_breakdown"].items())[:6]})
PY
done
