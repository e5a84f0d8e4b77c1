#!/bin/bash
# First hardware visit of the packed-sequence work.  Run from a tree with scripts/wip/packed_sequences.patch applied
# and the library rebuilt (`git apply scripts/wip/packed_sequences.patch && python -m rectools_amd.build`):
#   gpurun --timeout 600 -- 'bash scripts/wip/gpu_visit_packed.sh'
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/packed; mkdir -p $O; export TMPDIR=/tmp
# 1. kernels and paths against their references (stop at the first failure: the order goes from kernels to models)
timeout 400 python -m pytest tests/test_packed_gpu.py -q --maxfail=30 --tb=short -p no:cacheprovider 2>&1 | tail -60 | tee $O/pytest_packed.txt
# 2. recommend() end to end with and without the packed encoder (bench: short legs)
for packed in 1 0; do
  RT_PACKED=$packed timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rec-steps 3 --topk-steps 1 > $O/bench_auto_packed$packed.json 2> $O/bench_auto_packed$packed.err
done
# 3. the training step with packed batches (first form: modular autograd ops) against the default
for packed in 0 1; do
  RT_PACKED_TRAIN=$packed timeout 200 python bench.py --workload train --steps 100 --no-cpu-baseline > $O/bench_train_packed$packed.json 2> $O/bench_train_packed$packed.err
done
python - <<'P'
import json
def last(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e: return {"error": str(e)}
for packed in (1, 0):
    j = last(f"gpurun_out/packed/bench_auto_packed{packed}.json"); print("RT_PACKED", packed, "recommend_e2e", (j.get("recommend_e2e") or {}).get("value"), j.get("error"))
for packed in (0, 1):
    j = last(f"gpurun_out/packed/bench_train_packed{packed}.json"); print("RT_PACKED_TRAIN", packed, j.get("value"), "seqs/s", j.get("ms_per_step"), "loss", j.get("final_loss"), j.get("error"))
P
