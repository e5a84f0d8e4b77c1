#!/bin/bash
# Round-2 late visit: (1) the N>1 code path of bench.py with 2 ranks on this one-GPU box (gloo; LOCAL_RANK wraps), (2) the opt-in
# bf16x6 GEMM inner loop: accuracy tests, microbenchmark and the train line with and without it.
cd /root/repo; O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
RT_BENCH_BACKEND=gloo timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_auto_2ranks_on_1gpu_gloo.json 2> $O/bench_auto_2ranks_on_1gpu_gloo.err
echo "2rank rc=$?"; tail -c 600 $O/bench_auto_2ranks_on_1gpu_gloo.json | cut -c1-600; echo
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bf16x6 or gemm" -x 2>&1 | tail -15 | tee $O/pytest_bf16x6.txt
timeout 200 python scripts/gemm_bench.py 2>&1 | tail -8 | tee $O/gemm_bench_exact.txt
RT_GEMM_SPLIT=bf16x6 timeout 200 python scripts/gemm_bench.py 2>&1 | tail -8 | tee $O/gemm_bench_bf16x6.txt
timeout 200 python bench.py --workload train --steps 100 --no-cpu-baseline > $O/bench_train_exact.json 2> $O/bench_train_exact.err
RT_GEMM_SPLIT=bf16x6 timeout 200 python bench.py --workload train --steps 100 --no-cpu-baseline > $O/bench_train_bf16x6.json 2> $O/bench_train_bf16x6.err
python - <<'P'
import json
for t in ("exact", "bf16x6"):
    try:
        j = json.loads(open(f"gpurun_out/r2e/bench_train_{t}.json").read().strip().splitlines()[-1])
        print(t, j["value"], "seqs/s", j["ms_per_step"], "ms/step loss", j["final_loss"], {k: v for k, v in list(j["kernel_breakdown"].items())[:6]} if isinstance(j.get("kernel_breakdown"), dict) else "")
    except Exception as e:
        print(t, "failed:", e)
P
