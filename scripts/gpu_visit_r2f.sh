#!/bin/bash
# bf16x6 GEMM iteration visit: accuracy tests, microbenchmark and the train line with and without RT_GEMM_SPLIT=bf16x6.
cd /root/repo; O=gpurun_out/${1:-r2f}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bf16x6 or gemm or linear" -x 2>&1 | tail -5 | tee $O/pytest_bf16x6.txt
RT_GEMM_SPLIT=exact timeout 200 python scripts/gemm_bench.py 2>&1 | tail -6 | tee $O/gemm_bench_exact.txt
timeout 200 python scripts/gemm_bench.py 2>&1 | tail -6 | tee $O/gemm_bench_bf16x6.txt
RT_GEMM_SPLIT=exact timeout 200 python bench.py --workload train --steps 100 --no-cpu-baseline > $O/bench_train_exact.json 2> $O/bench_train_exact.err
timeout 200 python bench.py --workload train --steps 100 --no-cpu-baseline > $O/bench_train_bf16x6.json 2> $O/bench_train_bf16x6.err
python - $O <<'P'
import json, sys
for t in ("exact", "bf16x6"):
    try:
        j = json.loads(open(f"{sys.argv[1]}/bench_train_{t}.json").read().strip().splitlines()[-1])
        kb = j["kernel_breakdown"]
        print(t, j["value"], "seqs/s", j["ms_per_step"], "ms/step loss", j["final_loss"], {k: kb[k] for k in ("rt_gemm", "rt_gemm_grouped") if k in kb})
    except Exception as e:
        print(t, "failed:", e)
P
