#!/bin/bash
# First GPU visit: smoke, ranker parity tests, top-k benches at 3 users-per-pass settings, rocprof stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt; lscpu | grep "Model name" >> gpurun_out/gpu_info.txt
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests/test_rank_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_rank.log 2>&1
tail -15 gpurun_out/pytest_rank.log
for upp in 32 64 128; do
  timeout 300 python bench.py --workload topk5m --users-per-pass $upp --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_topk5m_$upp.json 2> gpurun_out/bench_topk5m_$upp.err
  cat gpurun_out/bench_topk5m_$upp.json
done
timeout 400 python bench.py > gpurun_out/bench_recommend.json 2> gpurun_out/bench_recommend.err
cat gpurun_out/bench_recommend.json
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_topk5m" -o topk5m -- python "$OLDPWD/bench.py" --workload topk5m --users-per-pass 64 --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_topk5m.log" 2>&1
cd "$OLDPWD"
find gpurun_out/prof_topk5m -name "*stats*" | head; 
f=$(find gpurun_out/prof_topk5m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
# keep merge-back small
find gpurun_out/prof_topk5m -name "*.db" -size +20M -delete
