#!/bin/bash
# Full validation + measurement visit: GPU test suite, smoke, the three bench workloads, rocprofv3 kernel traces
# and HBM-traffic PMC passes.  Summaries land in gpurun_out/ (copied to profiles/ afterwards).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -3
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | cut -c1-200
# opt-in tests (the two-stage top-k is off by default)
RT_TEST_TWO_STAGE=1 timeout 300 python -m pytest tests/test_rank_two_stage_gpu.py -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-200
timeout 600 python bench.py > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; tail -c 1500 gpurun_out/bench_train.json | head -c 1500; echo
timeout 300 python bench.py --workload recommend --no-cpu-baseline > gpurun_out/bench_recommend.json 2>/dev/null; cut -c1-700 gpurun_out/bench_recommend.json
timeout 300 python bench.py --workload topk5m > gpurun_out/bench_topk5m.json 2>/dev/null; cut -c1-900 gpurun_out/bench_topk5m.json
timeout 300 python bench.py --workload topk5m --users-per-step 1024 --users-per-pass 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_topk5m_u1024.json 2>/dev/null; cut -c1-500 gpurun_out/bench_topk5m_u1024.json
# the N>1 code path of bench.py (barrier, max over ranks, gradient all-reduce, aggregate value) with 2 ranks on this box's one GPU
for w in train recommend; do
  RT_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --workload $w --steps 5 --warmup 2 > gpurun_out/bench_${w}_2ranks_1gpu.json 2> gpurun_out/bench_${w}_2ranks_1gpu.err
  tail -c 600 gpurun_out/bench_${w}_2ranks_1gpu.json | cut -c1-600; echo
done
timeout 200 python scripts/itemnet_bench.py > gpurun_out/itemnet_bench.txt 2>/dev/null; cat gpurun_out/itemnet_bench.txt | cut -c1-200
prof() { name=$1; shift
  rm -rf gpurun_out/prof_$name
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$name -o p -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_$name.log 2>&1)
  python scripts/prof_summary.py $(find gpurun_out/prof_$name -name "*.db" | head -1) > gpurun_out/prof_$name.md
  head -12 gpurun_out/prof_$name.md | cut -c1-160
}
prof train --workload train --steps 10 --warmup 3
RT_SIDE_STREAM=0 prof train_single_stream --workload train --steps 10 --warmup 3   # kernel durations without wgrad overlap: what bench.py's roofline pass times
prof topk5m --workload topk5m --steps 4 --warmup 1
prof recommend --workload recommend --steps 5 --warmup 1
pmc() { name=$1; ctr=$2; shift; shift
  rm -rf gpurun_out/pmc_$name
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$name -o p -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$name.log 2>&1)
  f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$ctr" <<'PY' | tee gpurun_out/pmc_$name.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name")==sys.argv[2]:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:6]:
    print(f"{sys.argv[2]} {k:60s} calls={len(v)} avg={sum(v)/len(v):.1f} max={max(v):.1f}")
PY
}
pmc topk5m_fetch FETCH_SIZE --workload topk5m --steps 2 --warmup 1
pmc topk5m_write WRITE_SIZE --workload topk5m --steps 2 --warmup 1
bash scripts/gpu_pmc_train.sh > /dev/null 2>&1; head -4 gpurun_out/pmc_train_FETCH_SIZE.txt; head -3 gpurun_out/pmc_train_WRITE_SIZE.txt
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +5M -delete
