#!/bin/bash
# Round-2 visit H: kernel traces of the three other model families and the 2-rank code path, re-taken with the bf16x6 GEMM default.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r2g; mkdir -p $O; export TMPDIR=/tmp
prof() { name=$1; shift
  rm -rf $O/prof_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof_$name -o p -- python $R/bench.py --no-cpu-baseline "$@" > $O/prof_$name.log 2>&1)
  python scripts/prof_summary.py $(find $O/prof_$name -name "*.db" | head -1) 22 > $O/rocprof_kernel_trace_$name.md
  head -6 $O/rocprof_kernel_trace_$name.md | cut -c1-170
}
prof bert4rec --workload bert4rec --steps 10 --warmup 3
prof hstu --workload hstu --steps 10 --warmup 3
prof esasrec --workload esasrec --steps 10 --warmup 3
RT_BENCH_BACKEND=gloo timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 10 --warmup 3 2> $O/bench_auto_2ranks_on_1gpu_gloo.err | grep '^{' > $O/bench_auto_2ranks_on_1gpu_gloo.json
echo "2rank rc=$?"; head -c 300 $O/bench_auto_2ranks_on_1gpu_gloo.json; echo
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; find $O -name "*agent_info.csv" -delete
