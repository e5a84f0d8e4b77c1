#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for i in 1 2 3 4 5 6; do timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -1 | cut -c1-300; done
timeout 900 python -m pytest tests/test_baseline_shapes_gpu.py tests/test_rank_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "topk or rank or kats or golden or oracle or ties or fewer or large" 2>&1 | tail -8 | cut -c1-250
run() { name=$1; shift; env "$@" timeout 300 python bench.py --workload topk5m --no-cpu-baseline --steps 8 > gpurun_out/t5m_$name.json 2> gpurun_out/t5m_$name.err
  python - "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/t5m_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(f"{sys.argv[1]:22s} users/launch {r['users_per_launch']:4d} tile {r['users_per_register_tile']:3d}  {r['avg_launch_ms']:.4f} ms  {r['hbm_GBps']:.0f} GB/s frac {r['frac']}  wall/step {j['ms_per_step']}")
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
run u16_auto X=1
run u16_noseed RT_TOPK_SEED=0
run u16_ns6 RT_TOPK_STAGES=6
rm -rf gpurun_out/prof_t16
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_t16 -o p -- python $R/bench.py --workload topk5m --no-cpu-baseline --steps 6 > $R/gpurun_out/prof_t16.log 2>&1)
python scripts/prof_summary.py $(find gpurun_out/prof_t16 -name "*.db" | head -1) > gpurun_out/prof_t16.md
head -9 gpurun_out/prof_t16.md | cut -c1-200
find gpurun_out -name "*.db" -size +20M -delete
