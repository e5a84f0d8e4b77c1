#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_rank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
prof() { name=$1; shift
  rm -rf gpurun_out/prof_$name
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$name -o p -- python $R/bench.py --no-cpu-baseline $BARGS > $R/gpurun_out/prof_$name.log 2>&1)
  db=$(find gpurun_out/prof_$name -name "*.db" | head -1)
  echo "== $name"; grep '"metric"' gpurun_out/prof_$name.log | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); r=j['roofline']; print('users/s=%.1f ms/launch=%.3f hbm=%.1f mfma=%.2f'%(j['value'],r['avg_launch_ms'],r['hbm_GBps'],r['mfma_f32_TFLOPs']))"
  python scripts/prof_summary.py $db | grep -v "at::native" | head -8 | tee gpurun_out/prof_$name.md
}
BARGS="--workload topk5m --users-per-step 32 --steps 4 --warmup 1"
prof u32 X=1
prof u32_noseed RT_TOPK_SEED=0
BARGS="--workload topk5m --users-per-step 256 --users-per-pass 128 --steps 3 --warmup 1"
prof u256_t128 X=1
BARGS="--workload recommend --steps 5 --warmup 1"
prof rec_t64 X=1
find gpurun_out -name "*.db" -size +30M -delete
