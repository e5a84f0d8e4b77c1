#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python -m pytest tests/test_rank_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline $BARGS > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/b_{n}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(f"{n:28s} users/s={j['value']:10.1f} ms/launch={r['avg_launch_ms']:8.3f} hbm={r['hbm_GBps']:7.1f}GB/s mfma={r['mfma_f32_TFLOPs']:6.2f}TF")
except Exception as e:
    print(n,"FAILED",e); print(open(f"gpurun_out/b_{n}.err").read()[-600:])
PY
}
BARGS="--workload topk5m --users-per-step 32 --steps 4 --warmup 1"
run u32_auto X=1
run u32_noseed RT_TOPK_SEED=0
run u32_nosel RT_TOPK_DEBUG=1
run u32_globallists RT_TOPK_LDS_LISTS=0
run u32_s4 RT_TOPK_STAGES=4
run u32_s3_wg2 RT_TOPK_STAGES=3 RT_TOPK_WG_PER_CU=2
run u32_norot RT_TOPK_ROTATE=0
BARGS="--workload topk5m --users-per-step 64 --steps 4 --warmup 1"
run u64_auto X=1
run u64_nosel RT_TOPK_DEBUG=1
BARGS="--workload topk5m --users-per-step 256 --users-per-pass 64 --steps 3 --warmup 1"
run u256_t64 X=1
BARGS="--workload topk5m --users-per-step 256 --users-per-pass 128 --steps 3 --warmup 1"
run u256_t128 X=1
run u256_t128_nosel RT_TOPK_DEBUG=1
BARGS="--workload topk5m --users-per-step 1024 --users-per-pass 128 --steps 2 --warmup 1"
run u1024_t128 X=1
BARGS="--workload recommend --steps 5 --warmup 1"
run rec_t64 X=1
run rec_t64_nosel RT_TOPK_DEBUG=1
run rec_t64_wg2 RT_TOPK_STAGES=3 RT_TOPK_WG_PER_CU=2
BARGS="--workload recommend --users-per-pass 128 --steps 5 --warmup 1"
run rec_t128 X=1
BARGS="--workload recommend --users-per-pass 32 --steps 5 --warmup 1"
run rec_t32 X=1
