#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -2
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -6 | cut -c1-250
for f in 1 0; do RT_TOPK_BLOOM=$f python bench.py --workload recommend --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"bloom=$f\", j[\"value\"], j[\"ms_per_step\"], j[\"roofline\"][\"mfma_f32_TFLOPs\"])"; done
for w in train hstu esasrec; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline $( [ $w = train ] || echo "--steps 20 --warmup 4" ) > gpurun_out/f_$w.json 2> gpurun_out/f_$w.err
  python - $w <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/f_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], {k:j[k] for k in ("value","ms_per_step","final_loss","steps")})
    for k,v in list(j["kernel_breakdown"].items())[:9]: print(f"  {k:28s} {v}")
except Exception as e: print("parse failed", e)
PY
done
