#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for w in bert4rec hstu esasrec; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 4 > gpurun_out/fam_$w.json 2> gpurun_out/fam_$w.err; echo "$w rc=$?"; tail -2 gpurun_out/fam_$w.err | cut -c1-300
  python - $w <<'PY'
import json,sys
try:
    j=json.loads(open(f"gpurun_out/fam_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print({k:j[k] for k in ("value","ms_per_step","final_loss","steps")}); print({k:v for k,v in j["roofline"].items() if k!="kernel"}, j["roofline"]["kernel"][:60])
    for k,v in list(j["kernel_breakdown"].items())[:14]: print(f"  {k:28s} {v}")
    print(j["config"]["workload"][:200], "prep", j["config"]["dataset_prep_s"])
except Exception as e: print("parse failed", e)
PY
done
