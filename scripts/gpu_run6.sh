#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --workload train --steps 20 --warmup 5 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; tail -3 gpurun_out/bench_train.err; python - <<'PY'
import json
j=json.loads(open("gpurun_out/bench_train.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step","final_loss","cpu_baseline")})
print(j["roofline"])
for k,v in j["kernel_breakdown"].items(): print(f"  {k:24s} {v}")
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_train -o p -- python $R/bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_train.log 2>&1)
python scripts/prof_summary.py $(find gpurun_out/prof_train -name "*.db" | head -1) | cut -c1-190 | head -40 | tee gpurun_out/prof_train.md
find gpurun_out -name "*.db" -size +30M -delete
