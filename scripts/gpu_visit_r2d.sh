#!/bin/bash
# Round-2 visit D: fused STU block, packed-QKV attention op, LayerNorm-with-skip / dropout+add ops: whole GPU suite + family steps.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r2d; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250
for w in train bert4rec hstu esasrec; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
j=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1]); print("$w", j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in list(j["kernel_breakdown"].items())[:7]})
PY
done
