#!/bin/bash
mkdir -p gpurun_out/r6_late
run() { # name env
  env $2 timeout 300 python bench.py --workload train --steps 200 --warmup 30 --no-cpu-baseline --no-families --no-host-only > gpurun_out/r6_late/$1.json 2> gpurun_out/r6_late/$1.err
  python -c "
import json,sys
b=json.loads(open('gpurun_out/r6_late/$1.json').read().strip().splitlines()[-1])
print('$1', b['value'], b['ms_per_step'], b['final_loss'], b['roofline'].get('step_issue'))
" || tail -5 gpurun_out/r6_late/$1.err
}
for r in a b; do
run t0_$r RT_STEP_TABLE_ISSUE=0
run t1_$r RT_STEP_TABLE_ISSUE=1
run t2_$r RT_STEP_TABLE_ISSUE=2
run autograd_$r RT_NATIVE_STEP=0
done
