#!/bin/bash
mkdir -p gpurun_out/r6_ns
timeout 900 python -m pytest tests/test_native_step_gpu.py -x -q > gpurun_out/r6_ns/1_test.txt 2>&1
tail -30 gpurun_out/r6_ns/1_test.txt
timeout 300 python scripts/host_profile.py 300 > gpurun_out/r6_ns/2_host.txt 2>&1
head -30 gpurun_out/r6_ns/2_host.txt | cut -c1-170
RT_NATIVE_STEP=0 timeout 300 python scripts/host_profile.py 300 2>&1 | grep "steps:" > gpurun_out/r6_ns/2_host_autograd.txt
cat gpurun_out/r6_ns/2_host_autograd.txt
timeout 600 python bench.py --workload train > gpurun_out/r6_ns/3_bench_train.json 2> gpurun_out/r6_ns/3_bench_train.err
python -c "
import json
b=json.loads(open('gpurun_out/r6_ns/3_bench_train.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], {k:v for k,v in b.items() if 'host' in k})
"
