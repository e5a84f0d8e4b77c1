#!/bin/bash
# HBM traffic of the training step's kernels: FETCH_SIZE and WRITE_SIZE in separate PMC passes (kernel-trace only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
rm -rf gpurun_out/pmc_train_$ctr
(cd /tmp && RT_SIDE_STREAM=0 timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_train_$ctr -o p -- python $R/bench.py --no-cpu-baseline --workload train --steps 3 --warmup 1 > $R/gpurun_out/pmc_train_$ctr.log 2>&1)
f=$(find gpurun_out/pmc_train_$ctr -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" "$ctr" <<'PY' | tee gpurun_out/pmc_train_$ctr.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name")==sys.argv[2]:
        agg[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:58]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:16]:
    print(f"{sys.argv[2]} {k:58s} calls={len(v)} avg={sum(v)/len(v):.1f} sum={sum(v):.1f}")
PY
done
find gpurun_out -name "*.csv" -size +5M -delete
