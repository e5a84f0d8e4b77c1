#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
for ld in 0 2; do
rm -rf gpurun_out/pmc_topk
(cd /tmp && RT_TOPK_LOADERS=$ld timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_topk -o p -- python $R/bench.py --workload topk5m --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_topk.log 2>&1)
f=$(find gpurun_out/pmc_topk -name "*counter_collection.csv" | head -1)
echo "=== loaders=$ld"
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if "topk_stream" not in k: continue
    agg["topk_stream"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    for n,v in sorted(c.items()): print(f"    {n:28s} avg={sum(v)/len(v):14.0f}  max={max(v):14.0f} n={len(v)}")
PY
done
