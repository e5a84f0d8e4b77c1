#!/bin/bash
# PMC breakdown of the attention kernels on the C2 shape (coop schedule via RT_ATTN_COOP)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
for coop in 0 1; do
rm -rf gpurun_out/pmc_attn
(cd /tmp && RT_ATTN_COOP=$coop timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_attn -o p -- python $R/scripts/attn_bench.py > $R/gpurun_out/pmc_attn.log 2>&1)
f=$(find gpurun_out/pmc_attn -name "*counter_collection.csv" | head -1)
echo "=== coop=$coop  $f"
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if "attn_" not in k: continue
    k=k.split("(")[0].replace("(anonymous namespace)::","")[:50]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    print(k)
    for n,v in sorted(c.items()): print(f"    {n:28s} avg={sum(v)/len(v):14.0f}  n={len(v)}")
PY
done
tail -3 gpurun_out/pmc_attn.log
