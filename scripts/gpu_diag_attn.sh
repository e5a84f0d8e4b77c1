#!/bin/bash
# Diagnostic: where do the attention waves wait?  SQ wait / level counters for the resident and ring families at the C2 shape.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/diag; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|SQC|GRBM|TCP|TA|SPI)_[A-Z0-9_]+" | sort -u > $O/counters.txt); wc -l $O/counters.txt
grep -E "WAIT|LEVEL|WAVES|BARRIER|IFETCH|ICACHE|OCCUP|SPI_RA|BUSY" $O/counters.txt | tr '\n' ' ' | cut -c1-3000; echo
run() { tag=$1; impl=$2; shift; shift
  rm -rf $O/pmc_$tag
  (cd /tmp && RT_ATTN_IMPL=$impl timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$tag -o p -- python $R/scripts/attn_bench.py --n 5 > $O/pmc_$tag.log 2>&1)
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if "attn_" not in k: continue
    k=k.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in agg.items():
    print(k, {n: round(sum(v)/len(v)) for n,v in c.items()})
PY
}
for impl in res ring; do
  run ${impl}_a $impl SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  run ${impl}_b $impl SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
  run ${impl}_c $impl SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
done 2>&1 | cut -c1-600
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
