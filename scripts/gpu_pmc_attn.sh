#!/bin/bash
# SQ counters of the attention kernels on the C2 shape (scripts/attn_bench.py): matrix-pipe busy share per kernel.
# Counter pass only (kernel-trace + pmc), summaries land in gpurun_out/pmc_attn_<tag>.txt.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-dma}
rm -rf gpurun_out/pmc_attn
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_attn -o p -- python $R/scripts/attn_bench.py > $R/gpurun_out/pmc_attn.log 2>&1)
f=$(find gpurun_out/pmc_attn -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee gpurun_out/pmc_attn_$tag.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if "attn_" not in k: continue
    k=k.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:60]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | launches | GRBM_GUI_ACTIVE | SQ_BUSY_CYCLES | SQ_VALU_MFMA_BUSY_CYCLES | MFMA busy / (GUI_ACTIVE x 4 SIMD x 256 CU) | MFMA busy / SQ_BUSY_CYCLES / 4 | LDS bank conflict / LDS active |")
print("|---|---|---|---|---|---|---|---|")
for k,c in agg.items():
    m=lambda n: (sum(c[n])/len(c[n])) if c.get(n) else float("nan")
    gui, mf, sqb = m("GRBM_GUI_ACTIVE"), m("SQ_VALU_MFMA_BUSY_CYCLES"), m("SQ_BUSY_CYCLES")
    print(f"| `{k}` | {len(c['GRBM_GUI_ACTIVE'])} | {gui:.0f} | {sqb:.0f} | {mf:.0f} | {mf/(gui*4*256):.3f} | {mf/sqb/4 if sqb==sqb and sqb>0 else float('nan'):.3f} | {m('SQ_LDS_BANK_CONFLICT')/max(m('SQ_ACTIVE_INST_LDS'),1):.3f} |")
PY
tail -3 gpurun_out/pmc_attn.log
