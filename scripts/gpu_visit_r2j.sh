#!/bin/bash
# Round-2 visit J: SQ counters of the GEMM kernels (matrix-pipe busy share, VALU instructions, effective clock) for the bf16x6 loop and
# the exact loop on the products of scripts/gemm_bench.py, and the > 48-segment Adam test.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r2j; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "adam" 2>&1 | tail -3
sq() { name=$1; pat=$2; shift; shift
  rm -rf $O/sq_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$name -o p -- "$@" > $O/sq_$name.log 2>&1)
  f=$(find $O/sq_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$pat" <<'PY' | tee $O/sq_$name.md
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
dur=collections.defaultdict(list)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    key=(k, r.get("Grid_Size"))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: busy share = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024); clock = GUI_ACTIVE / 8 / duration")
print("| kernel | grid | launches | avg us | GRBM_GUI_ACTIVE / 8 | eff. clock GHz | SQ_VALU_MFMA_BUSY_CYCLES | matrix-pipe busy share | SQ_ACTIVE_INST_VALU | SQ_ACTIVE_INST_LDS | SQ_LDS_BANK_CONFLICT | SQ_WAVE_CYCLES |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for key,c in sorted(agg.items(), key=lambda kv: -len(kv[1]["GRBM_GUI_ACTIVE"])):
    m=lambda n: (sum(c[n])/len(c[n])) if c.get(n) else float("nan")
    gui, mf = m("GRBM_GUI_ACTIVE"), m("SQ_VALU_MFMA_BUSY_CYCLES")
    d=dur.get(key); us=sum(d)/len(d) if d else float("nan")
    name=key[0].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:60]
    print(f"| `{name}` | {key[1]} | {len(c['GRBM_GUI_ACTIVE'])} | {us:.1f} | {gui/8:.0f} | {gui/8/(us*1e3) if us==us else float('nan'):.2f} | {mf:.0f} | {mf/(gui/8*1024):.3f} | {m('SQ_ACTIVE_INST_VALU'):.0f} | {m('SQ_ACTIVE_INST_LDS'):.0f} | {m('SQ_LDS_BANK_CONFLICT'):.0f} | {m('SQ_WAVE_CYCLES'):.0f} |")
PY
}
sq gemm_bf16x6 gemm_dma python $R/scripts/gemm_bench.py
RT_GEMM_SPLIT=exact sq gemm_exact gemm_dma python $R/scripts/gemm_bench.py
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; find $O -name "*agent_info.csv" -delete
