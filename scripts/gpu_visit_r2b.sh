#!/bin/bash
# Round-2 visit B: ring attention kernels — correctness (all three families), A/B timing at the C2 and C4 shapes, SQ counters.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r2b; mkdir -p $O; export TMPDIR=/tmp
echo "== auto (ring) : attention + transformer + baseline-shape + model tests"
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_transformer_gpu.py tests/test_baseline_shapes_gpu.py tests/test_models_gpu.py tests/test_checkpoint.py tests/test_trajectory_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-200
for impl in res stream; do
  echo "== RT_ATTN_IMPL=$impl : attention tests"
  RT_ATTN_IMPL=$impl timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_baseline_shapes_gpu.py -m gpu -q -p no:cacheprovider -k "mha or hstu or dropout or stu" 2>&1 | tail -3 | cut -c1-200
done
echo "== timing"
for impl in ring res stream; do RT_ATTN_IMPL=$impl timeout 120 python scripts/attn_bench.py 2>&1 | grep "^\[" ; done | tee $O/attn_bench_c2.txt
for impl in ring stream; do RT_ATTN_IMPL=$impl timeout 120 python scripts/attn_bench.py --L 512 --hstu 2>&1 | grep "^\[" ; done | tee $O/attn_bench_c4_hstu.txt
for impl in ring stream; do RT_ATTN_IMPL=$impl timeout 120 python scripts/attn_bench.py --L 512 2>&1 | grep "^\[" ; done | tee $O/attn_bench_l512_softmax.txt
echo "== SQ counters (ring, C2 shape)"
rm -rf $O/sq_attention
(cd /tmp && RT_ATTN_IMPL=ring timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq_attention -o p -- python $R/scripts/attn_bench.py > $O/sq_attention.log 2>&1)
f=$(find $O/sq_attention -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" attn_ <<'PY' | tee $O/sq_attention_ring.md
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    k=k.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:64]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: busy share = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024)")
print("| kernel | launches | GRBM_GUI_ACTIVE | SQ_BUSY_CYCLES | SQ_VALU_MFMA_BUSY_CYCLES | matrix-pipe busy share | SQ_ACTIVE_INST_VALU | SQ_ACTIVE_INST_LDS | SQ_LDS_BANK_CONFLICT |")
print("|---|---|---|---|---|---|---|---|---|")
for k,c in agg.items():
    m=lambda n: (sum(c[n])/len(c[n])) if c.get(n) else float("nan")
    gui, mf = m("GRBM_GUI_ACTIVE"), m("SQ_VALU_MFMA_BUSY_CYCLES")
    print(f"| `{k}` | {len(c['GRBM_GUI_ACTIVE'])} | {gui:.0f} | {m('SQ_BUSY_CYCLES'):.0f} | {mf:.0f} | {mf/(gui/8*1024):.3f} | {m('SQ_ACTIVE_INST_VALU'):.0f} | {m('SQ_ACTIVE_INST_LDS'):.0f} | {m('SQ_LDS_BANK_CONFLICT'):.0f} |")
PY
echo "== train / hstu steps with the ring kernels"
for w in train hstu; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
j=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1]); print("$w", j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in list(j["kernel_breakdown"].items())[:6]})
PY
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; find $O -name "*agent_info.csv" -delete
