#!/bin/bash
# Round-2 evidence visit: the default three-leg bench line, rocprofv3 kernel traces of every leg and of the three other
# model families, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) and the SQ counters of
# the attention and top-k kernels.  Everything lands under gpurun_out/r2/ (summaries are then copied into profiles/).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/${R2_OUT:-r2}; mkdir -p $O; export TMPDIR=/tmp
LEAN=${R2_LEAN:-}   # R2_LEAN=1: only what the GEMM arithmetic touches (train traces + PMC, family lines)

# ---- 1. the line the driver records (defaults: 200 train steps after 20 warm-up, + recommend + topk5m legs)
timeout 900 python bench.py > $O/bench_auto.json 2> $O/bench_auto.err; echo "bench auto rc=$?"
python - <<PY
import json
j=json.loads(open("$O/bench_auto.json").read().strip().splitlines()[-1])
print("train", j["value"], j["ms_per_step"], "roof", j["roofline"]["frac"], "cpu", j["cpu_baseline"]["value"])
for leg in ("recommend_e2e","recommend","topk5m"):
    r=j[leg]; print(leg, r["value"], r.get("ms_per_step"), (r.get("roofline") or {}).get("frac"), (r.get("cpu_baseline") or {}).get("value"))
PY

# ---- 2. kernel traces
prof() { name=$1; shift
  rm -rf $O/prof_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_$name -o p -- python $R/bench.py --no-cpu-baseline "$@" > $O/prof_$name.log 2>&1)
  python scripts/prof_summary.py $(find $O/prof_$name -name "*.db" | head -1) 22 > $O/rocprof_kernel_trace_$name.md
  head -8 $O/rocprof_kernel_trace_$name.md | cut -c1-170
}
prof train --workload train --steps 20 --warmup 5
RT_SIDE_STREAM=0 prof train_single_stream --workload train --steps 20 --warmup 5
if [ -z "$LEAN" ]; then
prof topk5m --workload topk5m --steps 6
prof recommend --workload recommend --steps 6
prof bert4rec --workload bert4rec --steps 10 --warmup 3
prof hstu --workload hstu --steps 10 --warmup 3
prof esasrec --workload esasrec --steps 10 --warmup 3
fi

# ---- 3. family bench lines (kept)
for w in bert4rec hstu esasrec; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
j=json.loads(open("$O/bench_$w.json").read().strip().splitlines()[-1]); print("$w", j["value"], j["ms_per_step"], j["roofline"]["kernel"][:40], j["roofline"]["frac"])
PY
done

# ---- 4. HBM traffic (PMC): FETCH_SIZE and WRITE_SIZE in separate passes
pmc() { name=$1; ctr=$2; shift; shift
  rm -rf $O/pmc_${name}_$ctr
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -o p -- python $R/bench.py --no-cpu-baseline "$@" > $O/pmc_${name}_$ctr.log 2>&1)
  f=$(find $O/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$ctr" <<'PY' | tee $O/pmc_${name}_$ctr.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name")==sys.argv[2]:
        agg[r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:70]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print(f"{sys.argv[2]} {k:70s} calls={len(v)} avg={sum(v)/len(v):.1f} max={max(v):.1f} sum={sum(v):.1f}")
PY
}
if [ -z "$LEAN" ]; then
pmc topk5m FETCH_SIZE --workload topk5m --steps 3
pmc topk5m WRITE_SIZE --workload topk5m --steps 3
fi
RT_SIDE_STREAM=0 pmc train FETCH_SIZE --workload train --steps 3 --warmup 1
RT_SIDE_STREAM=0 pmc train WRITE_SIZE --workload train --steps 3 --warmup 1

# ---- 5. SQ counters: matrix-pipe busy share of the attention kernels (C2 shape) and of the top-k stream kernel
sq() { name=$1; pat=$2; shift; shift
  rm -rf $O/sq_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq_$name -o p -- "$@" > $O/sq_$name.log 2>&1)
  f=$(find $O/sq_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$pat" <<'PY' | tee $O/sq_$name.md
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
}
if [ -z "$LEAN" ]; then
sq attention attn_ python $R/scripts/attn_bench.py
sq attention_l512 attn_ python $R/scripts/attn_bench.py --L 512 --n 5
sq topk5m topk_stream python $R/bench.py --workload topk5m --steps 3 --no-cpu-baseline
sq recommend topk_stream python $R/bench.py --workload recommend --steps 3 --no-cpu-baseline
fi

# ---- 6. the N > 1 code path of bench.py on this one-GPU box (2 ranks over gloo; NOT a scaling number)
[ -z "$LEAN" ] && RT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_auto_2ranks_on_1gpu_gloo.json 2> $O/bench_auto_2ranks_on_1gpu_gloo.err
[ -z "$LEAN" ] && { tail -c 400 $O/bench_auto_2ranks_on_1gpu_gloo.json; echo; }

find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
