#!/bin/bash
# SQ counters of the packed softmax attention kernels on one C2 batch (scripts/attn_ablate.py --once: 20 launches of each kernel, nothing
# else on the device): where the waves' cycles go — issuing (ACTIVE_INST_ANY), stalled at issue (WAIT_INST_ANY), parked on s_waitcnt / a
# barrier (WAIT_ANY) — and the instruction counts behind them.  Two passes (8 SQ slots each), kernel-trace only.
#   bash scripts/gpu/attn_pmc.sh <out dir> [RT_VARLEN_IMPL value]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$1; IMPL=$2; mkdir -p $O; export TMPDIR=/tmp
[ -n "$IMPL" ] && export RT_VARLEN_IMPL=$IMPL
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
B="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
for pass in A B; do
  ctr=$A; [ $pass == B ] && ctr=$B
  rm -rf $O/pmc$pass
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/$O/pmc$pass -o p -- python $R/scripts/attn_ablate.py --once > $R/$O/pmc$pass.log 2>&1)
  f=$(find $O/pmc$pass -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/gpu/pmc_summ.py "$f" | grep -E "v[23]_(fwd|bwd)" | tee $O/pmc$pass.txt
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
