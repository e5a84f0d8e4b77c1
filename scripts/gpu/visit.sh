#!/bin/bash
# One parameterised GPU visit (replaces the ~35 one-shot gpu_visit*.sh / gpu_*.sh scripts of rounds 1-2):
#   gpurun --timeout 900 -- 'bash scripts/gpu/visit.sh <name> <step> [<step> ...]'
# Steps (results land in gpurun_out/<name>/<n>_<kind>.*; the judged summaries are copied to profiles/ afterwards):
#   tests[:<pytest args>]          pytest -m gpu (default: the whole suite)
#   smoke                          __graft_entry__.smoke()
#   bench[:<bench.py args>]        one bench.py run, headline numbers printed
#   env:<A=B,C=D>:<bench.py args>  bench.py under extra environment variables
#   prof:<bench.py args>           rocprofv3 --kernel-trace of bench.py -> per-kernel table (scripts/prof_summary.py)
#   pmc:<COUNTERS>:<bench.py args> rocprofv3 --kernel-trace --pmc <COUNTERS> (own pass, no other trace domain) -> per-kernel averages
#   py:<script and args>  /  sh:<command>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; NAME=${1:-visit}; shift
O=gpurun_out/$NAME; mkdir -p $O; export TMPDIR=/tmp
i=0
for step in "$@"; do
  i=$((i+1)); kind=${step%%:*}; arg=${step#*:}; [ "$arg" == "$step" ] && arg=""
  echo "== [$i] $step"
  case $kind in
    tests) timeout 900 python -m pytest ${arg:-tests} -m gpu -q --maxfail=25 --tb=short -p no:cacheprovider 2>&1 | tail -120 > $O/${i}_pytest.txt; tail -40 $O/${i}_pytest.txt | cut -c1-220 ;;
    smoke) timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -4 ;;
    bench) timeout 600 python bench.py $arg > $O/${i}_bench.json 2> $O/${i}_bench.err; tail -c 800 $O/${i}_bench.err; python scripts/gpu/summ.py $O/${i}_bench.json ;;
    env)   vars=${arg%%:*}; rest=${arg#*:}; env $(echo $vars | tr ',' ' ') timeout 600 python bench.py $rest > $O/${i}_bench.json 2> $O/${i}_bench.err; tail -c 800 $O/${i}_bench.err; python scripts/gpu/summ.py $O/${i}_bench.json ;;
    prof)  rm -rf $O/${i}_prof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$O/${i}_prof -o p -- python $R/bench.py --no-cpu-baseline $arg > $R/$O/${i}_prof.json 2> $R/$O/${i}_prof.err)
           python scripts/prof_summary.py $(find $O/${i}_prof -name "*.db" | head -1) 45 > $O/${i}_prof.md 2>&1; head -34 $O/${i}_prof.md | cut -c1-180
           python scripts/prof_summary.py $(find $O/${i}_prof -name "*.db" | head -1) --timeline > $O/${i}_timeline.txt 2>&1 ;;
    pmc)   ctr=${arg%%:*}; rest=${arg#*:}; rm -rf $O/${i}_pmc
           (cd /tmp && RT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --pmc $(echo $ctr | tr ',' ' ') --output-format csv -d $R/$O/${i}_pmc -o p -- python $R/bench.py --no-cpu-baseline $rest > $R/$O/${i}_pmc.log 2>&1)
           f=$(find $O/${i}_pmc -name "*counter_collection.csv" | head -1)
           [ -n "$f" ] && python scripts/gpu/pmc_summ.py "$f" | tee $O/${i}_pmc.txt ;;
    py)    timeout 900 python $arg 2>&1 | tail -80 | tee $O/${i}_py.txt ;;
    sh)    timeout 900 bash -c "$arg" 2>&1 | tail -80 | tee $O/${i}_sh.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +5M -delete
