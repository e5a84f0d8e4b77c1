cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for w in train recommend; do
  RT_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --workload $w --steps 5 --warmup 2 > gpurun_out/bench_${w}_2ranks_1gpu.json 2> gpurun_out/bench_${w}_2ranks_1gpu.err
  echo "rc=$?"; head -c 700 gpurun_out/bench_${w}_2ranks_1gpu.json; echo; tail -5 gpurun_out/bench_${w}_2ranks_1gpu.err | cut -c1-300
done
