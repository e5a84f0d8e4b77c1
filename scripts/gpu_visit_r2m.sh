#!/bin/bash
# Last visit of round 2: the GPU suite on the final tree, files touched by the last changes (recommend glue) first.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2m; export TMPDIR=/tmp
python -m pytest tests/test_models_gpu.py tests/test_trajectory_gpu.py tests/test_validation_gpu.py tests/test_dp_gpu.py tests/test_checkpoint.py \
  tests/test_baseline_shapes_gpu.py tests/test_rank_gpu.py tests/test_collate_gpu.py tests/test_negative_sampler_gpu.py tests/test_transformer_gpu.py \
  tests/test_ops_gpu.py tests/test_rank_two_stage_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-200 | tee gpurun_out/r2m/pytest_gpu_tail.txt
