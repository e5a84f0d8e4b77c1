"""bench.py contract (CPU): the committed bench lines carry every field the driver and the judge read, with consistent
arithmetic; without a GPU the script fails loudly instead of measuring something else."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline"]


def _lines():
    prof = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof)):
        if name[:1] == "r" and "_bench_" in name[:12] and name.endswith(".json"):     # r1_bench_*.json, r2_bench_*.json, ...
            text = [l for l in open(os.path.join(prof, name)).read().splitlines() if l.startswith("{")]
            assert len(text) == 1, f"{name}: rank 0 must print exactly ONE JSON line"
            yield name, json.loads(text[0])


@pytest.mark.parametrize("name,line", list(_lines()), ids=[n for n, _ in _lines()])
def test_committed_bench_lines_follow_the_contract(name, line):
    for key in REQUIRED:
        assert key in line, f"{name}: missing {key}"
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["data"] == "synthetic" and line["dtype"] == "fp32"
    assert line["vs_baseline"] is None                      # BASELINE.md publishes no number for these metrics
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, f"{name}: roofline.{key}"
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 2e-3
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["n_gpus"] >= 1
    if line["cpu_baseline"] is not None:
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"]
        assert line["cpu_baseline"]["kind"] in ("port", "reference") and line["cpu_baseline"]["unit"] == line["unit"]


def test_default_bench_line_is_the_baseline_metric():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    line = dict(_lines())["r1_bench_train.json"]
    assert "train seqs/sec" in base["metric"] and line["metric"].startswith("train seqs/sec") and line["unit"] == "seqs/s"
    assert line["n_gpus"] == 1 and line["cpu_baseline"] is not None and line["roofline"]["traffic"] is not None
    assert "d=256" in line["config"]["workload"] and "L=200" in line["config"]["workload"]


def test_bench_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and not any(l.startswith("{") for l in p.stdout.splitlines())   # no number without the HIP path


def _run_bench(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun around it) re-launches itself as 2 ranks and reports the world size the process
    group has; a world size that contradicts --gpus is refused.  Dry run: launcher + gloo rendezvous + line, no device work."""
    p = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"RT_BENCH_DRY_RUN": "1"})
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-2000:]
    assert lines[0]["n_gpus"] == 2 and lines[0]["dist"]["world_size"] == 2 and lines[0]["dist"]["ranks_counted_by_allreduce"] == 2
    env = dict(os.environ, RT_BENCH_DRY_RUN="1", WORLD_SIZE="1", RANK="0")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert q.returncode != 0 and "refusing to report n_gpus=2" in (q.stderr + q.stdout)
    assert not any(l.startswith("{") for l in q.stdout.splitlines())


@pytest.mark.gpu
def test_bench_gpus_2_runs_end_to_end_on_one_box():
    """The whole N > 1 bench path from a bare `python bench.py --gpus 2`: on a box with one GPU the two ranks share it and
    rendezvous over gloo (the line says so); with two or more devices it is RCCL, one rank per GPU."""
    import torch

    p = _run_bench(["--gpus", "2", "--workload", "train", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], {})
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-3000:])
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2 and line["config"]["global_batch"] == 256
    assert line["dist"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    if line["dist"]["backend"] == "nccl":
        assert line["dist"]["rccl_ranks"] == 2
    assert line["value"] > 0 and line["final_loss"] > 0
