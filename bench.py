#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X engine (contract: see the round brief / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W [--workload auto|train|recommend|topk5m]

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
  * train      : one SASRec training step (collated batch -> fwd -> loss -> bwd -> Adam [-> RCCL all-reduce])
                 on BASELINE.json configs[1] (d=256, 2 blocks, L=200, sampled_softmax, ML-20M-shaped);
  * recommend  : top-k (k=10, filter_viewed) for one batch of users against the ML-20M-shaped catalog;
  * topk5m     : top-k over the 5M x 512 synthetic catalog (BASELINE.json configs[4], the HBM-roofline run).
Rank 0 prints ONE JSON line.  N>1 is launched by torch.distributed.run (one rank per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_F32_PEAK_TF = 157.3


def dist_setup(n_gpus: int):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if world != n_gpus:
        if rank == 0:
            print(f"[bench] warning: --gpus {n_gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    return rank, world, local


def barrier_sync(world: int):
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x: float, world: int) -> float:
    if world == 1:
        return x
    import torch.distributed as dist

    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step_fn, steps: int, warmup: int, world: int):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize; per-step HIP events on the
    current stream (the stream every rt_* kernel is launched on) give the kernel-side duration."""
    for _ in range(warmup):
        step_fn()
    barrier_sync(world)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i][0].record()
        step_fn()
        evs[i][1].record()
    barrier_sync(world)
    wall = time.perf_counter() - t0
    ev_ms = [a.elapsed_time(b) for a, b in evs]
    return max_over_ranks(wall, world), float(np.mean(ev_ms))


# ---------------------------------------------------------------------------------------------------
# recommend / top-k workloads
# ---------------------------------------------------------------------------------------------------
def topk_bytes(n_items: int, d: int, n_users: int, k: int, nnz: int) -> float:
    """Algorithmic bytes of one user batch: catalog once + users + outputs + filter."""
    return 4.0 * d * n_items + 4.0 * d * n_users + 12.0 * n_users * k + 4.0 * nnz + 8.0 * (n_users + 1)


def make_topk_workload(n_items: int, d: int, users_per_step: int, upp: int, rank: int, with_filter: bool, seed: int):
    from rectools_amd.rank import HipRanker
    from rectools_amd import synth

    g = torch.Generator(device="cuda").manual_seed(seed + 2)
    items = torch.randn(n_items, d, generator=g, device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda").manual_seed(seed + 1 + 1000 * rank)
    users = torch.randn(users_per_step, d, generator=g, device="cuda", dtype=torch.float32)
    filt = None
    nnz = 0
    if with_filter:
        u, i, _ = synth.gen_interactions(users_per_step, n_items, mean_len=144.0, min_len=20, max_len=2000,
                                         seed=seed + 7 + rank)
        filt = synth.viewed_csr(u, i, users_per_step, n_items)
        nnz = int(filt.nnz)
    from rectools_amd.rank import DeviceCSR

    ranker = HipRanker("dot", "cuda", users, items, batch_size=upp)
    sids = np.arange(users_per_step)
    # inputs resident in HBM before the timed region: factors, and the viewed-items CSR
    dfilt = DeviceCSR.from_scipy(filt, "cuda") if filt is not None else None
    state = {}

    def step():
        state["out"] = ranker.rank_device(sids, k=10, filter_pairs_csr=dfilt)

    return step, ranker, dict(n_items=n_items, d=d, users=users_per_step, nnz=nnz, items=items, users_t=users,
                              filt=filt)


def cpu_baseline_topk(items_t: torch.Tensor, users_t: torch.Tensor, filt, budget_s: float = 15.0):
    """Oracle (numpy port of TorchRanker.rank) timed on the host cores on a bounded user sample."""
    from oracle import ranker_oracle

    items = items_t.cpu().numpy()
    users = users_t.cpu().numpy()
    n = min(users.shape[0], 256)
    t0 = time.perf_counter()
    done = 0
    while True:
        sl = np.arange(done % max(users.shape[0] - n + 1, 1), done % max(users.shape[0] - n + 1, 1) + n)
        f = filt[sl] if filt is not None else None
        ranker_oracle.rank(users, items, sl, k=10, filter_pairs_csr=f, batch_size=128)
        done += n
        el = time.perf_counter() - t0
        if el > budget_s or done >= 8 * n:
            break
    return done / el, done


def run_topk(args, rank, world, n_items, d, users_per_step, upp, with_filter, name):
    """One step = one rt_topk_score launch sequence over `users_per_step` users (inputs resident in HBM)."""
    step, ranker, info = make_topk_workload(n_items, d, users_per_step, upp, rank, with_filter, seed=0)
    wall, ev_ms = timed_steps(step, args.steps, args.warmup, world)
    users_total = users_per_step * args.steps * world
    value = users_total / wall
    # algorithmic bytes / flops of ONE launch: catalog read once for the whole user batch
    bytes_per_launch = topk_bytes(n_items, d, users_per_step, 10, info["nnz"])
    flops_per_launch = 2.0 * users_per_step * n_items * d
    t = ev_ms * 1e-3
    gbs = bytes_per_launch / t / 1e9
    tfs = flops_per_launch / t / 1e12
    hbm_bound = (bytes_per_launch / (HBM_PEAK_GBS * 1e9)) >= (flops_per_launch / (MFMA_F32_PEAK_TF * 1e12))
    if n_items * d * 4 <= 200e6:
        hbm_bound = False  # catalog resident in L2 / Infinity Cache: the HBM roof does not apply
    roof = {
        "kernel": "topk_stream_kernel + topk_merge_kernel (one rt_topk_score call)",
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": round(gbs if hbm_bound else tfs, 2),
        "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F32_PEAK_TF,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tfs / MFMA_F32_PEAK_TF), 4),
        "traffic": load_traffic(name),
        "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_flops_per_launch": flops_per_launch,
        "avg_launch_ms": round(ev_ms, 4), "hbm_GBps": round(gbs, 1), "mfma_f32_TFLOPs": round(tfs, 2),
        "users_per_launch": users_per_step, "users_per_register_tile": upp,
    }
    return value, wall, roof, info


def load_traffic(name: str):
    """PMC-measured HBM bytes per launch (profiles/traffic.json, written from rocprofv3 --pmc passes)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(name)
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="auto", choices=["auto", "train", "recommend", "topk5m"])
    ap.add_argument("--users-per-pass", type=int, default=0, help="register tile: 32/64/128 users (0 = auto)")
    ap.add_argument("--users-per-step", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    rank, world, local = dist_setup(args.gpus)

    from rectools_amd import _lib, synth

    _lib.load()  # fail loudly if the HIP extension is missing
    workload = args.workload
    if workload == "auto":
        workload = "recommend"

    extra = {}
    if workload == "recommend":
        if args.steps is None:
            args.steps = 20
        if args.warmup is None:
            args.warmup = 3
        V, d = synth.ML_20M["n_items"], 256
        users_per_step = args.users_per_step or 16384
        upp = args.users_per_pass or 64
        value, wall, roof, info = run_topk(args, rank, world, V, d, users_per_step, upp, True, "recommend_ml20m")
        metric, unit = "recommend() users/sec @k=10 (SASRec d=256, ML-20M-shaped catalog, filter_viewed)", "users/s"
        config = {"workload": f"recommend top-k: 26744 items x d256 fp32, {users_per_step} users/step, k=10, viewed-filter CSR",
                  "users_per_step": users_per_step, "users_per_register_tile": upp, "parallelism": f"dp{world}"}
    elif workload == "topk5m":
        if args.steps is None:
            args.steps = 3
        if args.warmup is None:
            args.warmup = 1
        V, d = 5_000_000, 512
        users_per_step = args.users_per_step or 32  # 32 users/launch: the HBM-bound regime (AI = B/2 flop/B)
        upp = args.users_per_pass or (32 if users_per_step <= 32 else 64)
        value, wall, roof, info = run_topk(args, rank, world, V, d, users_per_step, upp, False, "topk5m")
        metric, unit = "full-catalog top-k users/sec @k=10 (5M x 512 fp32 catalog)", "users/s"
        config = {"workload": f"top-k scoring: 5,000,000 items x d512 fp32 (10.24 GB), {users_per_step} users/step, k=10",
                  "users_per_step": users_per_step, "users_per_register_tile": upp, "parallelism": f"dp{world}"}
    else:
        raise SystemExit("train workload is not built yet in this revision")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize()
        v, n = cpu_baseline_topk(info["items"] if info["n_items"] <= 100_000 else info["items"][:200_000],
                                 info["users_t"], info["filt"])
        scale = 1.0 if info["n_items"] <= 100_000 else 200_000 / info["n_items"]
        cpu = {"value": round(v * scale, 2), "unit": unit, "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"oracle/ranker_oracle.rank (numpy) on {n} users"
                         + ("" if scale == 1.0 else f", first 200k catalog rows, rate scaled by {scale:.3f}")}

    if rank == 0:
        out = {
            "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
            "config": config, "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
