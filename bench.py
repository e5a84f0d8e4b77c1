#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X engine (contract: see the round brief / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W [--workload auto|train|recommend|topk5m]

BASELINE.json's metric is "train seqs/sec + recommend() users/sec@k=10, SASRec d=256 ML-20M" plus the HBM-roofline run of
the full-catalog top-k.  The default (`auto`) run therefore has three legs and prints ONE JSON line on rank 0:
  * train (the headline `value`; K timed steps after W warm-up steps): the product loop of `SASRecModel.fit()` on
    BASELINE.json configs[1] (d=256, 2 blocks, 4 heads, L=200, dropout 0.2, sampled_softmax N=128, ML-20M-shaped data):
    the model is built by the public API from a synthetic `Dataset`, and one step = `models._TrainLoop.step()` =
    device collate out of the HBM-resident session store (rt_collate) -> negatives (rt_sample_negatives) -> forward ->
    loss -> backward -> [RCCL all-reduce ->] fused Adam — nothing pre-collated, nothing replayed;
  * `recommend` sub-record: top-k (k=10, filter_viewed) of 16,384 users per step against the 26,744 x 256 catalog, and
    `recommend_e2e`: one `model.recommend()` call through the public API (session encoding + top-k + frame assembly);
  * `topk5m` sub-record: top-k over the 5M x 512 synthetic catalog (BASELINE.json configs[4], the HBM-roofline run).
Each leg carries its own `roofline` (HIP events on the launch stream) and `cpu_baseline` (the oracle on the host cores).
`--workload train|recommend|topk5m` runs one leg alone (profiling scripts).  N>1 is launched by torch.distributed.run
(one rank per GPU, RCCL); every rank runs every leg on its own shard, `value`s are whole-job aggregates.
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
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
# The exact-tile GEMMs compute every fp32 product as six bf16-MFMA products of an exact 3-way bf16 split (rt_gemm.hip:
# split_bf16x3; error against fp64 <= the f32-input MFMA kernel's, tests/test_ops_gpu.py): an algorithmic fp32 flop costs six
# bf16 flops, so the pipe's peak for this kernel is 2500 / 6.  RT_GEMM_SPLIT=exact runs v_mfma_f32_32x32x2_f32 (157.3 TF).
GEMM_X6 = os.environ.get("RT_GEMM_SPLIT", "bf16x6") != "exact"
GEMM_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0 if GEMM_X6 else MFMA_F32_PEAK_TF


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_gpus: int) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves — re-exec under
    `torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 (the driver's own form) and hand its exit code back.  With
    fewer visible devices than ranks (a 1-GPU test box) the ranks share devices and rendezvous over gloo (RCCL refuses two ranks
    on one device); the line then says so (`dist.backend`, `dist.devices_visible`) — it is a path check, not a scaling number."""
    if n_gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("RT_BENCH_DRY_RUN") != "1" and torch.cuda.is_available() and torch.cuda.device_count() < n_gpus:
        env.setdefault("RT_BENCH_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def dist_setup(n_gpus: int):
    """-> (rank, world, local device index, description of the process group for the line).  The world size the line reports is
    what the initialised process group says — never the `--gpus` argument — and a mismatch between the two is an error."""
    self_launch(n_gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = os.environ.get("RT_BENCH_DRY_RUN") == "1"      # tests/test_bench_contract.py: launcher + rendezvous + line, no device work
    info = {"backend": None, "world_size": 1, "devices_visible": 0 if dry else torch.cuda.device_count(), "rccl_ranks": 0,
            "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ or "WORLD_SIZE" in os.environ else "single process"}
    if world > 1:
        import torch.distributed as dist

        # RT_BENCH_BACKEND=gloo lets the N>1 path run on a box with fewer GPUs than ranks; the driver's multi-GPU runs use nccl =
        # RCCL, one rank per GPU.
        backend = "gloo" if dry else os.environ.get("RT_BENCH_BACKEND", "nccl")
        if not dry:
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # what the communicator itself says: every rank contributes 1 (over RCCL when the backend is nccl)
        ones = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        info.update(backend=backend, world_size=dist.get_world_size(), rccl_ranks=int(ones.item()) if backend == "nccl" else 0,
                    ranks_counted_by_allreduce=int(ones.item()))
        world = dist.get_world_size()
    elif not dry:
        torch.cuda.set_device(0)
    if world != n_gpus:
        raise SystemExit(f"[bench] --gpus {n_gpus} but the process group has {world} rank(s): refusing to report n_gpus={n_gpus} "
                         f"(launch with torch.distributed.run --nproc-per-node {n_gpus}, or plain `python bench.py --gpus {n_gpus}`)")
    return rank, world, local, info


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


def gather_floats(x: float, world: int):
    """-> the value of every rank (rank order), on every rank."""
    if world == 1:
        return [x]
    import torch.distributed as dist

    t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def timed_steps(step_fn, steps: int, warmup: int, world: int, finish=None):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize; per-step HIP events on the
    current stream (the stream every rt_* kernel is launched on) give the kernel-side duration.  finish: called once behind the K
    steps, INSIDE the timed region (the top-k legs: `HipRanker.settle()` — the one read of all the calls' proof flags)."""
    for _ in range(warmup):
        step_fn()
    if finish is not None:
        finish()
    barrier_sync(world)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    stats0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i][0].record()
        step_fn()
        evs[i][1].record()
    issue = time.perf_counter() - t0      # the host has issued every launch of the timed steps (the device may still be working)
    if finish is not None:
        finish()
    barrier_sync(world)
    wall = time.perf_counter() - t0
    ev_ms = [a.elapsed_time(b) for a, b in evs]
    timed_steps.host_issue_ms = issue / steps * 1e3
    # synchronous hipMalloc calls / allocator retries INSIDE the timed steps (packed batches change their row count every step: a
    # reservation that is too small shows up here, and as host time — VERDICT r5 weak #6)
    stats1 = torch.cuda.memory_stats()
    timed_steps.device_allocs = {k: int(stats1.get(k, 0) - stats0.get(k, 0)) for k in ("num_device_alloc", "num_alloc_retries", "num_device_free")}
    # every rank's own wall clock and device time per step: a scaling run that disappoints says which rank lagged (the first SCALE
    # record must be self-diagnosing — VERDICT r3 #9)
    timed_steps.per_rank_ms = [round(w / steps * 1e3, 4) for w in gather_floats(wall, world)]
    timed_steps.per_rank_device_ms = [round(v, 4) for v in gather_floats(float(np.mean(ev_ms)), world)]
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

    ranker = HipRanker("dot", "cuda", users, items, batch_size=upp or None)   # 0: the library's choice (two-stage top-k from 128 users up)
    sids = np.arange(users_per_step)
    # inputs resident in HBM before the timed region: factors, and the viewed-items CSR
    dfilt = DeviceCSR.from_scipy(filt, "cuda") if filt is not None else None
    state = {}

    def step():      # calls queue back to back; their proof flags are read once, by ranker.settle() behind the timed steps (in the timed region)
        state["out"] = ranker.rank_device(sids, k=10, filter_pairs_csr=dfilt, settle=False)

    return step, ranker, dict(n_items=n_items, d=d, users=users_per_step, nnz=nnz, items=items, users_t=users,
                              filt=filt)


def topk_extras(info, users_per_step: int, upp: int):
    """What the timed two-stage call does not show: the one-time cost and footprint of the catalog's coarse-pass image(s)
    — built by the first call on a catalog, kept by the model between recommend() calls (models._catalog_images) — and, for the
    16-user launch, the single-stage kernel on the fp32 rows (the path that reads the 10.24 GB catalog itself)."""
    from rectools_amd.rank import HipRanker

    items, users = info["items"], info["users_t"]
    n, d = items.shape
    sids = np.arange(users_per_step)
    out = {}
    r = HipRanker("dot", "cuda", users, items, batch_size=upp or None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    r.rank_device(sids, k=10)            # first call on this ranker: builds the image(s), then ranks
    e1.record()
    torch.cuda.synchronize()
    first_ms = e0.elapsed_time(e1)
    e0.record()
    r.rank_device(sids, k=10)
    e1.record()
    torch.cuda.synchronize()
    imgs = [getattr(r, a) for a in ("_items_h", "_items_hm", "_items_frag")]
    out["image_build_ms"] = round(first_ms - e0.elapsed_time(e1), 3)
    out["image_bytes"] = int(sum(t.numel() * t.element_size() for t in imgs if t is not None))
    out["image_kinds"] = [a for a, t in zip(("one_plane", "hm", "fragment_major"), imgs) if t is not None]
    del r, imgs
    if users_per_step <= 64:
        single = HipRanker("dot", "cuda", users, items, batch_size=upp or None, two_stage=False)
        for _ in range(3):
            single.rank_device(sids, k=10)
        torch.cuda.synchronize()
        reps = 10
        e0.record()
        for _ in range(reps):
            single.rank_device(sids, k=10)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        byts = topk_bytes(n, d, users_per_step, 10, 0)
        out["single_stage"] = {"kernel": "rt_topk_score (fp32 rows, f32-input MFMA, 16-user tile): reads the catalog itself, no image",
                               "ms_per_step": round(ms, 4), "users_per_s": round(users_per_step / ms * 1e3, 1),
                               "achieved": round(byts / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(byts / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": byts}
        del single
    torch.cuda.empty_cache()
    return out


def quiesce_host(seconds: float = 1.0) -> None:
    """After a CPU-baseline leg: let the 128 OpenMP workers of the reference's torch ops finish spinning before the next GPU leg is timed
    (a leg that started right behind a baseline ran at 2.92 instead of 1.93 ms per step in 2 of 5 driver-form runs, the 16-user top-k at
    3.2 instead of 1.1 ms once: the launching thread was competing for its core)."""
    time.sleep(seconds)


def cpu_baseline_topk(items_t: torch.Tensor, users_t: torch.Tensor, filt, budget_s: float = 15.0):
    """-> (users/s, users ranked, kind, sample): the unmodified reference's TorchRanker(device="cpu").rank when the reference is on this
    box (oracle/cpu_reference.py), else the numpy port (oracle/ranker_oracle), on a bounded user sample."""
    from oracle import cpu_reference

    items = items_t.cpu().numpy()
    users = users_t.cpu().numpy()
    if cpu_reference.available():
        rate, done, what = cpu_reference.rank_rate(users[:2048], items, filt[:2048] if filt is not None else None, budget_s=budget_s)
        return rate, done, "reference", what
    from oracle import ranker_oracle

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
    return done / el, done, "port", f"oracle/ranker_oracle.rank (numpy restatement of TorchRanker.rank) on {done} users"


def run_topk(steps, warmup, rank, world, n_items, d, users_per_step, upp, with_filter, name):
    """One step = one rt_topk_score launch sequence over `users_per_step` users (inputs resident in HBM)."""
    step, ranker, info = make_topk_workload(n_items, d, users_per_step, upp, rank, with_filter, seed=0)
    wall, ev_ms = timed_steps(step, steps, warmup, world, finish=ranker.settle)
    users_total = users_per_step * steps * world
    value = users_total / wall
    # algorithmic bytes / flops of ONE launch: catalog read once for the whole user batch
    bytes_per_launch = topk_bytes(n_items, d, users_per_step, 10, info["nnz"])
    flops_per_launch = 2.0 * users_per_step * n_items * d
    t = ev_ms * 1e-3
    gbs = bytes_per_launch / t / 1e9
    tfs = flops_per_launch / t / 1e12
    two_stage = ranker.two_stage_stats["calls"] > 0
    h_only = ranker.two_stage_stats.get("h_only_calls", 0) > 0
    if h_only:
        # the coarse pass streams a one-plane bf16 image of the catalog: the bytes this formulation has to move per launch are the image's
        # (2 B per value) + the users + the outputs; the fp32 rows are touched for 64 candidates per user only
        bytes_per_launch = 2.0 * d * n_items + 4.0 * d * users_per_step + 12.0 * users_per_step * 10 + 4.0 * d * 64 * users_per_step
        gbs = bytes_per_launch / t / 1e9
    # two-stage calls (rt_topk_score_two_stage) score on the bf16 matrix pipe: over the (h, m) images (h + m)(h' + m') = four bf16 products
    # per fp32 product, so the pipe's roof for ALGORITHMIC fp32 flops is 2,500 / 4 TF; over the one-plane image one product: 2,500 TF;
    # single-stage calls run the f32-input instruction (157.3 TF)
    mfma_peak = MFMA_BF16_PEAK_TF if h_only else MFMA_BF16_PEAK_TF / 4.0 if two_stage else MFMA_F32_PEAK_TF
    hbm_bound = (bytes_per_launch / (HBM_PEAK_GBS * 1e9)) >= (flops_per_launch / (mfma_peak * 1e12))
    if n_items * d * 4 <= 200e6:
        hbm_bound = False  # catalog resident in L2 / Infinity Cache: the HBM roof does not apply
    roof = {
        "kernel": ("rt_topk_score_two_stage call (users' image + coarse stream kernel on v_mfma_f32_32x32x16_bf16" +
                   (" over the ONE-plane bf16 image of the catalog (half the fp32 bytes; exactness from the exact pass + per-user proof)" if h_only else "") +
                   " + merge + exact pass over 64 candidates per user; HIP events around every call, calls issued back to back, the proof flags of all of them read once behind the last one — HipRanker.settle(), inside the timed region)") if two_stage else
                  "rt_topk_score call (seed prefix + topk stream kernel + merge; HIP events around the whole call)",
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": round(gbs if hbm_bound else tfs, 2),
        "peak": HBM_PEAK_GBS if hbm_bound else round(mfma_peak, 1),
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tfs / mfma_peak), 4),
        "two_stage": dict(ranker.two_stage_stats) if two_stage else None,
        "fp32_catalog_equivalent_GBps": round(topk_bytes(n_items, d, users_per_step, 10, info["nnz"]) / t / 1e9, 1) if h_only else None,
        # the one-plane pass walks the image once per 64-user tile (tiles of one XCD walk it together through that XCD's L2): what the
        # compute units pull through the LDS ring per second — the figure that bounds the many-user launch, not the pipe
        "image_ring_GBps": round(2.0 * d * n_items * max(1, (users_per_step + 63) // 64) / t / 1e9, 1) if h_only and users_per_step > 32 else None,
        "traffic": load_traffic(name),
        "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_flops_per_launch": flops_per_launch,
        "avg_launch_ms": round(ev_ms, 4), "hbm_GBps": round(gbs, 1), "mfma_f32_TFLOPs": round(tfs, 2),
        "users_per_launch": users_per_step, "users_per_register_tile": upp or "library default",
    }
    return value, wall, roof, info


def make_dataset(n_users: int, n_items: int, mean_len: float, min_len: int, max_len: int, seed: int = 0, clip_len=None):
    """Synthetic interactions of a given shape (clipped log-normal lengths, Zipf popularity, monotone
    timestamps) as a `Dataset`."""
    import pandas as pd

    from rectools_amd import synth
    from rectools_amd.dataset import Dataset

    t0 = time.perf_counter()
    u, it, ts = synth.gen_interactions(n_users, n_items, mean_len=mean_len, min_len=min_len, max_len=max_len, seed=seed,
                                       clip_len=clip_len)
    df = pd.DataFrame({"user_id": u, "item_id": it, "weight": 1.0, "datetime": pd.to_datetime(ts, unit="s")})
    t1 = time.perf_counter()
    ds = Dataset.construct(df)
    # the synthetic interactions are the bench's own cost; Dataset.construct is the user's (the reference's users call it too)
    make_dataset.last_times = {"synthetic_interactions_s": round(t1 - t0, 2), "dataset_construct_s": round(time.perf_counter() - t1, 2)}
    return ds


def make_ml20m_dataset(seed: int = 0):
    """ML-20M-shaped: 138,493 users / 26,744 items / ~19.9 M rows."""
    from rectools_amd import synth

    shape = {k: v for k, v in synth.ML_20M.items() if k in ("n_users", "n_items", "mean_len", "min_len", "max_len")}
    return make_dataset(seed=seed, **shape)


def family_spec(kind: str, n_neg: int):
    """(model, dataset maker, description, dense fwd flops per sequence and block, loss flops per sequence) of a training leg.
    Block flops per SURVEY.md §8d: SASRec 12Ld^2+4L^2d, PreLN 24Ld^2+4L^2d, LiGR(swiglu x4) 36Ld^2+4L^2d,
    STU 2Ld*4Hhd + 4L^2*H*hd + 2L*H*hd*d."""
    from rectools_amd import nn as hnn
    from rectools_amd.models import BERT4RecModel, HSTUModel, SASRecModel

    if kind == "train":      # BASELINE.json configs[1]
        d, H, nb, L, B = 256, 4, 2, 200, 128
        model = SASRecModel(n_factors=d, n_blocks=nb, n_heads=H, session_max_len=L, dropout_rate=0.2, loss="sampled_softmax",
                            n_negatives=n_neg, batch_size=B, lr=1e-3, epochs=1, seed=32)
        return dict(model=model, ds=make_ml20m_dataset, d=d, H=H, nb=nb, L=L, B=B, n_neg=n_neg,
                    blk=12.0 * L * d * d + 4.0 * L * L * d, loss=2.0 * L * (1 + n_neg) * d,
                    name="SASRec d=256 n_blocks=2 L=200 sampled_softmax, ML-20M-shaped",
                    desc=f"B=128/GPU x L=200, d=256, 2 blocks, 4 heads, dropout 0.2, sampled_softmax N={n_neg}")
    if kind == "bert4rec":   # BASELINE.json configs[2]
        d, H, nb, L, B = 256, 4, 2, 200, 128
        model = BERT4RecModel(n_factors=d, n_blocks=nb, n_heads=H, session_max_len=L, dropout_rate=0.2, loss="softmax",
                              mask_prob=0.15, batch_size=B, lr=1e-3, epochs=1, seed=32)
        V = 26_744 + 2
        return dict(model=model, ds=make_ml20m_dataset, d=d, H=H, nb=nb, L=L, B=B, n_neg=0,
                    blk=24.0 * L * d * d + 4.0 * L * L * d, loss=2.0 * 0.15 * L * V * d,
                    name="BERT4Rec d=256 n_blocks=2 L=200 full softmax (mask_prob 0.15), ML-20M-shaped",
                    desc="B=128/GPU x L=200, d=256, 2 Pre-LN blocks, 4 heads, dropout 0.2, full-catalog softmax over 26,746 classes "
                         "on the ~15 % masked positions")
    if kind == "hstu":       # BASELINE.json configs[3] (model shape; 1M items, 65,536 users resident instead of 10M)
        d, H, nb, L, B = 256, 4, 2, 512, 128
        hd = d // H
        model = HSTUModel(n_factors=d, n_blocks=nb, n_heads=H, session_max_len=L, dropout_rate=0.2, loss="sampled_softmax",
                          n_negatives=n_neg, batch_size=B, lr=1e-3, epochs=1, seed=32, relative_time_attention=True,
                          relative_pos_attention=True, lightning_module_kwargs={"logits_t": 0.05})
        return dict(model=model, ds=lambda: make_dataset(65_536, 1_000_000, 300.0, 20, 3000, seed=0, clip_len=513),
                    d=d, H=H, nb=nb, L=L, B=B, n_neg=n_neg,
                    blk=2.0 * L * d * 4 * H * hd + 4.0 * L * L * H * hd + 2.0 * L * H * hd * d, loss=2.0 * L * (1 + n_neg) * d,
                    name="HSTU d=256 n_blocks=2 H=4 L=512 rel time+pos bias, cosine, sampled_softmax, 1M-item catalog",
                    desc=f"B=128/GPU x L=512, d=256, 2 STU blocks, 4 heads (hd 64), relative time + position bias, cosine, logits_t "
                         f"0.05, sampled_softmax N={n_neg}, V=1,000,000 items, 65,536 user sequences resident (the 10 M-user set "
                         f"of the config is a host-side epoch length, not a per-step cost)")
    if kind == "esasrec":    # BASELINE.json configs[4] model: SASRec data path on LiGR blocks, d=512; 1M-item table for training
        d, H, nb, L, B = 512, 4, 2, 200, 128
        model = SASRecModel(n_factors=d, n_blocks=nb, n_heads=H, session_max_len=L, dropout_rate=0.2, loss="sampled_softmax",
                            n_negatives=n_neg, batch_size=B, lr=1e-3, epochs=1, seed=32, transformer_layers_type=hnn.LiGRLayers)
        return dict(model=model, ds=lambda: make_dataset(65_536, 1_000_000, 144.0, 20, 2000, seed=0),
                    d=d, H=H, nb=nb, L=L, B=B, n_neg=n_neg,
                    blk=36.0 * L * d * d + 4.0 * L * L * d, loss=2.0 * L * (1 + n_neg) * d,
                    name="eSASRec (SASRec + LiGR blocks) d=512 n_blocks=2 L=200 sampled_softmax, 1M-item catalog",
                    desc=f"B=128/GPU x L=200, d=512, 2 LiGR blocks (SwiGLU x4), 4 heads (hd 128), dropout 0.2, sampled_softmax "
                         f"N={n_neg}, V=1,000,000 items (the 5M x 512 catalog of the config is the SCORING run: `topk5m`)")
    if kind == "esasrec_kpm":   # the same model with use_key_padding_mask=True: the configuration under which the LiGR stack packs exactly
        spec = family_spec("esasrec", n_neg)
        d, H, nb, L, B = 512, 4, 2, 200, 128
        spec["model"] = SASRecModel(n_factors=d, n_blocks=nb, n_heads=H, session_max_len=L, dropout_rate=0.2, loss="sampled_softmax",
                                    n_negatives=n_neg, batch_size=B, lr=1e-3, epochs=1, seed=32, transformer_layers_type=hnn.LiGRLayers,
                                    use_key_padding_mask=True)
        spec["name"] += ", use_key_padding_mask=True (packed rows)"
        spec["desc"] += "; use_key_padding_mask=True: no real query sees a pad key, the stack runs on packed rows (72 % of B x L)"
        return spec
    raise SystemExit(f"unknown training workload {kind}")


def run_train(args, rank, world, kind="train"):
    from rectools_amd import ops

    spec = family_spec(kind, args.n_negatives)
    d, H, nb, L, B, n_neg = spec["d"], spec["H"], spec["nb"], spec["L"], spec["B"], spec["n_neg"]
    ds = spec["ds"]()
    model = spec["model"]
    t0 = time.perf_counter()
    model._build_model_from_dataset(ds)      # what fit() does before its first epoch (process dataset, build, xavier, broadcast)
    prep_s = time.perf_counter() - t0        # fit()'s own preparation; the dataset's construction is reported beside it
    V = model.data_preparator.item_id_map.size - model.data_preparator.n_item_extra_tokens
    loop = model.training_loop()
    model.lightning_model.train()
    loop.begin_epoch(0)
    state = {"loss": None}

    def step():
        state["loss"] = loop.step()
        state["seqs"].append(loop.sequences_done)

    state["seqs"] = []
    if world > 1:
        model.optimizer.exchange_events = []
    wall, ev_ms = timed_steps(step, args.steps, args.warmup, world)
    dp_report = None
    if world > 1:
        evs = model.optimizer.exchange_events[-args.steps:]
        model.optimizer.exchange_events = None
        ex_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if evs else 0.0
        dp_report = {"per_rank_ms_per_step": timed_steps.per_rank_ms, "per_rank_device_ms_per_step": timed_steps.per_rank_device_ms,
                     "exchange_ms_per_step_per_rank": [round(v, 4) for v in gather_floats(ex_ms, world)],
                     "exchange": "sharded (reduce-scatter + Adam on the slice + all-gather)" if model.optimizer._use_sharded(world)
                                 else ("all-reduce in two buckets (block weights from the backward pass's gradient hook, embeddings behind it)"
                                       if model.optimizer.early_stats["started"] > 0 else "all-reduce of the flat gradient"),
                     "early_bucket": dict(model.optimizer.early_stats, bytes=int((model.optimizer.flat_p.numel() - (model.optimizer.early_from or
                                                                                                          model.optimizer.flat_p.numel())) * 4)),
                     "gradient_bytes": int(model.optimizer.flat_p.numel() * 4)}
    # sequences of the timed steps as counted by the loop (an epoch's last batch is short: 17,312 sessions per rank at 8 ranks
    # = 135 full batches + 32 sessions); every rank holds an equally long shard, so the job total is this rank's count x ranks
    seqs = state["seqs"][args.warmup + args.steps - 1] - (state["seqs"][args.warmup - 1] if args.warmup > 0 else 0)
    value = seqs * world / wall
    # ---- roofline pass: 3 more steps with HIP events (on the launch stream) around every rt_* call.  By default the
    # weight-gradient products stay on their side stream, as in the timed region (durations then include the overlap, and
    # agree with rocprofv3 of this command); `single_stream_ms` repeats the pass with everything on one stream.
    def instrumented(single_stream):
        ops.start_timing(single_stream=single_stream)
        for _ in range(3):
            step()
        return ops.stop_timing()

    rec = instrumented(False)
    rec1 = instrumented(True)
    per_kernel = {k: (sum(t for t, _ in v) / 3.0, len(v) / 3.0) for k, v in rec.items()}
    per_kernel1 = {k: sum(t for t, _ in v) / 3.0 for k, v in rec1.items()}
    total_k = sum(t for t, _ in per_kernel.values())
    dom = max(per_kernel, key=lambda k: per_kernel[k][0])
    breakdown = {k: {"ms_per_step": round(t, 4), "calls_per_step": c, "single_stream_ms": round(per_kernel1.get(k, 0.0), 4)}
                 for k, (t, c) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])}
    roof = None
    if dom in ("rt_gemm", "rt_gemm_grouped"):
        # every launch of the GEMM kernel family: single products and grouped launches (tag = (sum of M N K, 1, 1))
        calls = rec.get("rt_gemm", []) + rec.get("rt_gemm_grouped", [])
        fl = sum(2.0 * m * n * k for _, (m, n, k) in calls)
        ms = sum(t for t, _ in calls)
        tf = fl / (ms * 1e-3) / 1e12
        ms1 = sum(t for t, _ in rec1.get("rt_gemm", []) + rec1.get("rt_gemm_grouped", []))
        roof = {"kernel": "GEMM family of the step: gemm_wp_kernel (forward / data-gradient products on pre-split weight planes, rt_gemm_wp) + "
                          "gemm_dma_kernel (weight gradients, grouped / split-K: rt_gemm, rt_gemm_grouped, rt_wgrad_grouped)", "bound": "mfma",
                "achieved": round(tf, 2), "peak": round(GEMM_PEAK_TF, 1), "unit": "TFLOP/s", "frac": round(tf / GEMM_PEAK_TF, 4),
                "arithmetic": ("fp32 via 6 bf16-MFMA products of an exact 3-way bf16 split, fp32 accumulate: peak = 2500 TF bf16 / 6; "
                               f"against the f32-input MFMA peak (157.3 TF) the same rate is {tf / MFMA_F32_PEAK_TF:.3f}")
                              if GEMM_X6 else "f32-input MFMA (v_mfma_f32_32x32x2_f32)",
                "traffic": load_traffic("train_gemm"), "avg_launch_ms": round(ms / len(calls), 4),
                "algorithmic_flops_per_launch": fl / len(calls), "launches_per_step": len(calls) / 3.0,
                "single_stream": {"avg_launch_ms": round(ms1 / len(calls), 4), "achieved": round(fl / (ms1 * 1e-3) / 1e12, 2)}}
    else:
        M = B * L
        attn = ("rt_mha_fwd", "rt_mha_bwd", "rt_hstu_attn_fwd", "rt_hstu_attn_bwd", "rt_mha_varlen_train_fwd", "rt_mha_varlen_bwd",
                "rt_mha_varlen_bidir_fwd", "rt_mha_varlen_bidir_bwd", "rt_hstu_attn_varlen_fwd", "rt_hstu_attn_varlen_bwd")
        if dom in attn:
            n2, what = float(L * L), "dense 4 L^2 d flops per sequence and block"
            if "varlen" in dom:      # packed sessions: the dense square of every session's OWN length (mean over the resident sessions)
                off = np.asarray(loop.store.offsets, dtype=np.int64)
                n = np.clip(off[1:] - off[:-1] - (0 if loop.bert else 1), 0, L).astype(np.float64)
                n2, what = float((n * n).mean()), "packed sessions: dense 4 n^2 d flops per session and block, n = its real length"
            fl = 4.0 * n2 * d * B * (2.5 if dom.endswith("bwd") else 1.0)       # dense; causal-useful half is executed
            ms = per_kernel[dom][0] / per_kernel[dom][1]
            tf = fl / (ms * 1e-3) / 1e12
            # the bf16-plane kernels (rt_attention_v3.hip / _v2.hip: head sizes 32 / 64 / 128 of the softmax families, 32 / 64 of packed HSTU) run six bf16 products per fp32
            # product; everything else is on the f32-input instruction
            x6 = (dom.startswith("rt_mha_varlen") and d // H in (32, 64, 128)) or \
                 (dom.startswith("rt_hstu_attn_varlen") and d // H in (32, 64) and os.environ.get("RT_HSTU_ATTN", "") != "ring")   # K6v2
            peak = MFMA_BF16_PEAK_TF / 6.0 if x6 else MFMA_F32_PEAK_TF
            roof = {"kernel": dom + f" ({what}; bwd x2.5)", "bound": "mfma", "achieved": round(tf, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                    "avg_launch_ms": round(ms, 4), "algorithmic_flops_per_launch": fl}
        elif dom.startswith("rt_sampled_loss"):
            byts = M * 0.72 * (1 + n_neg) * (4.0 * d + 8.0) * (2.0 if dom.endswith("bwd") else 1.0) + 4.0 * M * d
        elif dom.startswith("rt_adam"):
            byts = 5.0 * 4.0 * sum(p.numel() for p in model.torch_model.parameters())     # K13: 5 passes over the parameters
        else:
            byts = 8.0 * M * d
        if roof is None:
            ms = per_kernel[dom][0] / per_kernel[dom][1]
            gbs = byts / (ms * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": load_traffic("train_" + dom), "avg_launch_ms": round(ms, 4),
                    "algorithmic_bytes_per_launch": byts}
    roof["kernel_ms_per_step"] = round(total_k, 3)
    # wall time of the issuing loop: follows the GPU whenever the launch queue pushes back, so it is an UPPER bound of the host's own
    # cost; `host_only_ms_per_step` (every launch elided, scripts/microbench/dry_launch.cpp) is the host's cost proper
    roof["host_issue_ms_per_step"] = round(getattr(timed_steps, "host_issue_ms", 0.0), 4)
    roof["allocator_in_timed_steps"] = getattr(timed_steps, "device_allocs", None)
    # who issued the timed steps: the library (`rt_sasrec_step_run`: the stock packed SASRec step as one compiled call) or the autograd
    # nodes of rectools_amd/ops.py (every other configuration, data-parallel runs, RT_NATIVE_STEP=0); the per-kernel pass above always
    # runs through autograd (its event pairs bracket the Python-issued calls), same entry points in the same order
    roof["step_issue"] = "rt_sasrec_step_run" if getattr(getattr(loop, "_native", None), "steps", 0) > 0 else "autograd"
    roof["device_ms_per_step"] = round(ev_ms, 4)      # HIP events around each timed step on the launch stream
    roof["step_flops_dense"] = 3.0 * B * (nb * spec["blk"] + spec["loss"])     # fwd + bwd = 3 x fwd, on the PADDED [B, L] window
    # what the step EXECUTES: a packed loop runs the real rows only (mean over the epoch's batches; the padded window's flops would
    # overstate the rate by the padding share — VERDICT r3 weak #2 (iv)); the per-row GEMM / loss flops scale with the rows, the
    # attention square with every session's own length
    exe = roof["step_flops_dense"]
    if getattr(loop, "packed", False) and getattr(loop, "_cu_host", None) is not None:
        cu = np.asarray(loop._cu_host)[:, :B + 1].astype(np.float64)
        rows = float(cu[:, B].mean())
        n2 = float((np.diff(cu, axis=1) ** 2).sum(axis=1).mean())
        per_row = (spec["blk"] - 4.0 * L * L * d) / L      # GEMM flops of one row and block (fwd)
        exe = 3.0 * (nb * (per_row * rows + 4.0 * n2 * d) + spec["loss"] / L * rows)
        roof["rows_per_step_executed"] = round(rows, 1)
    roof["step_flops_executed"] = exe
    roof["step_TFLOPs"] = round(exe / (wall / args.steps) / 1e12, 2)              # executed flops / measured step time
    roof["step_TFLOPs_padded_window_equivalent"] = round(roof["step_flops_dense"] / (wall / args.steps) / 1e12, 2)
    info = dict(model=model, ds=ds, loop=loop, V=V, d=d, H=H, nb=nb, L=L, B=B, n_neg=n_neg, breakdown=breakdown, spec=spec, dp_report=dp_report,
                loss=float(state["loss"].detach()), prep_s=prep_s, steps_per_epoch=loop.batches_left() + loop.pos // B)
    return value, wall, roof, info


def _dry_shim_path() -> str:
    """Build (once) the launch-eliding measurement shim next to its source; -> path of the shared object."""
    import subprocess

    src = os.path.join(ROOT, "scripts", "microbench", "dry_launch.cpp")
    out = os.path.join(ROOT, "scripts", "microbench", "_bin", "libdry_launch.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", src, "-o", out, "-ldl"])
    return out


def host_only_leg(kind: str, n_neg: int, steps: int = 60, warmup: int = 10):
    """The host's OWN cost of issuing a training step: the same product loop in a child process under LD_PRELOAD of the shim that
    turns every kernel launch and async memset (this package's and torch's) into a counted no-op for the timed steps — the GPU has
    nothing to do, nothing can push back, what remains is Python + autograd + ctypes + the HIP runtime's host side."""
    import subprocess

    try:
        shim = _dry_shim_path()
        env = dict(os.environ, LD_PRELOAD=shim, RT_DRY_SHIM=shim)
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", kind, "--host-only-child", "--steps", str(steps), "--warmup", str(warmup),
               "--n-negatives", str(n_neg)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if res.returncode != 0 or not line:
            return {"error": (res.stderr or res.stdout)[-400:]}
        return json.loads(line[-1])
    except Exception as e:   # the measurement must never take the line down
        return {"error": repr(e)[:400]}


def host_only_child(args):
    """Child side of `host_only_leg` (runs under LD_PRELOAD of the shim)."""
    import ctypes

    shim = ctypes.CDLL(os.environ["RT_DRY_SHIM"])
    shim.rt_dry_count.restype = ctypes.c_longlong
    spec = family_spec(args.workload if args.workload != "auto" else "train", args.n_negatives)
    model = spec["model"]
    model._build_model_from_dataset(spec["ds"]())
    loop = model.training_loop()
    model.lightning_model.train()
    loop.begin_epoch(0)
    for _ in range(args.warmup):          # real steps: allocator, lazy initialisations, autotuned paths
        loop.step()
    torch.cuda.synchronize()
    assert shim.rt_dry_selftest() == 1, "the shim cannot reach the HIP runtime"
    shim.rt_dry_set(1)
    for _ in range(3):
        loop.step()
    shim.rt_dry_set(1)                    # resets the counters
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step()
    el = time.perf_counter() - t0
    n_l, n_m = shim.rt_dry_count(0), shim.rt_dry_count(1)
    shim.rt_dry_set(0)
    torch.cuda.synchronize()
    print(json.dumps({"host_only_ms_per_step": round(el / args.steps * 1e3, 4), "launches_per_step": round(n_l / args.steps, 1),
                      "memsets_per_step": round(n_m / args.steps, 1), "steps": args.steps,
                      "what": "wall time of loop.step() with every kernel launch / async memset elided by an LD_PRELOAD shim (GPU idle)"}))


def cpu_baseline_train(info, budget_s=25.0):
    """-> (seqs/s, kind, sample).  kind "reference": the UNMODIFIED reference (oracle/cpu_reference.py over /root/reference or the staged
    oracle/_ref): its own SASRecModel at the bench's configuration — B=128, L=200, d=256, 2 blocks, sampled_softmax N — on 4,096
    ML-20M-shaped users of the same 26,744-item catalog (a step's cost does not depend on the number of users), its own DataLoader,
    training_step + backward + Adam.step.  Fallback "port" (no reference tree on this box): the oracle on 32-sequence sub-batches."""
    from oracle import cpu_reference

    if cpu_reference.available():
        import pandas as pd

        from rectools_amd import synth

        shape = synth.ML_20M
        u, it, ts = synth.gen_interactions(4096, shape["n_items"], mean_len=shape["mean_len"], min_len=shape["min_len"], max_len=shape["max_len"],
                                           seed=0)
        df = pd.DataFrame({"user_id": u, "item_id": it, "weight": 1.0, "datetime": pd.to_datetime(ts, unit="s")})
        kw = dict(n_factors=info["d"], n_blocks=info["nb"], n_heads=info["H"], session_max_len=info["L"], dropout_rate=0.2,
                  loss="sampled_softmax", n_negatives=info["n_neg"], batch_size=info["B"], lr=1e-3, epochs=1)
        rate, n, b, what = cpu_reference.train_step_rate(df, kw, budget_s=budget_s)
        return rate, "reference", what
    return cpu_baseline_train_port(info, budget_s) + ()


def cpu_baseline_train_port(info, budget_s=25.0):
    """The oracle (plain torch fp32 restatement of the reference step, CPU) on the same model and on batches cut by the same product
    collate + sampler: forward + backward + dense Adam, timed on the host cores on a bounded sample (32-sequence sub-batches)."""
    from oracle import transformer_oracle as T

    model, loop = info["model"], info["loop"]
    cfg = dict(V=info["V"], B=info["B"], L=info["L"], d=info["d"], H=info["H"], n_blocks=info["nb"], N=info["n_neg"],
               loss="sampled_softmax", dist="dot", logits_t=1.0, causal=True, keypad=False, layers="sasrec", n_extra=1,
               gbce_t=0.2, lr=1e-3)
    params = {k: v.detach().cpu().clone() for k, v in model.torch_model.state_dict().items()}
    adam = T.AdamState(lr=1e-3)
    bsub = 32  # bounded sample: 32 sequences per CPU step
    dp = model.data_preparator
    batches = []
    for i in range(2):
        idx = loop.mine_t[i * bsub:(i + 1) * bsub]
        batches.append({k: v.cpu() for k, v in dp.add_negatives(dp.collate_train_device(loop.dstore, idx)).items()})
    t0 = time.perf_counter()
    n = 0
    while True:
        _, grads = T.loss_and_grads(cfg, params, batches[n % len(batches)])
        params = adam.step(params, grads)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 6:
            break
    return bsub * n / el, "port", (f"oracle/transformer_oracle (torch CPU fp32 restatement; no reference tree on this box) "
                                   f"fwd+bwd+Adam, {n} steps of {bsub} sequences (sub-batches of the GPU run's B=128)")


def run_recommend_e2e(info, n_users=None):
    """`model.recommend()` through the public API: id mapping, device glue (sessions, viewed CSR), session encoding, exact top-k, result
    frame.  `value` = users / wall clock of ONE whole call (it ends with the D2H of the results); `phases_ms` from a second, instrumented
    call (a device synchronisation between the phases: their sum is a little above the un-instrumented call)."""
    model, ds = info["model"], info["ds"]
    users = np.asarray(ds.user_id_map.external_ids)          # ALL users (138,493 at the ML-20M shape), full catalog
    if n_users is not None:
        users = users[:n_users]
    model.is_fitted = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.recommend(users=users[:2048], dataset=ds, k=10, filter_viewed=True)   # first call on this Dataset (also the warm-up)
    first = time.perf_counter() - t0
    model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)          # allocator warm-up at the full request size
    times, rows = [], 0
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reco = model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)
        times.append(time.perf_counter() - t0)
        rows = int(len(reco))
    med = float(np.median(times))
    model.phase_log = {}
    model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)
    phases = {k: round(v * 1e3, 3) for k, v in model.phase_log.items()}
    model.phase_log = None
    return {"value": round(len(users) / med, 1), "unit": "users/s", "users": int(len(users)), "seconds": round(med, 5), "rows": rows,
            "seconds_of_each_call": [round(t, 5) for t in times], "best_users_per_s": round(len(users) / min(times), 1),
            "phases_ms": phases, "first_call_seconds_2048_users": round(first, 4),
            "what": "SASRecModel.recommend(ALL users, dataset, k=10, filter_viewed=True), public API, MEDIAN of 5 whole calls.  phases_ms (one "
                    "instrumented call): glue = id mapping + session rows + H2D; encoder = packed sessions -> user embeddings; ranker = "
                    "viewed-items CSR rows + rt_topk_score; frame = D2H + pandas frame.  The Dataset's session store and viewed-items CSR "
                    "are built on the device by the FIRST recommend() on it (`first_call_seconds_2048_users`, 19.8 M rows) and reused"}


def family_recommend(info, kind: str, n_users: int = 8192):
    """recommend() of another model family through the public API: users / wall clock of one whole call (best of 2 after a warm-up).
    HSTU needs the request's context (one row per user: a time behind every history)."""
    import pandas as pd

    model, ds = info["model"], info["ds"]
    users = np.asarray(ds.user_id_map.external_ids)[:n_users]
    model.is_fitted = True
    kw = {}
    if kind == "hstu":
        last = pd.to_datetime(ds.interactions.df["datetime"]).max()
        kw["context"] = pd.DataFrame({"user_id": users, "datetime": last + pd.Timedelta(days=1)})
    model.recommend(users=users[:1024], dataset=ds, k=10, filter_viewed=True, **kw)
    model.recommend(users=users, dataset=ds, k=10, filter_viewed=True, **kw)
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reco = model.recommend(users=users, dataset=ds, k=10, filter_viewed=True, **kw)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    return {"value": round(len(users) / best, 1), "unit": "users/s", "users": int(len(users)), "seconds": round(best, 5), "rows": int(len(reco)),
            "what": f"{type(model).__name__}.recommend(users, dataset, k=10, filter_viewed=True" + (", context" if kw else "") +
                    "), public API, device path (packed encoder, two-stage exact top-k), best of 2 whole calls"}


def load_traffic(name: str):
    """PMC-measured HBM bytes per launch (profiles/traffic.json, written from rocprofv3 --pmc passes)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(name)
        except Exception:
            return None
    return None


def topk_leg(kind, args, rank, world, cpu_baseline):
    """-> sub-record of one top-k leg: `recommend` (C2 catalog, 16,384 users/step, viewed filter) or `topk5m` (C5)."""
    from rectools_amd import synth

    if kind == "recommend":
        V, d = synth.ML_20M["n_items"], 256
        ups = args.users_per_step or 16384
        upp = args.users_per_pass   # 0 = the library's choice
        steps, warmup = args.rec_steps, 3
        metric = "recommend() users/sec @k=10 (SASRec d=256, ML-20M-shaped catalog, filter_viewed)"
        workload = f"recommend top-k: 26744 items x d256 fp32, {ups} users/step, k=10, viewed-filter CSR"
        name, with_filter = "recommend_ml20m", os.environ.get("RT_BENCH_NO_FILTER") != "1"   # (diagnostic: cost of the viewed-items test)
    else:
        V, d = 5_000_000, 512
        ups = args.users_per_step or 16   # 16 users/launch: the HBM-bound regime (AI = B/2 flop/B; 32 users is the fp32 ridge)
        upp = args.users_per_pass or (16 if ups <= 16 else 32 if ups <= 32 else 64 if ups < 128 else 0)
        steps, warmup = args.topk_steps, (2 if ups > 64 else 5)
        metric = "full-catalog top-k users/sec @k=10 (5M x 512 fp32 catalog)"
        workload = f"top-k scoring: 5,000,000 items x d512 fp32 (10.24 GB), {ups} users/step, k=10, viewed-filter CSR"
        # the metric is recommend(filter_viewed=True): the C5 launches carry a viewed-items CSR too (ML-20M-shaped histories over the 5 M
        # items; the filter is tested per surviving candidate: its cost is in the figure)
        name, with_filter = ("topk5m" if ups <= 64 else f"topk5m_u{ups}"), os.environ.get("RT_BENCH_NO_FILTER") != "1"
    value, wall, roof, info = run_topk(steps, warmup, rank, world, V, d, ups, upp, with_filter, name)
    rec = {"metric": metric, "value": round(value, 2), "unit": "users/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(wall / steps * 1e3, 4), "dtype": "fp32",
           "config": {"workload": workload, "users_per_step": ups, "users_per_register_tile": upp or "library default", "parallelism": f"dp{world}"},
           "roofline": roof, "cpu_baseline": None}
    if kind != "recommend" and rank == 0:
        rec["roofline"].update(topk_extras(info, ups, upp))
    if cpu_baseline:
        torch.cuda.synchronize()
        small = info["n_items"] <= 100_000
        filt = info["filt"] if small or info["filt"] is None else info["filt"][:, :200_000]
        v, n, kind, what = cpu_baseline_topk(info["items"] if small else info["items"][:200_000], info["users_t"], filt)
        scale = 1.0 if small else 200_000 / info["n_items"]
        rec["cpu_baseline"] = {"value": round(v * scale, 2), "unit": "users/s", "cores": torch.get_num_threads(), "kind": kind,
                               "sample": what + ("" if small else f"; first 200k catalog rows, rate scaled by {scale:.3f}")}
        quiesce_host()
    del info
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps of the headline leg (default 200 train steps)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="auto", choices=["auto", "train", "recommend", "topk5m", "bert4rec", "hstu", "esasrec", "esasrec_kpm"])
    ap.add_argument("--users-per-pass", type=int, default=0, help="register tile: 16/32/64/128 users (0 = auto)")
    ap.add_argument("--users-per-step", type=int, default=0)
    ap.add_argument("--n-negatives", type=int, default=128, help="sampled_softmax negatives (tutorial setting 128)")
    ap.add_argument("--rec-steps", type=int, default=20, help="timed steps of the recommend sub-leg")
    ap.add_argument("--topk-steps", type=int, default=40, help="timed steps of the topk5m sub-leg (a step is ~1.2 ms: enough of them that one host hiccup does not set the figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true", help="skip the BERT4Rec / HSTU / eSASRec sub-records of the auto run")
    ap.add_argument("--host-only-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-host-only", action="store_true", help="skip the host-only (launch-elided) child run")
    args = ap.parse_args()
    rank, world, local, dist_info = dist_setup(args.gpus)
    if os.environ.get("RT_BENCH_DRY_RUN") == "1":   # launcher / rendezvous / line shape only (CPU test of the N > 1 start-up)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "dist": dist_info}))
        if world > 1:
            import torch.distributed as dist

            dist.destroy_process_group()
        return

    from rectools_amd import _lib

    _lib.load()  # fail loudly if the HIP extension is missing
    workload = args.workload
    if args.host_only_child:
        args.steps, args.warmup = args.steps or 60, args.warmup if args.warmup is not None else 10
        host_only_child(args)
        return
    cpu_ok = rank == 0 and world == 1 and not args.no_cpu_baseline
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("RT_")}   # every engine knob that is set

    if workload in ("recommend", "topk5m"):   # one top-k leg alone: its record is the line
        if args.steps is not None:
            args.rec_steps = args.topk_steps = args.steps
        out = topk_leg(workload, args, rank, world, cpu_ok)
        if args.warmup is not None:
            out["warmup_requested"] = args.warmup
        out.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                    "env": env})
    else:
        if args.steps is None:
            args.steps = 200 if workload in ("auto", "train") else 40     # 200 timed steps after 20 warm-up
        if args.warmup is None:
            args.warmup = 20 if workload in ("auto", "train") else 5
        kind = "train" if workload == "auto" else workload
        value, wall, roof, info = run_train(args, rank, world, kind)
        spec = info["spec"]
        dp_train = info.get("dp_report")
        out = {
            "metric": f"train seqs/sec ({spec['name']})"
                      + (" [+ recommend() users/sec@k=10 and 5Mx512 top-k in the sub-records]" if workload == "auto" else ""),
            "value": round(value, 2), "unit": "seqs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32", "data": "synthetic",
            "config": {"workload": f"SASRecModel.fit() steady-state step (models._TrainLoop.step): device collate from the HBM session "
                                   f"store + on-device negatives + fwd + bwd + Adam" + (" + RCCL all-reduce" if world > 1 else "")
                                   + f"; {spec['desc']}, V={info['V']} items, {info['steps_per_epoch']} steps/epoch",
                       "global_batch": info["B"] * world, "seq_len": info["L"], "parallelism": f"dp{world}", "n_negatives": info["n_neg"],
                       "fit_prep_s": round(info["prep_s"], 2), **getattr(make_dataset, "last_times", {}),
                       "gemm_arithmetic": "fp32 in / fp32 out; products as 6 bf16-MFMA terms of an exact 3-way bf16 split, fp32 "
                                          "accumulate (fp32-accurate; RT_GEMM_SPLIT=exact = f32-input MFMA)" if GEMM_X6
                                          else "f32-input MFMA (exact fp32)"},
            "roofline": roof, "cpu_baseline": None,
            "kernel_breakdown": info["breakdown"], "final_loss": round(info["loss"], 5),
        }
        if rank == 0 and world == 1 and not args.no_host_only and kind == "train":
            ho = host_only_leg(kind, args.n_negatives)
            roof["host_only"] = ho
            if "host_only_ms_per_step" in ho:
                roof["host_only_ms_per_step"] = ho["host_only_ms_per_step"]
                roof["launches_per_step_all_kernels"] = ho["launches_per_step"]
        if kind in ("bert4rec", "hstu", "esasrec") and world == 1 and rank == 0:      # a family leg run alone carries its recommend() figure too
            out["recommend"] = family_recommend(info, kind)
        if cpu_ok and kind == "train":
            v, kind_b, what = cpu_baseline_train(info)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "seqs/s", "cores": torch.get_num_threads(), "kind": kind_b, "sample": what}
            quiesce_host()
        if workload == "auto":
            # the same loop with the GEMMs on the f32-input MFMA (RT_GEMM_SPLIT=exact is read per call): the number to compare when the
            # bf16x6 arithmetic of the default line is questioned ("carry it as a sub-record of the driver line")
            os.environ["RT_GEMM_SPLIT"] = "exact"
            try:
                st = {"seqs": []}

                def step_exact():
                    info["loop"].step()
                    st["seqs"].append(info["loop"].sequences_done)

                wall_x, _ = timed_steps(step_exact, 40, 5, world)
                seqs_x = st["seqs"][-1] - st["seqs"][4]
                out["train_exact_gemm"] = {"value": round(seqs_x * world / wall_x, 2), "unit": "seqs/s", "steps": 40, "warmup": 5,
                                           "ms_per_step": round(wall_x / 40 * 1e3, 4),
                                           "what": "the train leg with RT_GEMM_SPLIT=exact: every GEMM on v_mfma_f32_32x32x2_f32 (the "
                                                   "packed attention keeps its bf16 planes)"}
            finally:
                os.environ.pop("RT_GEMM_SPLIT", None)
            e2e = run_recommend_e2e(info) if world == 1 else None
            del info
            torch.cuda.empty_cache()
            kernel_leg = topk_leg("recommend", args, rank, world, cpu_ok)
            if e2e is not None:
                # BASELINE's second metric is recommend() users/sec through the API: that is the record's value; the ranker kernel
                # (the dominant kernel of the call) supplies its roofline and keeps its own rate as `ranker_kernel`
                rec = {"metric": kernel_leg["metric"], "value": e2e["value"], "unit": "users/s", "users": e2e["users"],
                       "seconds": e2e["seconds"], "rows": e2e["rows"], "phases_ms": e2e["phases_ms"],
                       "first_call_seconds_2048_users": e2e["first_call_seconds_2048_users"], "dtype": "fp32", "config": {"workload": e2e["what"]},
                       "roofline": kernel_leg["roofline"], "cpu_baseline": kernel_leg["cpu_baseline"],
                       "ranker_kernel": {k: kernel_leg[k] for k in ("value", "unit", "steps", "ms_per_step", "config")}}
                out["recommend"] = rec
            else:
                out["recommend"] = kernel_leg
            out["topk5m"] = topk_leg("topk5m", args, rank, world, cpu_ok)
            big = argparse.Namespace(**vars(args))
            big.users_per_step, big.users_per_pass, big.topk_steps = 4096, 0, 2     # users per pass: the library's choice
            out["topk5m_u4096"] = topk_leg("topk5m", big, rank, world, cpu_ok)   # >= 4096 users over 5M x 512 (the MFMA regime)
            if not args.no_families:
                # the other BASELINE configs' model families at their stated shapes (configs[2..4]: BERT4Rec d256 L200 full softmax; HSTU d256
                # L512 relative time + position bias, 1 M items; eSASRec = SASRec on LiGR blocks, d512, 1 M items): the same product loop, 20
                # timed steps each — short on purpose (the line must finish within minutes); `--workload <family>` runs one at length
                fam = argparse.Namespace(**vars(args))
                fam.steps, fam.warmup = 20, 6
                out["families"] = {}
                for kind_f in ("bert4rec", "hstu", "esasrec", "esasrec_kpm"):
                    v_f, wall_f, roof_f, info_f = run_train(fam, rank, world, kind_f)
                    out["families"][kind_f] = {"metric": f"train seqs/sec ({info_f['spec']['name']})", "value": round(v_f, 2), "unit": "seqs/s",
                                                "steps": fam.steps, "warmup": fam.warmup, "ms_per_step": round(wall_f / fam.steps * 1e3, 4),
                                                "config": {"workload": info_f["spec"]["desc"], "global_batch": info_f["B"] * world},
                                                "roofline": {k: roof_f[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "kernel_ms_per_step",
                                                                                   "host_issue_ms_per_step", "device_ms_per_step", "allocator_in_timed_steps") if k in roof_f},
                                                "final_loss": round(info_f["loss"], 5),
                                                "kernel_breakdown": {k: v["ms_per_step"] for k, v in list(info_f["breakdown"].items())[:8]}}
                    if kind_f in ("bert4rec", "hstu", "esasrec") and world == 1:      # the families whose recommend() takes the packed device path
                        out["families"][kind_f]["recommend"] = family_recommend(info_f, kind_f)
                    del info_f
                    torch.cuda.empty_cache()
            if not args.no_families and rank == 0 and world == 1:
                # BASELINE configs[3] AT its stated size — 10 M users x 1 M items (393 M interactions) — when the host can hold it (62 GB of
                # resident memory at the peak, measured: profiles/r5_c4_scale_10m.json), else at 2 M users: the host path of
                # HSTUModel.fit() (Dataset.construct, process_dataset_train, store upload, epoch bookkeeping) + 20 product steps, in a child
                # process (its memory goes back to the system); `families.hstu` above is the model-shape leg (65,536 users, long histories)
                import subprocess

                big = False
                try:
                    import psutil

                    big = psutil.virtual_memory().available >= 160 * 2 ** 30 and os.environ.get("RT_BENCH_C4_USERS", "") != "2000000"
                except Exception:
                    pass
                n_c4 = 10_000_000 if big else 2_000_000
                key = "hstu_10m_users" if big else "hstu_2m_users"
                try:
                    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "c4_scale.py"), str(n_c4), "40", "20"],
                                         capture_output=True, text=True, timeout=600)
                    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
                    out["families"][key] = json.loads(line[-1]) if line else {"error": (res.stderr or res.stdout)[-300:]}
                except Exception as e:      # never take the line down
                    out["families"][key] = {"error": repr(e)[:300]}
        out["env"] = env
        if dp_train is not None:
            dist_info["train"] = dp_train
    out["dist"] = dist_info

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
