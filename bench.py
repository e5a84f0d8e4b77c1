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

        # RT_BENCH_BACKEND=gloo lets the N>1 path run on a box with fewer GPUs than ranks (scripts/gpu_full.sh does that
        # with 2 ranks on the one GPU of the test box); the driver's multi-GPU runs use nccl = RCCL, one rank per GPU.
        backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
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


# ---------------------------------------------------------------------------------------------------
# training workload (BASELINE.json configs[1]: SASRec d=256, 2 blocks, L=200, sampled_softmax, ML-20M-shaped)
# ---------------------------------------------------------------------------------------------------
def make_sasrec(V, d, H, n_blocks, L, dropout, loss, n_neg, device="cuda"):
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn

    n_tokens = V + 1
    torch.manual_seed(31)  # before construction: biases / LayerNorm / embeddings take their default init from this stream
    item_model = hnn.SumOfEmbeddingsConstructor(n_tokens, [hnn.IdEmbeddingsItemNet(d, n_tokens, 0.0)])
    pos = hnn.LearnableInversePositionalEncoding(True, L, d)
    layers = hnn.SASRecTransformerLayers(n_blocks, d, H, dropout)
    bb = hnn.TransformerTorchBackbone(H, dropout, item_model, pos, layers, hnn.DistanceSimilarityModule("dot"), True, False)
    lm = hl.TransformerLossModule(bb, loss, n_neg, 0.2, 1.0, 1).to(device)
    torch.manual_seed(32)
    hl.xavier_normal_init(lm.torch_model)
    return lm


def make_train_batches(n_batches, B, L, V, n_neg, rank, seed=0):
    """SASRec training batches (x, y, yw, negatives) from ML-20M-shaped synthetic histories, collated exactly as
    SASRecDataPreparator._collate_fn_train does (sasrec.py:86-104): last L+1 items, left padding, shift by one."""
    from rectools_amd import synth

    n_users = n_batches * B
    u, it, _ = synth.gen_interactions(n_users, V, mean_len=144.0, min_len=20, max_len=9254, seed=seed + 17 * rank)
    it = it + 1  # internal ids: 0 is PAD
    bounds = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=n_users))])
    x = np.zeros((n_users, L), np.int64)
    y = np.zeros((n_users, L), np.int64)
    for i in range(n_users):
        ses = it[bounds[i]:bounds[i + 1]][-(L + 1):]
        x[i, L - (len(ses) - 1):] = ses[:-1]
        y[i, L - (len(ses) - 1):] = ses[1:]
    rng = np.random.default_rng(seed + 5 + rank)
    out = []
    for b in range(n_batches):
        sl = slice(b * B, (b + 1) * B)
        yb = torch.from_numpy(y[sl])
        batch = {"x": torch.from_numpy(x[sl]).cuda(), "y": yb.cuda(), "yw": (yb != 0).float().cuda()}
        if n_neg:
            batch["negatives"] = torch.from_numpy(rng.integers(1, V + 1, size=(B, L, n_neg))).cuda()
        out.append(batch)
    return out


def sasrec_step_flops(B, L, d, n_blocks, n_neg):
    """Dense algorithmic flops of one training step (fwd + bwd = 3x fwd): SASRec block 12 L d^2 + 4 L^2 d."""
    blk = 12.0 * L * d * d + 4.0 * L * L * d
    loss = 2.0 * L * (1 + n_neg) * d
    return 3.0 * B * (n_blocks * blk + loss)


def run_train(args, rank, world):
    from rectools_amd import lightning as hl
    from rectools_amd import ops, synth

    V, d, H, nb, L, B = synth.ML_20M["n_items"], 256, 4, 2, 200, 128
    n_neg = args.n_negatives
    lm = make_sasrec(V, d, H, nb, L, 0.2, "sampled_softmax", n_neg)
    lm.train()
    opt = hl.FlatAdam(lm.torch_model, lr=1e-3)
    opt.broadcast_parameters()  # N > 1: replicas start from rank 0's weights, as DDP does
    n_batches = min(args.steps + args.warmup, 24)
    batches = make_train_batches(n_batches, B, L, V, n_neg, rank)
    state = {"i": 0, "loss": None}

    def step():
        batch = batches[state["i"] % n_batches]
        state["i"] += 1
        ops.RNG.next_step()
        opt.zero_grad()
        loss = lm.training_loss(batch)
        loss.backward()
        opt.step(world)
        state["loss"] = loss

    wall, ev_ms = timed_steps(step, args.steps, args.warmup, world)
    value = B * args.steps * world / wall
    # ---- roofline pass: 3 more steps with HIP events around every rt_* launch (on the launch stream) ----
    ops.start_timing()
    for _ in range(3):
        step()
    rec = ops.stop_timing()
    per_kernel = {k: (sum(t for t, _ in v) / 3.0, len(v) / 3.0) for k, v in rec.items()}
    total_k = sum(t for t, _ in per_kernel.values())
    dom = max(per_kernel, key=lambda k: per_kernel[k][0])
    breakdown = {k: {"ms_per_step": round(t, 4), "calls_per_step": c} for k, (t, c) in
                 sorted(per_kernel.items(), key=lambda kv: -kv[1][0])}
    roof = None
    if dom == "rt_gemm":
        calls = rec["rt_gemm"]
        fl = sum(2.0 * m * n * k for _, (m, n, k) in calls)
        ms = sum(t for t, _ in calls)
        tf = fl / (ms * 1e-3) / 1e12
        roof = {"kernel": "gemm_dma_kernel (rt_gemm: all forward/dgrad/wgrad products of the step)", "bound": "mfma",
                "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                "traffic": load_traffic("train_gemm"), "avg_launch_ms": round(ms / len(calls), 4),
                "algorithmic_flops_per_launch": fl / len(calls), "launches_per_step": len(calls) / 3.0}
    else:
        M = B * L
        if dom in ("rt_sampled_loss_fwd", "rt_sampled_loss_bwd"):
            byts = M * 0.72 * (1 + n_neg) * (4.0 * d + 8.0) * (2.0 if dom.endswith("bwd") else 1.0) + 4.0 * M * d
        else:
            byts = 8.0 * M * d
        ms = per_kernel[dom][0] / per_kernel[dom][1]
        gbs = byts / (ms * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": load_traffic("train_" + dom), "avg_launch_ms": round(ms, 4),
                "algorithmic_bytes_per_launch": byts}
    roof["kernel_ms_per_step"] = round(total_k, 3)
    roof["step_flops_dense"] = sasrec_step_flops(B, L, d, nb, n_neg)
    info = dict(lm=lm, batches=batches, V=V, d=d, H=H, nb=nb, L=L, B=B, n_neg=n_neg, breakdown=breakdown,
                loss=float(state["loss"]))
    return value, wall, roof, info


def cpu_baseline_train(info, budget_s=25.0):
    """The oracle (plain torch fp32 restatement of the reference step, CPU) on the same model/batches:
    forward + backward + dense Adam, timed on the host cores."""
    from oracle import transformer_oracle as T

    cfg = dict(V=info["V"], B=info["B"], L=info["L"], d=info["d"], H=info["H"], n_blocks=info["nb"], N=info["n_neg"],
               loss="sampled_softmax", dist="dot", logits_t=1.0, causal=True, keypad=False, layers="sasrec", n_extra=1,
               gbce_t=0.2, lr=1e-3)
    params = {k: v.detach().cpu().clone() for k, v in info["lm"].torch_model.state_dict().items()}
    adam = T.AdamState(lr=1e-3)
    bsub = 32  # bounded sample: 32 sequences per CPU step
    t0 = time.perf_counter()
    n = 0
    while True:
        b = {k: v[:bsub].cpu() for k, v in info["batches"][n % len(info["batches"])].items()}
        _, grads = T.loss_and_grads(cfg, params, b)
        params = adam.step(params, grads)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 6:
            break
    return bsub * n / el, f"oracle/transformer_oracle (torch CPU fp32) fwd+bwd+Adam, {n} steps of {bsub} sequences"


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
    ap.add_argument("--n-negatives", type=int, default=128, help="sampled_softmax negatives (tutorial setting 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    rank, world, local = dist_setup(args.gpus)

    from rectools_amd import _lib, synth

    _lib.load()  # fail loudly if the HIP extension is missing
    workload = args.workload
    if workload == "auto":
        workload = "train"

    extra = {}
    cpu = None
    if workload == "train":
        if args.steps is None:
            args.steps = 30
        if args.warmup is None:
            args.warmup = 5
        value, wall, roof, info = run_train(args, rank, world)
        metric, unit = "train seqs/sec (SASRec d=256 n_blocks=2 L=200 sampled_softmax, ML-20M-shaped)", "seqs/s"
        config = {"workload": f"SASRec fit() step: B=128/GPU x L=200, d=256, 2 blocks, 4 heads, dropout 0.2, sampled_softmax "
                              f"N={args.n_negatives}, V=26744 items, fwd+bwd+Adam" + (" + RCCL all-reduce" if world > 1 else ""),
                  "global_batch": 128 * world, "seq_len": 200, "parallelism": f"dp{world}", "n_negatives": args.n_negatives}
        extra["kernel_breakdown"] = info["breakdown"]
        extra["final_loss"] = round(info["loss"], 5)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            v, what = cpu_baseline_train(info)
            cpu = {"value": round(v, 2), "unit": unit, "cores": torch.get_num_threads(), "kind": "port", "sample": what}
    elif workload == "recommend":
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

    if workload != "train" and rank == 0 and world == 1 and not args.no_cpu_baseline:
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
