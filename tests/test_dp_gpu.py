"""Data-parallel training step on the GPU path with world_size 2: two processes share the one GPU of the test box and
exchange gradients through gloo (RCCL refuses two ranks on one device; the collective is not what is under test).
Checks the product code of `FlatAdam.step(world)`: side-stream join -> gradient packing -> ONE all-reduce -> flat Adam
kernel with the 1/world scale, i.e. every replica lands on Adam(mean of the per-rank gradients) and stays in sync."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _wire_early_bucket(lm, opt, world, force=False):
    """What `models._TrainLoop` does for a data-parallel loop: the block weights' exchange starts from the gradient hook of the blocks' input."""
    tm = lm.torch_model
    opt.set_early_bucket([*tm.item_model.parameters(), *tm.pos_encoding_layer.parameters()])
    assert opt.early_first is not None and opt.early_from > 0
    log = []

    def fire():
        late = [p.grad is not None for p in opt.params[:opt.early_first] if p.ndim == 2 and p.shape[0] < 1024]      # (positions: no flat-buffer home)
        early = [p.grad is not None for p in opt.params[opt.early_first:]]
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        assert opt.begin_early_exchange(world, force=force)
        log.append((early, late, ev))

    tm.on_input_gradient = fire
    return log


def _check_early_log(log, opt, steps):
    """Every step took the two-bucket path; at the hook every block gradient existed and the lookup's had not been produced; the
    lookup's backward kernels ran on the device AFTER the point the exchange was issued behind."""
    assert opt.early_stats == {"started": steps, "redone": 0} and len(log) == steps, (opt.early_stats, len(log))
    for early, late, _ in log:
        assert all(early) and not any(late), (early, late)


def _worker(rank, world, port, out_dir, exchange="allreduce", shape="small"):
    early = exchange == "early"
    exchange = "allreduce" if early else exchange
    os.environ["RT_DP_EXCHANGE"] = exchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers_dp as bench
    from rectools_amd import lightning as hl
    from rectools_amd import ops

    # "c2": the BASELINE configs[1] model (26,744 items, d 256, 2 blocks, L 200, N 128; 31 MB of gradient), 8 sequences per rank
    V, d, H, nb, L, B, n_neg = (300, 64, 2, 2, 24, 16, 8) if shape == "small" else (26_744, 256, 4, 2, 200, 8, 128)
    lm = bench.make_sasrec(V, d, H, nb, L, 0.1, "sampled_softmax", n_neg)       # same seed: identical replicas
    lm.train()
    if rank == 1:   # replicas that drifted apart before the start must be pulled back by the broadcast
        with torch.no_grad():
            for prm in lm.torch_model.parameters():
                if prm.ndim == 1:
                    prm.add_(0.05)
    opt = hl.FlatAdam(lm.torch_model, lr=1e-2)
    opt.broadcast_parameters()
    log = _wire_early_bucket(lm, opt, world) if early else None
    p0 = {n: p.detach().clone() for n, p in lm.torch_model.named_parameters()}
    batch = bench.make_train_batches(1, B, L, V, n_neg, rank)[0]                 # rank-dependent data
    ops.RNG.next_step()
    opt.zero_grad()
    lm.training_loss(batch).backward()
    bwd_end = torch.cuda.Event(enable_timing=True)
    bwd_end.record()
    local = {n: p.grad.detach().clone() for n, p in lm.torch_model.named_parameters()}
    opt.step(world)
    torch.cuda.synchronize()
    if early:
        assert log[0][2].elapsed_time(bwd_end) > 0.0      # device time between the hook and the end of the backward pass: the lookup's backward
    g1 = {}
    for n, p in lm.torch_model.named_parameters():
        g = local[n].clone()
        dist.all_reduce(g)                                                       # independent of the product path
        g /= world
        g1[n] = g
        want = p0[n] - 1e-2 * g / (g.abs() + 1e-8)                               # first Adam step: m_hat = g, v_hat = g^2
        torch.testing.assert_close(p.detach(), want, rtol=2e-4, atol=2e-6, msg=lambda m, n=n: f"{n}: {m}")
    if shape == "c2":
        # SECOND step: from now on the sampled loss writes the item table's gradient into its segment of the flat gradient buffer
        # (`ops._TABLE_GRAD_HOME`, registered by the first pack) and the pack skips it — same Adam step as from separate gradients
        p1 = {n: p.detach().clone() for n, p in lm.torch_model.named_parameters()}
        ops.RNG.next_step()
        opt.zero_grad()
        lm.training_loss(batch).backward()
        table_name, table = max(lm.torch_model.named_parameters(), key=lambda kv: kv[1].numel())
        i = [q is table for q in opt.params].index(True)
        assert table.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * opt._offsets[i], "the table's gradient was not produced in place"
        local = {n: p.grad.detach().clone() for n, p in lm.torch_model.named_parameters()}
        opt.step(world)
        torch.cuda.synchronize()
        b1, b2 = opt.betas
        for n, p in lm.torch_model.named_parameters():
            g = local[n].clone()
            dist.all_reduce(g)
            g /= world
            m = (b1 * (1 - b1) * g1[n] + (1 - b1) * g) / (1 - b1 ** 2)
            v = (b2 * (1 - b2) * g1[n] ** 2 + (1 - b2) * g ** 2) / (1 - b2 ** 2)
            want = p1[n] - 1e-2 * m / (v.sqrt() + 1e-8)
            solid = (g1[n].abs() + g.abs()) > 1e-7      # (entries both gradients leave at ~0 move by rounding noise over eps)
            torch.testing.assert_close(p.detach()[solid], want[solid], rtol=5e-4, atol=5e-6, msg=lambda m_, n=n: f"second step, {n}: {m_}")
    bad = []
    for n, p in lm.torch_model.named_parameters():                               # replicas stay bit-identical
        for what, t in (("init", p0[n]), ("after", p.detach())):
            both = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(both, t.contiguous())
            if not torch.equal(both[0], both[1]):
                bad.append((what, n, float((both[0] - both[1]).abs().max())))
    assert not bad, bad
    if early:
        _check_early_log(log, opt, 2 if shape == "c2" else 1)
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.ones(1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape", ["small", "c2"])
@pytest.mark.parametrize("exchange", ["allreduce", "sharded", "early"])
def test_two_rank_step_matches_mean_gradient_adam(tmp_path, exchange, shape):
    """`early`: the all-reduce exchange in two buckets — the block weights' all-reduce is started by the gradient hook of the blocks' input
    (FlatAdam.begin_early_exchange: packed on the exchange stream behind the weight-gradient side streams), the embeddings' follows in
    step(), Adam runs per bucket; asserted: every block gradient exists at the hook and no lookup gradient does, the device ran the lookup's
    backward after the hook, same Adam step, replicas bit-identical.  `sharded`: reduce-scatter -> rt_adam_step on the rank's 1/N slice of (p, m, v) -> all-gather of the parameters
    (FlatAdam.step_sharded; over gloo with both ranks on this GPU): the same first Adam step, replicas bit-identical — on a small model
    and at the C2 model size (the HIP Adam kernel on a 15.5 MB slice of the 31 MB flat buffers)."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), exchange, shape), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}.npy").exists() for r in range(world))


def _reco_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pandas as pd

    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    rng = np.random.default_rng(3)
    n = 4000
    df = pd.DataFrame({"user_id": rng.integers(0, 200, n) * 2 + 5, "item_id": rng.integers(0, 90, n) + 100, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 40_000, n), unit="m")})
    ds = Dataset.construct(df)
    # fit() under the process group = data-parallel training: both replicas end with the same weights
    model = SASRecModel(n_factors=32, n_blocks=1, n_heads=2, session_max_len=10, batch_size=32, epochs=1, loss="sampled_softmax",
                        n_negatives=4, seed=5).fit(ds)
    users = rng.permutation(ds.user_id_map.external_ids)[:101]                   # odd count: uneven slices
    for kw in (dict(k=5, filter_viewed=True), dict(k=3, filter_viewed=False, items_to_recommend=np.arange(100, 130))):
        whole = model.recommend(users=users, dataset=ds, **kw)                     # single-process path on this rank
        sharded = model.recommend_distributed(users, ds, **kw)                     # slice per rank + gather
        pd.testing.assert_frame_equal(sharded, whole)
    only_one = model.recommend_distributed(users[:1], ds, k=2, filter_viewed=False)   # rank 0's slice is empty
    assert only_one["user_id"].nunique() == 1 and len(only_one) == 2
    np.save(os.path.join(out_dir, f"reco_ok{rank}.npy"), np.ones(1))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_recommend_shards_users_and_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_reco_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"reco_ok{r}.npy").exists() for r in range(world))


def _nccl_worker(rank, world, port, out_dir):
    """The RCCL code path itself: `init_process_group("nccl", device_id=...)` as bench.py does it, parameter broadcast, the
    packed-gradient all-reduce, the flat Adam kernel on the reduced buffer, barrier, teardown — one rank per visible GPU
    (a single rank on the one-GPU test box: the collectives still go through RCCL's stream and launch machinery)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import helpers_dp
    from rectools_amd import lightning as hl
    from rectools_amd import ops

    V, d, H, nb, L, B, n_neg = 300, 64, 2, 2, 24, 16, 8
    lm = helpers_dp.make_sasrec(V, d, H, nb, L, 0.1, "sampled_softmax", n_neg, device=f"cuda:{rank}")
    lm.train()
    opt = hl.FlatAdam(lm.torch_model, lr=1e-2)
    opt.broadcast_parameters(force=True)
    log = _wire_early_bucket(lm, opt, world, force=True)      # the two-bucket exchange through RCCL's own streams (async collectives)
    p0 = {n: p.detach().clone() for n, p in lm.torch_model.named_parameters()}
    with torch.cuda.device(rank):
        batch = helpers_dp.make_train_batches(1, B, L, V, n_neg, rank)[0]
    losses = []
    for it in range(3):                                   # several steps: the side-stream join / RCCL stream ordering repeats
        ops.RNG.next_step()
        opt.zero_grad()
        loss = lm.training_loss(batch)
        loss.backward()
        if it == 0:
            local = {n: p.grad.detach().clone() for n, p in lm.torch_model.named_parameters()}
        opt.step(world, flat=True)
        losses.append(float(loss.detach()))
        if it == 0:
            torch.cuda.synchronize()
            for n, p in lm.torch_model.named_parameters():
                g = local[n].clone()
                dist.all_reduce(g)
                g /= world
                want = p0[n] - 1e-2 * g / (g.abs() + 1e-8)
                torch.testing.assert_close(p.detach(), want, rtol=2e-4, atol=2e-6, msg=lambda m, n=n: f"{n}: {m}")
    assert losses[2] < losses[0], losses                  # same batch three times: the loss must go down
    _check_early_log(log, opt, 3)
    t = torch.tensor([float(rank + 1)], device=f"cuda:{rank}")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # bench.py's max-over-ranks
    assert float(t) == world
    dist.barrier()
    np.save(os.path.join(out_dir, f"nccl_ok{rank}.npy"), np.ones(1))
    dist.destroy_process_group()


def test_rccl_backend_step_on_every_visible_gpu(tmp_path):
    world = min(torch.cuda.device_count(), 2)
    mp.spawn(_nccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"nccl_ok{r}.npy").exists() for r in range(world))


@pytest.mark.gpu
def test_rt_dp_entry_points_single_rank():
    """The library's own RCCL binding (rt_dp_unique_id / init / allreduce / broadcast / finalize): a 1-rank communicator on this
    GPU — the all-reduce of one rank is the identity — and FlatAdam stepping through it equals the plain step."""
    import torch

    from rectools_amd import lightning as hl

    ex = hl.RcclExchange(0, 1)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    ex.all_reduce(x)
    ex.broadcast(x, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(x, ref, rtol=0, atol=0)
    ex.close()

    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        return torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, device="cuda")) for s in [(64, 32), (32,), (128, 64)]])
    pa, pb = make(), make()
    oa, ob = hl.FlatAdam(pa, lr=1e-2), hl.FlatAdam(pb, lr=1e-2)
    ob.use_rccl_exchange(0, 1)
    grads = [torch.randn_like(p) for p in pa]
    for opt, ps in ((oa, pa), (ob, pb)):
        for p, g in zip(ps, grads):
            p.grad = g.clone()
    oa.step()                               # one GPU: segmented kernel, no collective
    ob.step(world_size=1, flat=True)        # pack -> rt_dp_allreduce (1 rank: identity) -> flat Adam kernel
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    ob.exchange.close()
    # the sharded exchange's two collectives on the 1-rank communicator (identity), and a sharded step through them
    ex = hl.RcclExchange(0, 1)
    src = torch.randn(4096, device="cuda")
    dst = torch.zeros(4096, device="cuda")
    ex.reduce_scatter(src, dst)
    out = torch.zeros(4096, device="cuda")
    ex.all_gather(dst, out)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, src, rtol=0, atol=0)
    ex.close()
    pc = make()
    oc = hl.FlatAdam(pc, lr=1e-2)
    oc.use_rccl_exchange(0, 1)
    for p, g in zip(pc, grads):
        p.grad = g.clone()
    oc.step_sharded(1, 0)                   # rt_dp_reduce_scatter -> rt_adam_step on the (whole) slice -> rt_dp_allgather
    for a, c in zip(pa, pc):
        torch.testing.assert_close(a, c, rtol=1e-6, atol=1e-7)
    oc.exchange.close()
