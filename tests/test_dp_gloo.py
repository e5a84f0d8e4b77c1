"""N > 1 data-parallel path on CPU with the gloo backend (world_size 2): the flat-gradient all-reduce of `FlatAdam`
and the sharded sampler give every rank identical averaged gradients over disjoint samples."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rectools_amd.data_preparator import epoch_permutation, shard_indices
    from rectools_amd.lightning import FlatAdam

    torch.manual_seed(0)  # replicas start from identical parameters
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    opt = FlatAdam(model, lr=1e-3)
    p0 = opt.flat_p.clone()
    opt.zero_grad()
    x = torch.full((4, 6), float(rank + 1))
    model(x).sum().backward()                                   # rank-dependent gradients
    local = opt.gather_gradients().clone()                      # per-parameter gradients packed into the flat buffer
    scale = opt.reduce_gradients(world)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(opt.flat_g, sum(gathered))             # sum all-reduce of the flat buffer
    assert scale == 1.0 / world                                  # mean is applied inside the Adam kernel
    assert torch.equal(opt.flat_p, p0)                           # parameters still views of the flat buffer, untouched
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in model.parameters())
    perm = epoch_permutation(11, epoch=0, seed=5, shuffle=True)
    mine = shard_indices(perm, rank, world)
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), mine)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_and_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")
    assert len(a) == len(b) == 6                                  # 11 samples padded to 12
    assert set(a.tolist()) | set(b.tolist()) == set(range(11))


def _torch_adam(p, g, m, v, hyper):
    """torch restatement of rt_adam_step (lightning.py:214-218: Adam, eps outside the bias-corrected sqrt as torch.optim.Adam)."""
    step, lr, b1, b2, eps, scale = hyper
    g = g * scale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v / (1 - b2 ** step)).sqrt_().add_(eps)
    p.addcdiv_(m / (1 - b1 ** step), denom, value=-lr)


def _worker_sharded(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rectools_amd.lightning import FlatAdam

    results = {}
    for mode in ("allreduce", "sharded"):
        os.environ["RT_DP_EXCHANGE"] = mode
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Embedding(30000, 8), torch.nn.Linear(8, 5), torch.nn.Linear(5, 3))   # table >= 1024 rows: padded segment; 240 k floats: every rank's slice holds parameters
        opt = FlatAdam(model, lr=1e-2)
        opt._adam_flat = lambda p, g, m, v, hyper: _torch_adam(p, g, m, v, hyper)      # the HIP kernel's seam (no GPU in this test)
        assert opt.sharded == (mode == "sharded") and opt.flat_p.numel() % 1024 == 0
        for step in range(3):
            opt.zero_grad()
            ids = torch.arange(4000) * (7 * rank + 5) % 30000
            model(ids).pow(2).sum().mul(rank + 1.0).backward()                       # rank-dependent gradients
            opt.step(world, flat=True)
        lo, hi = opt.shard_bounds(world, rank)
        if mode == "sharded":      # the moments outside the rank's own slice are never touched on this rank
            assert opt.partial_moments == (world, rank)
            other = torch.ones_like(opt.m, dtype=torch.bool); other[lo:hi] = False
            assert float(opt.m[other].abs().max()) == 0.0 and (float(opt.m[lo:hi].abs().max()) > 0.0) == (lo < opt.n_used)
            # ... and a checkpoint / state_dict written from them would resume from corrupt Adam state: refused until gathered
            from rectools_amd import checkpoint as ckpt
            for fn in (lambda: ckpt.adam_state_dict(opt), opt.state_dict):
                try:
                    fn()
                    raise AssertionError("partial moments were written out")
                except RuntimeError as e:
                    assert "consolidate_moments" in str(e)
            opt.consolidate_moments()      # collective: what fit() does after its last step
            assert opt.partial_moments is None
            ckpt.adam_state_dict(opt)       # local from here on (any single rank may save)
        m_full, v_full = opt.full_moments(world, rank)
        results[mode] = (opt.flat_p.clone(), m_full.clone(), v_full.clone(), opt.m.clone())
    for a, b in zip(results["allreduce"][:3], results["sharded"][:3]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)                         # same parameters, same (gathered) moments
    mine = [torch.zeros_like(results["sharded"][0]) for _ in range(world)]
    dist.all_gather(mine, results["sharded"][0])
    assert all(torch.equal(mine[0], t) for t in mine)                                  # replicas stay bit-identical
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exchange_equals_the_allreduce_exchange(tmp_path, world):
    """reduce-scatter -> Adam on the rank's 1/N slice of (p, m, v) -> all-gather of the parameters (FlatAdam.step_sharded) against the
    all-reduce exchange, world 2 and 3 over gloo (3 does not divide a power-of-two buffer: the flat buffers are padded to a multiple
    of lcm(1024, 840) floats), 3 steps: same parameters on every rank, same moments once gathered; partial moments are never
    written out."""
    mp.spawn(_worker_sharded, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)


def test_a_world_size_that_does_not_cut_the_buffer_falls_back_to_the_allreduce():
    from rectools_amd.lightning import FLAT_QUANTUM, FlatAdam

    os.environ["RT_DP_EXCHANGE"] = "sharded"
    try:
        opt = FlatAdam(torch.nn.Linear(6, 5), lr=1e-3)
    finally:
        os.environ.pop("RT_DP_EXCHANGE", None)
    assert opt.sharded and opt.flat_p.numel() % FLAT_QUANTUM == 0
    assert all(opt._use_sharded(w) for w in (2, 3, 4, 5, 6, 7, 8, 16)) and not opt._use_sharded(11) and not opt._use_sharded(1)


def _worker_stop_flag(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rectools_amd.models import TransformerModelBase

    dev = torch.device("cpu")
    # only rank 1's callback asks to stop: every rank must leave the epoch loop together (a rank that stops alone meets
    # `consolidate_moments()` while the others enter the next epoch's gradient exchange)
    stop = TransformerModelBase._all_reduce_host([1.0 if rank == 1 else 0.0], "max", dev)[0] > 0.0
    assert stop
    assert not TransformerModelBase._all_reduce_host([0.0], "max", dev)[0] > 0.0
    # ... and the epoch metrics the callbacks decide on are the mean over the ranks, identical everywhere
    train, val = TransformerModelBase._all_reduce_host([1.0 + rank, 10.0 * (rank + 1)], "mean", dev)
    assert abs(train - (1.0 + (world - 1) / 2)) < 1e-12 and abs(val - 10.0 * (world + 1) / 2) < 1e-12
    dist.barrier()
    dist.destroy_process_group()


def test_early_stop_and_epoch_metrics_are_reduced_over_the_ranks(tmp_path):
    mp.spawn(_worker_stop_flag, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    from rectools_amd.models import TransformerModelBase

    assert TransformerModelBase._all_reduce_host([3.0, 4.0], "mean", torch.device("cpu")) == [3.0, 4.0]      # no process group: as given


def _worker_early(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rectools_amd.lightning import FlatAdam

    class Net(torch.nn.Module):      # input embeddings (late: their gradient is the LAST of a backward pass) in front of "block" weights
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Embedding(3000, 8)
            self.pos = torch.nn.Embedding(16, 8)
            self.a, self.b = torch.nn.Linear(8, 8), torch.nn.Linear(8, 3)
            self.on_input_gradient = None

        def forward(self, ids):
            x = self.table(ids) + self.pos(torch.arange(ids.shape[1]))
            if self.on_input_gradient is not None:
                notify = self.on_input_gradient
                x.register_hook(lambda _g: notify())
            return self.b(torch.tanh(self.a(x)))

    results = {}
    for mode in ("one_bucket", "early", "early_stale"):
        os.environ["RT_DP_EXCHANGE"] = "allreduce"
        torch.manual_seed(0)
        net = Net()
        opt = FlatAdam(net, lr=1e-2)
        opt._adam_flat = lambda p, g, m, v, hyper: _torch_adam(p, g, m, v, hyper)
        seen = []
        if mode != "one_bucket":
            opt.set_early_bucket([*net.table.parameters(), *net.pos.parameters()])
            assert opt.early_first == 2 and opt.early_from == opt._offsets[2]      # a, b behind table, pos
            def fire():
                # every gradient of the early bucket exists, none of the late one does: the lookup's backward has not run yet
                seen.append(([p.grad is not None for p in opt.params[opt.early_first:]], [p.grad is not None for p in opt.params[:opt.early_first]]))
                assert opt.begin_early_exchange(world)
            net.on_input_gradient = fire
        for step in range(3):
            opt.zero_grad()
            ids = (torch.arange(64).view(4, 16) * (7 * rank + 5) + step) % 3000
            net(ids).pow(2).sum().mul(rank + 1.0).backward()
            if mode == "early_stale" and step == 1:      # somebody replaces a block gradient behind the hook: the bucket is exchanged again
                net.a.weight.grad = net.a.weight.grad * 1.0
            opt.step(world)
        if mode != "one_bucket":
            assert len(seen) == 3 and all(early == [True] * 4 and late == [False, False] for early, late in seen), seen
            assert opt.early_stats == {"started": 3, "redone": 1 if mode == "early_stale" else 0}
            assert opt._early is None
        results[mode] = (opt.flat_p.clone(), opt.m.clone(), opt.v.clone())
    for mode in ("early", "early_stale"):
        for a, b in zip(results["one_bucket"], results[mode]):
            assert torch.equal(a, b) if world == 2 else torch.allclose(a, b, rtol=1e-6, atol=1e-8)      # (a sum of two is order-free)
    mine = [torch.zeros_like(results["early"][0]) for _ in range(world)]
    dist.all_gather(mine, results["early"][0])
    assert all(torch.equal(mine[0], t) for t in mine)                                  # replicas stay bit-identical
    # an abandoned step (no opt.step behind the hook) is drained by the next zero_grad
    opt.zero_grad()
    net(torch.zeros(2, 16, dtype=torch.long)).sum().backward()
    assert opt._early is not None
    opt.zero_grad()
    assert opt._early is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_early_bucket_exchange_equals_the_one_bucket_step(tmp_path, world):
    """FlatAdam.set_early_bucket / begin_early_exchange: the block weights' all-reduce starts from a hook INSIDE the backward pass (before the
    lookup's gradient exists), the late bucket follows in step(), Adam runs per bucket — same parameters and moments as the one-bucket
    step on every rank; a gradient replaced behind the hook makes the step exchange the bucket again."""
    mp.spawn(_worker_early, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
