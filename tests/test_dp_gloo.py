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
