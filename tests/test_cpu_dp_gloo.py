"""CPU, world_size 2, gloo: the data-parallel plumbing of the hot loop -- the flat-gradient
all-reduce + 1/world scaling gives every rank the mean gradient; ranks take disjoint frames of a shared permutation."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynhor_amd import dist as dh_dist
    from dynhor_amd import schedules
    n = 802491
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(n, generator=g)
    local = grad.clone()
    dh_dist.allreduce_sum_(grad)
    others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    expect = sum(others)
    assert torch.allclose(grad, expect, atol=1e-6)
    assert torch.equal(others[rank], local)
    mean = grad * (1.0 / world)                       # what dh_adam_step's grad_scale applies
    stats = dh_dist.mean_stats(torch.full((8,), float(rank)))
    assert torch.allclose(stats, torch.full((8,), (world - 1) / 2.0))
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(4321))
    frames = [int(perm[schedules.frame_slot(it, rank, world) % 64]) for it in range(32)]
    torch.save({"mean": mean, "frames": frames}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_and_frame_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["mean"], r1["mean"]), "every rank must step with the identical mean gradient"
    assert len(set(r0["frames"]) & set(r1["frames"])) == 0, "ranks take disjoint frames within an epoch"
    assert sorted(r0["frames"] + r1["frames"]) == list(range(64)), "one epoch covers every frame exactly once"
