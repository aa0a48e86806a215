"""N>1 path on CPU: world_size-2 gloo processes run the sharding + single-gather logic of edgestyle_amd/dist.py."""
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edgestyle_amd.dist import gather_images, run_sharded, shard_range

    def fake_pipeline(lo, hi, seeds):       # "image" i is filled with its per-image seed -> world-size independent
        return torch.stack([torch.full((3, 4, 4), float(s)) for s in seeds]) if hi > lo else torch.zeros(0, 3, 4, 4)

    lo, hi = shard_range(8, rank, world)
    even = gather_images(fake_pipeline(lo, hi, [42 + i for i in range(lo, hi)]), world)
    ragged = run_sharded(fake_pipeline, 5, seed=42)
    if rank == 0:
        q.put((even[:, 0, 0, 0].tolist(), ragged[:, 0, 0, 0].tolist()))
    else:
        assert even is None and ragged is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    even, ragged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert even == [42.0 + i for i in range(8)]
    assert ragged == [42.0 + i for i in range(5)]
