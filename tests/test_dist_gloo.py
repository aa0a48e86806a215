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


def test_bench_two_rank_control_flow_rehearsal():
    """bench.py's own multi-rank control flow (`--gpus 2`, launched exactly as the driver launches it: one process per rank
    through torch.distributed.run) on CPU with gloo and a stand-in pipeline (`--fake-pipeline`: no kernels, not a
    measurement): default of 8 try-ons per rank = BASELINE configs[3], per-image seeds from the GLOBAL image index, barrier +
    max-over-ranks timing, ONE gather of the decoded images on rank 0, one JSON line from rank 0 only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--fake-pipeline"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["images_per_gpu"] == 8 and "configs[3]" in d["config"]["workload"]
    reh = d["rehearsal"]
    assert len(reh["gathered_image_means"]) == 16 and reh["gathered_image_means"] == reh["expected"]
    assert len(set(reh["expected"])) == 16                 # sixteen different try-ons, ordered by global index


def test_bench_launches_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` with NO launcher in front (the form the driver records for its single-GPU run): the parent
    must start the two ranks itself - as child processes, never by touching a device or re-executing itself - forward rank 0's
    one JSON line and exit with the children's code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--fake-pipeline"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo"      # (rehearsal backend)
    assert d["rehearsal"]["gathered_image_means"] == d["rehearsal"]["expected"]
    # a failing child is the parent's failure
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--fake-pipeline",
                          "--batch", "-1"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0
