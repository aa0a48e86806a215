"""Data-parallel sharding of independent try-ons over the GPUs of one node.

The reference has no multi-GPU inference (every script is single-device, TT:216 / APP:42); one 512x512 try-on
fits one GPU and nothing in PL:435-557 couples images, so the path shards by image with NO collective inside the
loop.  One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm and runs over xGMI.  The only
exchange is a single gather of the decoded images at the end (12.6 MB fp32 per 8-image rank: latency-bound, so a
plain gather-to-root — every peer writes its slice over its own point-to-point xGMI link — not a ring).

Per-image RNG seeds depend on the GLOBAL image index, so results are independent of the world size.
"""
from typing import List, Optional, Tuple

import torch


def shard_range(num_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the global image list owned by `rank` (remainder spread over the low ranks)."""
    if num_images < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    base, rem = divmod(num_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seed(seed: int, rank: int, images_per_rank: int) -> int:
    """Seed of the first image of this rank: seed + global image index (weak scaling: images_per_rank fixed)."""
    return seed + rank * images_per_rank


def gather_images(images: torch.Tensor, world: Optional[int] = None, dst: int = 0) -> Optional[torch.Tensor]:
    """[b,3,H,W] per rank -> [world*b,3,H,W] on rank `dst` (None elsewhere).  world == 1: identity, no collective."""
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return images
    images = images.contiguous()
    if dist.get_backend() == "gloo" and images.is_cuda:
        images = images.cpu()            # CPU rehearsal of the multi-rank flow: gloo has no CUDA gather
    rank = dist.get_rank()
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(images) for _ in range(world)] if rank == dst else None
    dist.gather(images, bufs, dst=dst)
    return torch.cat(bufs) if rank == dst else None


def run_sharded(fn, num_images: int, seed: int = 42):
    """Run `fn(lo, hi, seeds) -> images[b,...]` on this rank's shard and gather on rank 0.  Ranks with an empty shard
    contribute a zero-length tensor (gather_object path keeps ragged shards legal)."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(num_images, rank, world)
    out = fn(lo, hi, [seed + i for i in range(lo, hi)])
    if world == 1:
        return out
    if num_images % world == 0:
        return gather_images(out, world)
    objs = [None] * world if rank == 0 else None
    dist.gather_object(out.cpu(), objs, dst=0)
    return torch.cat([o for o in objs if o.numel()]) if rank == 0 else None
