"""Multi-GPU sharding of image batches.

Images are independent units (the reference processes one image per call and has no exchange
step), so a batch is split over ranks by image index and NO collective touches the data path.
The only communication is the benchmark's bookkeeping: total pixels (sum) and the slowest rank's
time (max)."""
import torch
import torch.distributed as dist


def shard_indices(n_items, world, rank):
    """Contiguous block of item indices owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def bench_image_plan(images_per_gpu, rank):
    """(mode, seed) for the images one rank encodes in the weak-scaling benchmark: global image g
    uses the photo-like generator with seed 12345+g, every 4th image the graphics generator."""
    plan = []
    for k in range(images_per_gpu):
        g = rank * images_per_gpu + k
        plan.append((4 if g % 4 == 3 else 3, 12345 + g))
    return plan


def global_throughput(pixels_local, seconds_local, device=None):
    """(total pixels over all ranks, slowest rank's seconds).  Works without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(pixels_local), float(seconds_local)
    dev = device if device is not None else torch.device("cpu")
    px = torch.tensor([float(pixels_local)], dtype=torch.float64, device=dev)
    sec = torch.tensor([float(seconds_local)], dtype=torch.float64, device=dev)
    dist.all_reduce(px, op=dist.ReduceOp.SUM)
    dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    return int(px.item()), float(sec.item())
