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


def visible_gpu_count(kfd_root=None, environ=None):
    """Number of GPUs this process could use, found WITHOUT any HIP call (a launcher that forks one rank per GPU must
    not touch the GPU runtime itself): the driver's topology lists every node with its `simd_count` (0 for CPU nodes),
    and the visible-devices variables of the runtime (a comma-separated list each) can only narrow that down.  None
    when the topology cannot be read at all (the answer is then unknown, which is not the same as zero)."""
    import glob
    import os
    env = os.environ if environ is None else environ
    if kfd_root is None:  # VP8L_KFD_TOPOLOGY: another directory of the same layout (tests)
        kfd_root = env.get("VP8L_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    count = 0
    if not os.path.isdir(kfd_root):
        return None
    for path in glob.glob(os.path.join(kfd_root, "*", "properties")):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            count += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in env:
            listed = [t for t in env[var].split(",") if t.strip() != ""]
            count = min(count, len(listed))
    return count


def launch_ranks(n_ranks, argv, master_port=None, env_extra=None, timeout=None):
    """Starts `argv` n_ranks times as child processes, one per GPU of this node, with the environment torch.distributed
    expects (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT), waits for all of them and returns
    (worst return code, rank 0's stdout).  The caller must not have touched the GPU: the children are fresh processes
    and each one initialises its own device.  The other ranks' stdout is dropped, everybody's stderr passes through."""
    import os
    import socket
    import subprocess
    import sys
    if master_port is None:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        master_port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port))
        env.update(env_extra or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=sys.stderr))
    out0 = b""
    worst = 0
    try:
        out0, _ = procs[0].communicate(timeout=timeout)
        for p in procs:
            rc = p.wait(timeout=timeout)
            worst = rc if rc != 0 and worst == 0 else worst
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()   # exactly the children started here
    return worst, out0.decode()
