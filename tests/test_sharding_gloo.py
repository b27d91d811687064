"""world_size-2 check (gloo, CPU) of the only distributed logic on the path: which rank owns
which images, and the throughput bookkeeping (sum of pixels, max of seconds)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from webp_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owned = list(sharding.shard_indices(11, world, rank))
    plan = sharding.bench_image_plan(4, rank)
    px, sec = sharding.global_throughput(1000 * (rank + 1), 0.5 + rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, owned)
    q.put((rank, owned, plan, px, sec, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_partition_and_bookkeeping():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    all_owned = []
    for rank, owned, plan, px, sec, gathered in res:
        all_owned += owned
        assert px == 3000 and sec == 1.5          # sum of pixels, slowest rank
        assert [i for g in gathered for i in g] == list(range(11))
        assert plan == [(4 if (rank * 4 + k) % 4 == 3 else 3, 12345 + rank * 4 + k) for k in range(4)]
    assert sorted(all_owned) == list(range(11))     # every image exactly once, no exchange needed
    assert abs(len(res[0][1]) - len(res[1][1])) <= 1


def test_single_process_is_identity():
    assert sharding.global_throughput(7, 0.25) == (7, 0.25)
    assert list(sharding.shard_indices(5, 1, 0)) == [0, 1, 2, 3, 4]
    assert [len(sharding.shard_indices(10, 4, r)) for r in range(4)] == [3, 3, 2, 2]


_LAUNCH_CHILD = """
import os, json
import torch, torch.distributed as dist
dist.init_process_group("gloo")          # RANK / WORLD_SIZE / MASTER_* come from the launcher's environment
t = torch.tensor([float(int(os.environ["LOCAL_RANK"]) + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print(json.dumps({"world": dist.get_world_size(), "sum": t.item()}))
else:
    print("noise from another rank")
dist.barrier()
dist.destroy_process_group()
"""


def test_launcher_starts_one_child_per_rank():
    """What `bench.py --gpus N` does when it is not already under a launcher: N children with the torch.distributed
    environment, rank 0's line returned, every rank's failure reported."""
    import json
    import sys
    rc, out = sharding.launch_ranks(2, [sys.executable, "-c", _LAUNCH_CHILD], timeout=120)
    assert rc == 0
    assert json.loads(out.strip().splitlines()[-1]) == {"world": 2, "sum": 3.0}
    assert "noise" not in out
    rc, out = sharding.launch_ranks(2, [sys.executable, "-c", "import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)"], timeout=60)
    assert rc == 3


def test_visible_gpu_count_reads_the_topology_without_the_runtime(tmp_path):
    root = tmp_path / "nodes"
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        d = root / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    assert sharding.visible_gpu_count(str(root), {}) == 3
    assert sharding.visible_gpu_count(str(root), {"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert sharding.visible_gpu_count(str(root), {"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}) == 1
    assert sharding.visible_gpu_count(str(root), {"CUDA_VISIBLE_DEVICES": ""}) == 0
    assert sharding.visible_gpu_count(str(tmp_path / "missing"), {}) is None


def test_bench_refuses_more_ranks_than_gpus_before_starting_any(tmp_path):
    """`bench.py --gpus N` with N beyond the visible devices ends with an error from the launcher parent itself: no
    child is started (a marker the children would write stays absent) and the GPU runtime is never asked."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    topo = tmp_path / "nodes" / "0"
    topo.mkdir(parents=True)
    (topo / "properties").write_text("simd_count 1024\n")
    have = 1
    marker = tmp_path / "child_started"
    code = ("import os,sys\n"
            "if 'WORLD_SIZE' in os.environ: open(%r,'w').close()\n"
            "sys.argv=['bench.py','--gpus',%r]\n"
            "import runpy; runpy.run_path(%r, run_name='__main__')\n") % (str(marker), str(have + 1), os.path.join(root, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VP8L_BENCH_SHARE_GPU")}
    env["VP8L_KFD_TOPOLOGY"] = str(tmp_path / "nodes")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "device(s) visible" in r.stderr
    assert not marker.exists()
