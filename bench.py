#!/usr/bin/env python3
"""Headline benchmark: VP8L encode throughput on MI355X.

Workload (N=1): BASELINE.json configs[1] -- a batch of 256 synthetic 3840x2160 RGBA images encoded
on one MI355X (image k: photo-like generator, seed 12345+k; every 4th image the graphics generator).
A "step" is one pass of the whole encode path over that batch with the inputs already resident in
HBM; the streams stay in HBM.  With N GPUs every rank encodes its own batch of the same size (images
are independent: no collective on the data path), so value = pixels of all ranks / slowest rank.

Prints ONE JSON line on rank 0.  Extra objects in that line:
  roofline      the dominant kernel (k_lz77_sparse, the match search, which also makes the subtract-green
                residual): algorithmic bytes (13 B per coded pixel: 4 read; residual 4, match 4 and
                advance 1 written) over its mean launch duration, measured here with HIP
                events on the encoder's stream; "traffic" is the HBM byte count of one launch from the
                committed rocprofv3 counter passes (profiles/lz77_traffic.json), null if none fits
  cpu_baseline  the CPU oracle (a port of the reference's single-threaded algorithm, image-parallel
                over host threads) on a bounded sample of the same images, rank 0 at N=1 only
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import webp_amd  # noqa: E402
from webp_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
LZ77_BYTES_PER_PIXEL = 13.0    # 4 B pixel read; 4 B residual + 4 B (length, distance, code) + 1 B advance written


def cpu_baseline(w, h, plan, threads, budget_images):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    O.lib()
    sample = plan[:budget_images]
    imgs = [O.gen_image(m, w, h, s, 200) for m, s in sample]

    def work(i):
        rc, data = O.encode(imgs[i], w, h)
        assert rc == 0
        return len(data)

    t1 = time.perf_counter()
    work(0)                                  # one image on one thread: the reference's own (single-threaded) situation
    one = w * h / 1e6 / (time.perf_counter() - t1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        sizes = list(ex.map(work, range(len(imgs))))
    dt = time.perf_counter() - t0
    mp = len(imgs) * w * h / 1e6
    return {
        "value": mp / dt, "unit": "MP/s", "cores": threads, "kind": "port",
        "sample": "%d of the benchmark's %dx%d images (modes %s), one image per host thread, %.1f s wall"
                  % (len(imgs), w, h, "".join(str(m) for m, _ in sample), dt),
        "single_thread_value": one,
        "bytes": int(sum(sizes)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = min(16, cores))")
    ap.add_argument("--cpu-images", type=int, default=0, help="images in the CPU-baseline sample (0 = one per thread)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    w, h, batch = args.width, args.height, args.batch
    npx = w * h
    plan = sharding.bench_image_plan(batch, rank)

    # ---- inputs resident in HBM (generated on the device) -------------------------------------
    webp_amd.init()
    d_in = torch.empty((batch, npx), dtype=torch.int32, device=dev)
    for k, (mode, seed) in enumerate(plan):
        webp_amd.generate_device(mode, w, h, seed, 0, d_in[k].data_ptr(), device=local_rank)
    enc = webp_amd.Encoder(local_rank)
    cap_one = enc.bound(w, h)
    d_out = torch.empty(batch * cap_one, dtype=torch.uint8, device=dev)
    in_ptrs = [d_in[k].data_ptr() for k in range(batch)]
    ws, hs = [w] * batch, [h] * batch

    def step():
        return enc.encode_device(in_ptrs, ws, hs, d_out.data_ptr(), d_out.numel())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        off, lens, status = step()
        assert (status == 0).all(), status
    lz_times, dev_times, stage_acc = [], [], {}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        off, lens, status = step()
        st = enc.last_stage_seconds()
        lz_times.append(st["match"] - st.get("match_fallback", 0.0))   # the candidate-list kernel alone
        dev_times.append(enc.last_device_seconds())
        for k2, v in st.items():
            stage_acc[k2] = stage_acc.get(k2, 0.0) + v
    fence()
    elapsed = time.perf_counter() - t0
    assert (status == 0).all(), status
    total_px, elapsed = sharding.global_throughput(batch * npx * args.steps, elapsed, dev)

    total_bytes = int(lens.sum())
    bpp = 8.0 * total_bytes / (batch * npx)

    # ---- correctness of what was just measured (outside the timed region) ---------------------
    verify = None
    decode_mps = None
    if not args.no_verify:
        dec = webp_amd.Decoder(local_rank)
        d_dec = torch.empty((batch, npx), dtype=torch.int32, device=dev)
        dargs = ([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], ws, hs,
                 [d_dec[k].data_ptr() for k in range(batch)])
        stt = dec.decode_device(*dargs)          # first call allocates the workspace
        verify = bool((stt == 0).all() and torch.equal(d_dec, d_in))
        d_dec.zero_()
        torch.cuda.synchronize()
        td = time.perf_counter()
        stt = dec.decode_device(*dargs)
        torch.cuda.synchronize()
        td = time.perf_counter() - td
        verify = verify and bool((stt == 0).all() and torch.equal(d_dec, d_in))
        decode_mps = batch * npx / 1e6 / td
        del d_dec

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_px / 1e6 / elapsed
        lz_avg = float(np.mean(lz_times))
        achieved = LZ77_BYTES_PER_PIXEL * batch * npx / lz_avg / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "lz77_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("batch") == batch and tj.get("width") == w and tj.get("height") == h:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "vp8l_encode_megapixels_per_sec", "value": value, "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic %dx%d RGBA images, VP8L encode on one MI355X per rank "
                                   "(BASELINE configs[1]); inputs and streams resident in HBM" % (batch, w, h),
                       "images_per_gpu": batch, "width": w, "height": h, "parallelism": "images sharded over %d GPU(s), no collective" % world},
            "bits_per_pixel": bpp,
            "device_ms_per_step": 1e3 * float(np.mean(dev_times)),
            "stage_ms_per_step": {k2: 1e3 * v / args.steps for k2, v in stage_acc.items()},
            "roundtrip_verified": verify,
            "decode_megapixels_per_sec": decode_mps,
            "roofline": {"kernel": "k_lz77_sparse", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": 1e3 * lz_avg, "algorithmic_bytes_per_launch": LZ77_BYTES_PER_PIXEL * batch * npx},
        }
        if world == 1 and not args.no_cpu:
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            nimg = args.cpu_images or threads
            out["cpu_baseline"] = cpu_baseline(w, h, plan, threads, min(nimg, batch))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
