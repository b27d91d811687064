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


# ---------------------------------------------------------------------------------------------
# The other BASELINE configurations (not the headline line; same JSON shape, one line each).
# ---------------------------------------------------------------------------------------------
def _fence(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _encode_batch_on_device(enc, d_in, ws, hs, dev):
    batch = d_in.shape[0]
    cap = sum(enc.bound(w, h) for w, h in zip(ws, hs))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    ptrs = [d_in[k].data_ptr() for k in range(batch)]
    return d_out, ptrs


def _timed_steps(args, world, dev, step, units_per_step):
    for _ in range(args.warmup):
        step()
    _fence(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _fence(world)
    elapsed = time.perf_counter() - t0
    return sharding.global_throughput(units_per_step * args.steps, elapsed, dev)


def _base_line(args, world, metric, value, elapsed, workload, extra):
    out = {"metric": metric, "value": value, "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic", "config": {"workload": workload, "parallelism": "images sharded over %d GPU(s), no collective" % world}}
    out.update(extra)
    return out


def workload_decode1080p(args, world, rank, local_rank, dev):
    """BASELINE configs[3]: 4096 1920x1080 streams over 8 GPUs = 512 per GPU, half photo-like, half graphics."""
    w, h, batch = 1920, 1080, 512
    npx = w * h
    webp_amd.init()
    d_in = torch.empty((batch, npx), dtype=torch.int32, device=dev)
    plan = [(3 if k % 2 == 0 else 4, 12345 + rank * batch + k) for k in range(batch)]
    for k, (mode, seed) in enumerate(plan):
        webp_amd.generate_device(mode, w, h, seed, 0, d_in[k].data_ptr(), device=local_rank)
    enc = webp_amd.Encoder(local_rank)
    ws, hs = [w] * batch, [h] * batch
    d_out, ptrs = _encode_batch_on_device(enc, d_in, ws, hs, dev)
    off, lens, status = enc.encode_device(ptrs, ws, hs, d_out.data_ptr(), d_out.numel())
    assert (status == 0).all()
    dec = webp_amd.Decoder(local_rank)
    d_dec = torch.empty((batch, npx), dtype=torch.int32, device=dev)
    dargs = ([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], ws, hs, [d_dec[k].data_ptr() for k in range(batch)])
    dev_times = []

    def step():
        st = dec.decode_device(*dargs)
        assert (st == 0).all()
        dev_times.append(dec.last_device_seconds())

    total_px, elapsed = _timed_steps(args, world, dev, step, batch * npx)
    ok = bool(torch.equal(d_dec, d_in))
    if rank != 0:
        return None
    bpp = 8.0 * float(lens.sum()) / (batch * npx)
    dev_s = float(np.mean(dev_times[args.warmup:]))
    alg = (bpp / 8.0 + 4.0) * batch * npx  # stream bytes read + pixels written
    out = _base_line(args, world, "vp8l_decode_megapixels_per_sec", total_px / 1e6 / elapsed, elapsed,
                     "batch of %d synthetic %dx%d streams per rank (half photo-like, half graphics), VP8L decode "
                     "(BASELINE configs[3] = 4096 streams over 8 GPUs); streams and pixels resident in HBM" % (batch, w, h),
                     {"bits_per_pixel": bpp, "roundtrip_verified": ok, "device_ms_per_step": 1e3 * dev_s,
                      "roofline": {"kernel": "whole decode path (k_par_scout_lanes, k_par_place, k_par_jump, inverse transforms) over the end-to-end algorithmic bytes", "bound": "hbm",
                                   "achieved": alg / dev_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / dev_s / 1e9 / HBM_PEAK_GBS,
                                   "traffic": None, "algorithmic_bytes_per_launch": alg}})
    out["cpu_baseline"] = None
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        sample = [bytes(d_out[int(off[k]):int(off[k]) + int(lens[k])].cpu().numpy()) for k in range(threads)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            rcs = list(ex.map(lambda s: O.decode(s)[0], sample))
        dt = time.perf_counter() - t0
        assert all(r == 0 for r in rcs)
        out["cpu_baseline"] = {"value": len(sample) * npx / 1e6 / dt, "unit": "MP/s", "cores": threads, "kind": "port",
                               "sample": "%d of the benchmark's streams, one per host thread, %.1f s wall" % (len(sample), dt)}
    return out


def _encode_workload(args, world, rank, local_rank, dev, w, h, plan, workload, metric="vp8l_encode_megapixels_per_sec"):
    batch = len(plan)
    npx = w * h
    webp_amd.init()
    d_in = torch.empty((batch, npx), dtype=torch.int32, device=dev)
    for k, (mode, seed, ncol) in enumerate(plan):
        webp_amd.generate_device(mode, w, h, seed, ncol, d_in[k].data_ptr(), device=local_rank)
    enc = webp_amd.Encoder(local_rank)
    ws, hs = [w] * batch, [h] * batch
    d_out, ptrs = _encode_batch_on_device(enc, d_in, ws, hs, dev)
    res = {}
    dev_times, stage_acc = [], {}

    def step():
        res["r"] = enc.encode_device(ptrs, ws, hs, d_out.data_ptr(), d_out.numel())
        assert (res["r"][2] == 0).all()
        dev_times.append(enc.last_device_seconds())
        for k2, v in enc.last_stage_seconds().items():
            stage_acc[k2] = stage_acc.get(k2, 0.0) + v

    total_px, elapsed = _timed_steps(args, world, dev, step, batch * npx)
    off, lens, status = res["r"]
    ok, dec_ms = None, None
    if not args.no_verify:
        dec = webp_amd.Decoder(local_rank)
        d_dec = torch.empty((batch, npx), dtype=torch.int32, device=dev)
        dargs = ([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], ws, hs, [d_dec[k].data_ptr() for k in range(batch)])
        dec.decode_device(*dargs)
        st = dec.decode_device(*dargs)
        dec_ms = 1e3 * dec.last_device_seconds()
        ok = bool((st == 0).all() and torch.equal(d_dec, d_in))
    if rank != 0:
        return None
    bpp = 8.0 * float(lens.sum()) / (batch * npx)
    dev_s = float(np.mean(dev_times[args.warmup:]))
    alg = (4.0 + bpp / 8.0) * batch * npx  # the fused ideal: every pixel read once, the stream written once
    n = len(dev_times)
    out = _base_line(args, world, metric, total_px / 1e6 / elapsed, elapsed, workload,
                     {"bits_per_pixel": bpp, "roundtrip_verified": ok, "decode_device_ms": dec_ms, "device_ms_per_step": 1e3 * dev_s,
                      "stage_ms_per_step": {k2: 1e3 * v / n for k2, v in stage_acc.items()},
                      "roofline": {"kernel": "whole encode path", "bound": "hbm", "achieved": alg / dev_s / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": alg / dev_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                   "algorithmic_bytes_per_launch": alg}})
    out["cpu_baseline"] = None
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        O.lib()
        mode, seed, ncol = plan[0]
        sw, sh = (w, h) if npx <= (1 << 23) else (4096, 2048)  # a bounded sample of the same content
        img = O.gen_image(mode, sw, sh, seed, ncol or 200)
        t0 = time.perf_counter()
        rc, data = O.encode(img, sw, sh)
        dt = time.perf_counter() - t0
        assert rc == 0
        out["cpu_baseline"] = {"value": sw * sh / 1e6 / dt, "unit": "MP/s", "cores": 1, "kind": "port",
                               "sample": "one %dx%d image of the first image's content on one host thread, %.1f s" % (sw, sh, dt)}
    return out


def workload_palette2k(args, world, rank, local_rank, dev):
    """BASELINE configs[4]: 1024 2048x2048 palette images over 8 GPUs = 128 per GPU; 2, 4, 16 and 200 colours in turn."""
    plan = [(5, 777 + rank * 128 + k, (2, 4, 16, 200)[k % 4]) for k in range(128)]
    return _encode_workload(args, world, rank, local_rank, dev, 2048, 2048, plan,
                            "batch of 128 synthetic 2048x2048 palette images per rank (2/4/16/200 colours), colour indexing + pixel "
                            "bundling + VP8L encode (BASELINE configs[4] = 1024 images over 8 GPUs); inputs and streams resident in HBM")


def workload_single16k(args, world, rank, local_rank, dev):
    """BASELINE configs[2]: one 16383x16383 image on one GPU (every rank encodes its own: replicas)."""
    plan = [(4, 4242 + rank, 0)]
    return _encode_workload(args, world, rank, local_rank, dev, 16383, 16383, plan,
                            "one synthetic 16383x16383 graphics image per rank (maximum VP8L size), VP8L encode (BASELINE configs[2]); "
                            "input and stream resident in HBM")


OTHER_WORKLOADS = {"decode1080p": workload_decode1080p, "palette2k": workload_palette2k, "single16k": workload_single16k}


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
    ap.add_argument("--workload", default="encode4k", choices=sorted(OTHER_WORKLOADS) + ["encode4k"],
                    help="encode4k = BASELINE configs[1] (the headline line); the others measure configs[2..4]")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    if args.workload != "encode4k":
        out = OTHER_WORKLOADS[args.workload](args, world, rank, local_rank, dev)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

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
