#!/usr/bin/env python3
"""Headline benchmark: VP8L encode throughput on MI355X.

Workload (N=1): BASELINE.json configs[1] -- a batch of 256 synthetic 3840x2160 RGBA images encoded
on one MI355X (image k: photo-like generator, seed 12345+k; every 4th image the graphics generator).
A "step" is one pass of the whole encode path over that batch with the inputs already resident in
HBM; the streams stay in HBM.  With N GPUs every rank encodes its own batch of the same size (images
are independent: no collective on the data path), so value = pixels of all ranks / slowest rank.

`python bench.py --gpus N` without a launcher's environment starts N ranks itself (one child process per
GPU, before this process touches a GPU); under torch.distributed.run it is one of the ranks.

Prints ONE JSON line on rank 0.  Extra objects in that line:
  roofline         the dominant kernel (the match search, which also makes the subtract-green residual): algorithmic
                   bytes over its mean launch duration, measured here with HIP events on the encoder's stream;
                   "traffic" is the HBM byte count of one launch from the committed rocprofv3 counter passes
                   (profiles/lz77_traffic.json), null if none fits this batch
  roofline_stages  every stage of the encode path: algorithmic bytes per coded pixel, ms per step, fraction of HBM peak;
                   and the sum of the stages' bytes against the fused ideal (4 + bpp/8 bytes per pixel)
  decode           the same streams decoded back on the device: MP/s over >= 3 timed repetitions and its roofline
  host_to_host     pinned host memory -> pinned host memory through vp8l_encode_batch / vp8l_decode_batch (PCIe included;
                   never `value`)
  cpu_baseline     the CPU oracle (a port of the reference's single-threaded algorithm, image-parallel
                   over host threads) on a bounded sample of the same images, rank 0 at N=1 only
"""
import argparse
import ctypes as C
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


def stage_bytes_per_pixel(bpp, palette_share=0.0, predictor=False):
    """Algorithmic HBM bytes per coded pixel of every encode stage (reads + writes each stage cannot avoid given the
    arrays it is handed), keyed like Encoder.last_stage_seconds().  palette_share: the share of the pixels that belongs to
    palette images (only those are scanned to the end and go through k_transform; the others leave the palette scan after a
    few hundred pixels per block and get their residual from the match kernel).  predictor: the predictor transform runs."""
    if predictor:
        transform = 1.0 + 8.0             # mode choice (every fourth pixel is a sample; green subtracted on the way in), residual (4 + 4)
    else:
        transform = 8.0 * palette_share   # indices: 4 read, at most 4 written
    return {
        "palette": 4.0 * palette_share,
        "transform": transform,
        "match": 13.0,                       # 4 read; residual 4 + match 4 + advance 1 written
        # advances read twice, block exits, token-start bits written; residual + match read to count the symbols; the symbol
        # counts of every unit of 4096 positions written (1400 x 2 bytes)
        "parse": 2.0 + 1.0 / 8 + 1.0 / 128 + 8.0 + 2800.0 / 4096,
        "histogram": 0.0,                    # the default coder counts inside "parse"; only the extended coder has this stage
        "header": 0.0,                       # per image, not per pixel
        "bitcount_scan": 2800.0 / 4096,      # the units' symbol counts read (their bits are a sum over the symbols: no pass over the pixels)
        "emit": 8.0 + 1.0 / 8 + bpp / 8.0,   # residual + match + token-start bits read, the stream written
    }


def host_threads():
    """Host threads for the CPU baseline: the cores this process may run on, capped by the share the box gives one GPU's
    job (OMP_NUM_THREADS there; os.cpu_count() shows the whole machine's CPUs, many times as many)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


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
        "host_cpus_visible": os.cpu_count(),
        "bytes": int(sum(sizes)),
    }


# VP8L_BENCH_SHARE_GPU=1: ranks may share GPUs and talk over gloo -- only to rehearse `--gpus N` where N GPUs are not at hand;
# the figures of such a run mean nothing.
SHARE_GPU = os.environ.get("VP8L_BENCH_SHARE_GPU") == "1"


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
    return sharding.global_throughput(units_per_step * args.steps, elapsed, None if SHARE_GPU else dev)


def _base_line(args, world, metric, value, elapsed, workload, extra):
    out = {"metric": metric, "value": value, "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic", "config": {"workload": workload, "parallelism": "images sharded over %d GPU(s), no collective" % world}}
    out.update(extra)
    return out


def _decode_args(data_ptrs, lengths, widths, heights, out_ptrs):
    """The arguments of Decoder.decode_device as C arrays, made once: a caller of the C ABI has them as arrays already, and
    converting five Python lists per call is not part of the decoder."""
    return (webp_amd.c_array(C.c_void_p, data_ptrs), webp_amd.c_array(C.c_size_t, lengths), webp_amd.c_array(C.c_uint32, widths),
            webp_amd.c_array(C.c_uint32, heights), webp_amd.c_array(C.c_void_p, out_ptrs))


def _timed_decode(dec, dargs, npx_total, reps):
    """Device-resident decode of a batch: one untimed call (workspace), then `reps` timed ones.  Returns (MP/s over the
    wall clock of the timed calls, mean device seconds per call, status of the last call)."""
    st = dec.decode_device(*dargs)
    torch.cuda.synchronize()
    dev_s = []
    t0 = time.perf_counter()
    for _ in range(reps):
        st = dec.decode_device(*dargs)
        dev_s.append(dec.last_device_seconds())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * npx_total / 1e6 / dt, float(np.mean(dev_s)), st


def _decode_traffic(batch, w, h):
    """HBM bytes of one decode call from the committed counter passes (profiles/decode_traffic.json), if they were made for this batch."""
    path = os.path.join(ROOT, "profiles", "decode_traffic.json")
    try:
        tj = json.load(open(path))
        if (tj.get("batch"), tj.get("width"), tj.get("height")) == (batch, w, h):
            return tj.get("hbm_bytes_per_call")
    except Exception:
        pass
    return None


def _decode_roofline(bpp, npx_total, dev_s, traffic=None):
    alg = (bpp / 8.0 + 4.0) * npx_total   # stream bytes read + pixels written
    return {"kernel": "whole decode path (k_par_scout_lanes, k_par_place, k_par_jump, inverse transforms) over the end-to-end algorithmic bytes",
            "bound": "hbm", "achieved": alg / dev_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / dev_s / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "algorithmic_bytes_per_launch": alg}


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
    dargs = _decode_args([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], ws, hs, [d_dec[k].data_ptr() for k in range(batch)])
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
    out = _base_line(args, world, "vp8l_decode_megapixels_per_sec", total_px / 1e6 / elapsed, elapsed,
                     "batch of %d synthetic %dx%d streams per rank (half photo-like, half graphics), VP8L decode "
                     "(BASELINE configs[3] = 4096 streams over 8 GPUs); streams and pixels resident in HBM" % (batch, w, h),
                     {"bits_per_pixel": bpp, "roundtrip_verified": ok, "device_ms_per_step": 1e3 * dev_s,
                      "roofline": _decode_roofline(bpp, batch * npx, dev_s, _decode_traffic(batch, w, h))})
    out["cpu_baseline"] = None
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        threads = args.cpu_threads or host_threads()
        sample = [bytes(d_out[int(off[k]):int(off[k]) + int(lens[k])].cpu().numpy()) for k in range(threads)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            rcs = list(ex.map(lambda s: O.decode(s)[0], sample))
        dt = time.perf_counter() - t0
        assert all(r == 0 for r in rcs)
        out["cpu_baseline"] = {"value": len(sample) * npx / 1e6 / dt, "unit": "MP/s", "cores": threads, "kind": "port",
                               "host_cpus_visible": os.cpu_count(),
                               "sample": "%d of the benchmark's streams, one per host thread, %.1f s wall" % (len(sample), dt)}
    return out


def _stage_table(stage_ms, bpp, npx_total, palette_share=0.0, predictor=False):
    per_px = stage_bytes_per_pixel(bpp, palette_share, predictor)
    rows, total_bytes = [], 0.0
    for name, ms in stage_ms.items():
        if name not in per_px or name == "histogram":
            continue
        b = per_px[name]
        total_bytes += b
        if b == 0.0:  # nothing per pixel to move (per-image work, or a stage these images leave at once): no rate to quote
            rows.append({"stage": name, "alg_bytes_per_px": 0.0, "ms": ms, "GBps": None, "frac": None})
            continue
        gbs = b * npx_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        rows.append({"stage": name, "alg_bytes_per_px": round(b, 4), "ms": ms, "GBps": gbs, "frac": gbs / HBM_PEAK_GBS})
    ideal = 4.0 + bpp / 8.0
    return rows, {"sum_of_stages": total_bytes, "fused_ideal": ideal, "ratio": total_bytes / ideal}


def _encode_workload(args, world, rank, local_rank, dev, w, h, plan, workload, metric="vp8l_encode_megapixels_per_sec",
                     predictor_bits=0, with_predictor_variant=False):
    batch = len(plan)
    npx = w * h
    webp_amd.init()
    d_in = torch.empty((batch, npx), dtype=torch.int32, device=dev)
    for k, (mode, seed, ncol) in enumerate(plan):
        webp_amd.generate_device(mode, w, h, seed, ncol, d_in[k].data_ptr(), device=local_rank)
    enc = webp_amd.Encoder(local_rank)
    if predictor_bits:
        enc.set_predictor_bits(predictor_bits)
    ws, hs = [w] * batch, [h] * batch
    d_out, ptrs = _encode_batch_on_device(enc, d_in, ws, hs, dev)
    res = {}
    dev_times, stage_acc = [], {}

    c_ptrs, c_ws, c_hs = webp_amd.c_array(C.c_void_p, ptrs), webp_amd.c_array(C.c_uint32, ws), webp_amd.c_array(C.c_uint32, hs)

    def step():
        res["r"] = enc.encode_device(c_ptrs, c_ws, c_hs, d_out.data_ptr(), d_out.numel())
        assert (res["r"][2] == 0).all()
        dev_times.append(enc.last_device_seconds())
        for k2, v in enc.last_stage_seconds().items():
            stage_acc[k2] = stage_acc.get(k2, 0.0) + v

    total_px, elapsed = _timed_steps(args, world, dev, step, batch * npx)
    off, lens, status = res["r"]
    dense, tiles = enc.last_dense_tiles()
    kinds = enc.last_tile_kinds()
    ok, decode = None, None
    bpp = 8.0 * float(lens.sum()) / (batch * npx)
    if not args.no_verify:
        dec = webp_amd.Decoder(local_rank)
        d_dec = torch.empty((batch, npx), dtype=torch.int32, device=dev)
        dargs = _decode_args([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], ws, hs, [d_dec[k].data_ptr() for k in range(batch)])
        mps, dsec, st = _timed_decode(dec, dargs, batch * npx, 3)
        ok = bool((st == 0).all() and torch.equal(d_dec, d_in))
        decode = {"value": mps, "unit": "MP/s", "reps": 3, "device_ms": 1e3 * dsec, "roofline": _decode_roofline(bpp, batch * npx, dsec)}
        del d_dec, dec
    extra_variant = None
    if with_predictor_variant:
        del d_out
        enc.close()
        torch.cuda.empty_cache()
        sub = argparse.Namespace(**vars(args))
        sub.no_cpu = True
        extra_variant = _encode_workload(sub, world, rank, local_rank, dev, w, h, plan, workload + " -- with the predictor transform (tiles of 32)",
                                         metric, predictor_bits=5)
    if rank != 0:
        return None
    dev_s = float(np.mean(dev_times[args.warmup:]))
    alg = (4.0 + bpp / 8.0) * batch * npx  # the fused ideal: every pixel read once, the stream written once
    n = len(dev_times)
    stage_ms = {k2: 1e3 * v / n for k2, v in stage_acc.items()}
    palette_share = sum(1 for m, _, _ in plan if m == 5) / float(batch)
    rows, sums = _stage_table(stage_ms, bpp, batch * npx, palette_share, bool(predictor_bits))
    out = _base_line(args, world, metric, total_px / 1e6 / elapsed, elapsed, workload,
                     {"bits_per_pixel": bpp, "roundtrip_verified": ok, "decode": decode, "device_ms_per_step": 1e3 * dev_s,
                      "stage_ms_per_step": stage_ms, "roofline_stages": rows, "encode_bytes_per_pixel": sums,
                      "exhaustive_search_tiles": {"tiles": dense, "of": tiles}, "tile_kinds": kinds,
                      "roofline": {"kernel": "whole encode path", "bound": "hbm", "achieved": alg / dev_s / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": alg / dev_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                   "algorithmic_bytes_per_launch": alg}})
    if extra_variant is not None:
        out["predictor_variant"] = {k2: extra_variant[k2] for k2 in ("value", "ms_per_step", "bits_per_pixel", "roundtrip_verified", "decode",
                                                                       "device_ms_per_step", "stage_ms_per_step", "exhaustive_search_tiles")}
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
    """BASELINE configs[2]: one 16383x16383 image on one GPU (every rank encodes its own: replicas), without and with the
    tile-parallel predictor in front of the LZ77 coder."""
    plan = [(4, 4242 + rank, 0)]
    return _encode_workload(args, world, rank, local_rank, dev, 16383, 16383, plan,
                            "one synthetic 16383x16383 graphics image per rank (maximum VP8L size), VP8L encode (BASELINE configs[2]); "
                            "input and stream resident in HBM", with_predictor_variant=True)


def workload_predictor4k(args, world, rank, local_rank, dev):
    """The headline batch's content with the predictor transform switched on (not the reference encoder's bytes)."""
    plan = [(m, s, 0) for m, s in sharding.bench_image_plan(64, rank)]
    return _encode_workload(args, world, rank, local_rank, dev, 3840, 2160, plan,
                            "batch of 64 synthetic 3840x2160 images per rank, predictor transform (tiles of 32) + VP8L encode; inputs and "
                            "streams resident in HBM", predictor_bits=5)


def workload_single512(args, world, rank, local_rank, dev):
    """BASELINE configs[0]: one 512x512 image, encode + decode -- the case the reference's own CPU path runs.  Latency of one
    call, through the host-memory entry points a caller of the reference would use (vp8l_encode / vp8l_decode: upload, kernels,
    download, allocation of the result) and device-resident; the CPU port on one thread beside it (the reference is
    single-threaded)."""
    w = h = 512
    npx = w * h
    webp_amd.init()
    d_in = torch.empty(npx, dtype=torch.int32, device=dev)
    webp_amd.generate_device(3, w, h, 12345, 0, d_in.data_ptr(), device=local_rank)
    img = d_in.cpu().numpy().view(np.uint32)
    reps = max(5, args.steps * 5)

    def timed(fn):
        for _ in range(max(1, args.warmup)):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps, r

    enc_ms, stream = timed(lambda: webp_amd.encode(img, w, h))
    dec_ms, dec = timed(lambda: webp_amd.decode(stream))
    ok = bool(np.array_equal(dec[0], img))
    enc = webp_amd.Encoder(local_rank)
    cap = enc.bound(w, h)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    e_dev_ms, res = timed(lambda: enc.encode_device([d_in.data_ptr()], [w], [h], d_out.data_ptr(), cap))
    e_dev_kernels = 1e3 * enc.last_device_seconds()
    off, lens, st = res
    dec_ctx = webp_amd.Decoder(local_rank)
    d_dec = torch.empty(npx, dtype=torch.int32, device=dev)
    dargs = _decode_args([d_out.data_ptr() + int(off[0])], [int(lens[0])], [w], [h], [d_dec.data_ptr()])
    d_dev_ms, st2 = timed(lambda: dec_ctx.decode_device(*dargs))
    d_dev_kernels = 1e3 * dec_ctx.last_device_seconds()
    ok = ok and bool(st2[0] == 0 and torch.equal(d_dec, d_in))
    total_ms = enc_ms + dec_ms
    out = _base_line(args, world, "vp8l_encode_plus_decode_megapixels_per_sec", 2 * npx / 1e6 / (total_ms * 1e-3), total_ms * 1e-3 * args.steps,
                     "one synthetic 512x512 image, VP8L encode then decode, one call each through vp8l_encode / vp8l_decode from and to host "
                     "memory (BASELINE configs[0]); mean of %d calls" % reps,
                     {"bits_per_pixel": 8.0 * len(stream) / npx, "roundtrip_verified": ok,
                      "encode_ms": enc_ms, "decode_ms": dec_ms,
                      "device_resident": {"encode_ms": e_dev_ms, "decode_ms": d_dev_ms, "encode_kernels_ms": e_dev_kernels, "decode_kernels_ms": d_dev_kernels},
                      "reference_cpu_one_thread_ms": {"encode": 131.0, "decode": 6.3,
                                                      "source": "the reference's own encoder and decoder compiled with its flags, survey container (not this box)"}})
    out["ms_per_step"] = total_ms
    out["cpu_baseline"] = None
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        O.lib()
        t0 = time.perf_counter()
        for _ in range(3):
            rc, data = O.encode(img, w, h)
        t1 = time.perf_counter()
        for _ in range(10):
            rc2 = O.decode(data)[0]
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0 and data == stream
        pe, pd = 1e3 * (t1 - t0) / 3, 1e3 * (t2 - t1) / 10
        out["cpu_baseline"] = {"value": 2 * npx / 1e6 / ((pe + pd) * 1e-3), "unit": "MP/s", "cores": 1, "kind": "port", "encode_ms": pe, "decode_ms": pd,
                               "sample": "the same image, encode (3 calls) and decode (10 calls) on one host thread"}
    return out


OTHER_WORKLOADS = {"decode1080p": workload_decode1080p, "palette2k": workload_palette2k, "single16k": workload_single16k,
                   "predictor4k": workload_predictor4k, "single512": workload_single512}
# what the default run adds to its line under "configs": BASELINE's other configurations, a few steps each
CONFIG_RUNS = (("configs[0] single512", "single512"), ("configs[2] single16k", "single16k"), ("configs[3] decode1080p", "decode1080p"),
               ("configs[4] palette2k", "palette2k"), ("predictor4k", "predictor4k"))
CONFIG_KEYS = ("metric", "value", "unit", "ms_per_step", "bits_per_pixel", "roundtrip_verified", "device_ms_per_step", "stage_ms_per_step", "roofline",
               "exhaustive_search_tiles", "tile_kinds", "encode_ms", "decode_ms", "device_resident", "reference_cpu_one_thread_ms", "cpu_baseline")


def host_to_host(local_rank, d_in, w, h, images):
    """vp8l_encode_batch / vp8l_decode_batch from page-locked host memory to page-locked host memory (the library returns
    its results in page-locked blocks), wall clock around the calls: what a caller with host buffers sees."""
    L = webp_amd.lib()
    n = min(images, d_in.shape[0])
    npx = w * h
    host = torch.empty((n, npx), dtype=torch.int32, pin_memory=True)
    host.copy_(d_in[:n])
    torch.cuda.synchronize()
    ptrs = (C.c_void_p * n)(*[host[k].data_ptr() for k in range(n)])
    ws = (C.c_uint32 * n)(*([w] * n))
    hs = (C.c_uint32 * n)(*([h] * n))
    status = (C.c_int * n)()
    enc_s, dec_s = [], []
    outs = lens = None
    for rep in range(5):   # the first call builds the lanes' contexts and staging buffers; the mean of the other four counts
        if outs is not None:
            for k in range(n):
                L.vp8l_free(C.c_void_p(outs[k]))
        outs = (C.c_void_p * n)()
        lens = (C.c_size_t * n)()
        t0 = time.perf_counter()
        rc = L.vp8l_encode_batch(local_rank, n, ptrs, ws, hs, outs, lens, status)
        enc_s.append(time.perf_counter() - t0)
        assert rc == 0 and all(s == 0 for s in status), (rc, list(status))
    ok = True
    for rep in range(5):
        argb = (C.c_void_p * n)()
        ow = (C.c_uint32 * n)()
        oh = (C.c_uint32 * n)()
        t0 = time.perf_counter()
        rc = L.vp8l_decode_batch(local_rank, n, outs, lens, argb, ow, oh, status)
        dec_s.append(time.perf_counter() - t0)
        assert rc == 0 and all(s == 0 for s in status), (rc, list(status))
        if rep == 4:
            for k in range(0, n, max(1, n // 8)):
                got = np.ctypeslib.as_array(C.cast(argb[k], C.POINTER(C.c_uint32)), shape=(npx,))
                ok = ok and bool(np.array_equal(got, host[k].numpy().view(np.uint32)))
        for k in range(n):
            L.vp8l_free(C.c_void_p(argb[k]))
    for k in range(n):
        L.vp8l_free(C.c_void_p(outs[k]))
    e, d = sum(enc_s[1:]) / len(enc_s[1:]), sum(dec_s[1:]) / len(dec_s[1:])
    return {"encode": n * npx / 1e6 / e, "decode": n * npx / 1e6 / d, "unit": "MP/s", "images": n,
            "encode_GBps_in": n * npx * 4 / 1e9 / e, "decode_GBps_out": n * npx * 4 / 1e9 / d, "roundtrip_verified": ok,
            "what": "page-locked host memory in, page-locked host memory out, through vp8l_encode_batch / vp8l_decode_batch; mean of 4 "
                    "calls after one that builds the staging buffers; PCIe both ways included",
            "encode_calls_s": [round(x, 4) for x in enc_s[1:]], "decode_calls_s": [round(x, 4) for x in dec_s[1:]]}


def workload_encode4k(args, world, rank, local_rank, dev):
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

    for _ in range(args.warmup):
        off, lens, status = step()
        assert (status == 0).all(), status
    lz_times, dev_times, stage_acc = [], [], {}
    _fence(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        off, lens, status = step()
        st = enc.last_stage_seconds()
        lz_times.append(st["match"] - st.get("match_fallback", 0.0))   # the candidate-list kernel alone
        dev_times.append(enc.last_device_seconds())
        for k2, v in st.items():
            stage_acc[k2] = stage_acc.get(k2, 0.0) + v
    _fence(world)
    elapsed = time.perf_counter() - t0
    assert (status == 0).all(), status
    total_px, elapsed = sharding.global_throughput(batch * npx * args.steps, elapsed, None if SHARE_GPU else dev)

    total_bytes = int(lens.sum())
    bpp = 8.0 * total_bytes / (batch * npx)

    # ---- correctness of what was just measured (outside the timed region), and the decode figure -------------
    verify, decode = None, None
    if not args.no_verify:
        dec = webp_amd.Decoder(local_rank)
        d_dec = torch.empty((batch, npx), dtype=torch.int32, device=dev)
        dargs = _decode_args([d_out.data_ptr() + int(o) for o in off], [int(x) for x in lens], ws, hs,
                             [d_dec[k].data_ptr() for k in range(batch)])
        mps, dsec, stt = _timed_decode(dec, dargs, batch * npx, 3)
        verify = bool((stt == 0).all() and torch.equal(d_dec, d_in))
        decode = {"value": mps, "unit": "MP/s", "reps": 3, "device_ms": 1e3 * dsec, "roofline": _decode_roofline(bpp, batch * npx, dsec)}
        del d_dec, dec
        torch.cuda.empty_cache()
    if rank != 0:
        return None
    ms_per_step = 1e3 * elapsed / args.steps
    value = total_px / 1e6 / elapsed
    lz_avg = float(np.mean(lz_times))
    lz_bytes = stage_bytes_per_pixel(bpp)["match"]
    achieved = lz_bytes * batch * npx / lz_avg / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "lz77_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("batch") == batch and tj.get("width") == w and tj.get("height") == h:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    stage_ms = {k2: 1e3 * v / args.steps for k2, v in stage_acc.items()}
    rows, sums = _stage_table(stage_ms, bpp, batch * npx)
    dev_s = float(np.mean(dev_times))
    out = {
        "metric": "vp8l_encode_megapixels_per_sec", "value": value, "unit": "MP/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "batch of %d synthetic %dx%d RGBA images, VP8L encode on one MI355X per rank "
                               "(BASELINE configs[1]); inputs and streams resident in HBM" % (batch, w, h),
                   "images_per_gpu": batch, "width": w, "height": h, "parallelism": "images sharded over %d GPU(s), no collective" % world},
        "bits_per_pixel": bpp,
        "device_ms_per_step": 1e3 * dev_s,
        "stage_ms_per_step": stage_ms,
        "roundtrip_verified": verify,
        "decode": decode,
        "decode_megapixels_per_sec": decode["value"] if decode else None,
        "roofline": {"kernel": "k_lz77_sparse", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "launch_ms": 1e3 * lz_avg, "algorithmic_bytes_per_launch": lz_bytes * batch * npx},
        "roofline_stages": rows,
        "encode_bytes_per_pixel": sums,
        "roofline_end_to_end": {"bound": "hbm", "achieved": (4.0 + bpp / 8.0) * batch * npx / dev_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": (4.0 + bpp / 8.0) * batch * npx / dev_s / 1e9 / HBM_PEAK_GBS},
    }
    out["host_to_host"] = None
    if world == 1 and not args.no_host:
        del d_out
        torch.cuda.empty_cache()
        out["host_to_host"] = host_to_host(local_rank, d_in, w, h, args.host_images)
    if world == 1 and not args.no_cpu:
        threads = args.cpu_threads or host_threads()
        nimg = args.cpu_images or threads
        out["cpu_baseline"] = cpu_baseline(w, h, plan, threads, min(nimg, batch))
    else:
        out["cpu_baseline"] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = the cores this job may use)")
    ap.add_argument("--cpu-images", type=int, default=0, help="images in the CPU-baseline sample (0 = one per thread)")
    ap.add_argument("--host-images", type=int, default=128, help="images of the batch used for the host-memory-to-host-memory figure")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="the headline workload only: leave BASELINE's other configurations out of the line")
    ap.add_argument("--workload", default="encode4k", choices=sorted(OTHER_WORKLOADS) + ["encode4k"],
                    help="encode4k = BASELINE configs[1] (the headline line); the others measure configs[2..4]")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start one rank per GPU ourselves, before anything here touches a GPU
        have = sharding.visible_gpu_count()   # from the driver's topology files: no call into the GPU runtime here
        if have is not None and have < args.gpus and not SHARE_GPU:
            sys.exit("bench.py: --gpus %d but only %d device(s) visible" % (args.gpus, have))
        rc, out0 = sharding.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(out0)
        sys.exit(rc)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if SHARE_GPU:  # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks: every rank uses a GPU that exists
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if SHARE_GPU:
            dist.init_process_group("gloo")  # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    fn = workload_encode4k if args.workload == "encode4k" else OTHER_WORKLOADS[args.workload]
    out = fn(args, world, rank, local_rank, dev)
    if args.workload == "encode4k" and world == 1 and not args.no_configs:
        # BASELINE's other configurations in the same line (each with its own timed steps; `value` stays configs[1]'s)
        out["configs"] = {}
        for label, name in CONFIG_RUNS:
            torch.cuda.empty_cache()
            sub = argparse.Namespace(**vars(args))
            sub.steps, sub.warmup, sub.workload = 3, 1, name
            sub.no_cpu = args.no_cpu or name != "single512"
            try:
                r = OTHER_WORKLOADS[name](sub, world, rank, local_rank, dev)
            except Exception as exc:  # a failing extra must not take the headline figure with it
                out["configs"][label] = {"error": repr(exc)}
                continue
            ent = {k: r[k] for k in CONFIG_KEYS if k in r}
            ent["workload"] = r["config"]["workload"]
            if r.get("decode"):
                ent["decode_value"] = r["decode"]["value"]
            if "predictor_variant" in r:
                ent["predictor_variant"] = {k: r["predictor_variant"][k] for k in ("value", "ms_per_step", "bits_per_pixel", "roundtrip_verified",
                                                                                      "stage_ms_per_step", "exhaustive_search_tiles")}
                if r["predictor_variant"].get("decode"):
                    ent["predictor_variant"]["decode_value"] = r["predictor_variant"]["decode"]["value"]
            out["configs"][label] = ent
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
