#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/<tag>/) into the committed
summaries under profiles/:
  <tag>_kernel_stats_bench.csv   per-kernel calls / total / average duration of `bench.py --steps 3 --warmup 1`
  <tag>_pmc_match_kernel.json    FETCH_SIZE / WRITE_SIZE of one k_lz77_sparse launch (separate passes)
  lz77_traffic.json              what bench.py reports as roofline.traffic for the same workload
FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a 16-byte-per-lane streaming read at half
its bytes (128-byte requests tallied as 64), so it is doubled; WRITE_SIZE is exact for this kernel (it equals
9 bytes x coded pixels to the byte: residual 4, match 4, advance 1)."""
import csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats_bench.csv"))
for wl in ("decode1080p", "palette2k", "single16k", "predictor4k"):
    f = os.path.join(src, "trace_" + wl, "trace_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, wl)))

def rows_of(kind, name):
    return [r for r in csv.DictReader(open(os.path.join(src, kind, kind + "_counter_collection.csv"))) if r["Counter_Name"] == name]

def one(kind, name, kernel="k_lz77_sparse"):
    rows = [r for r in rows_of(kind, name) if kernel in r["Kernel_Name"]]
    assert rows, (kind, name, kernel)
    r = max(rows, key=lambda r: int(r["Grid_Size"]))   # the headline batch's launch
    return float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"].split("(")[0]

fetch, fms, kname = one("fetch", "FETCH_SIZE")
write, wms, _ = one("write", "WRITE_SIZE")
batch, w, h = 256, 3840, 2160
# the other per-pixel encode kernels of the same launch sequence (same corrections)
others = {}
for r in rows_of("fetch", "FETCH_SIZE"):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1]
    others.setdefault(k, {})["fetch_bytes_corrected_x2"] = 2.0 * float(r["Counter_Value"]) * 1024
for r in rows_of("write", "WRITE_SIZE"):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1]
    others.setdefault(k, {})["write_bytes"] = float(r["Counter_Value"]) * 1024
for k, v in others.items():
    v["bytes_per_pixel"] = (v.get("fetch_bytes_corrected_x2", 0.0) + v.get("write_bytes", 0.0)) / (batch * w * h)
hbm = 2.0 * fetch * 1024 + write * 1024
pmc = {
    "command": "rocprofv3 --pmc <counter> --kernel-include-regex k_lz77_sparse -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-verify (one pass per counter)",
    "kernel": kname, "workload": {"batch": batch, "width": w, "height": h},
    "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
    "fetch_bytes_corrected_x2": 2.0 * fetch * 1024, "write_bytes": write * 1024,
    "hbm_bytes_per_launch": hbm, "launch_ms_under_counters": [fms, wms],
    "algorithmic_bytes_per_launch": 13.0 * batch * w * h,
    "per_pixel_kernels": others,
}
json.dump(pmc, open(os.path.join(dst, tag + "_pmc_match_kernel.json"), "w"), indent=1)
json.dump({"batch": batch, "width": w, "height": h, "kernel": kname, "hbm_bytes_per_launch": hbm,
           "source": "profiles/%s_pmc_match_kernel.json" % tag}, open(os.path.join(dst, "lz77_traffic.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1))
