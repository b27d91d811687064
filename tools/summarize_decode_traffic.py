#!/usr/bin/env python3
"""Turns the counter passes of tools/collect_decode_traffic.sh (gpurun_out/<tag>/) into profiles/<tag>_pmc_decode1080p.json and
profiles/decode_traffic.json (what bench.py reports as the decode roofline's `traffic` for the same workload).
FETCH_SIZE / WRITE_SIZE are KiB summed over the launches of one decode call (bench.py --steps 1 --warmup 0 makes exactly one
timed call; the kernels of the encode that prepares the streams are not in the filter).  FETCH_SIZE counts a 16-byte-per-lane
streaming read at half its bytes on gfx950 and is doubled for the kernels that read that way (the scouts' whole-line stream
reads, the placing kernel's record reads); for scattered 4- and 8-byte accesses (pointer jumping, record stores) the counters are
uncalibrated, so both the raw and the doubled figure are kept."""
import csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
batch, w, h = 512, 1920, 1080

def per_kernel(kind, name):
    acc, launches = {}, {}
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"]) * 1024
            launches[k] = launches.get(k, 0) + 1
    return acc, launches

fetch, fl = per_kernel("dfetch", "FETCH_SIZE")
write, wl = per_kernel("dwrite", "WRITE_SIZE")
# the passes run the whole benchmark: an untimed warm call is not made with --warmup 0, the verification decode is one more
# call, so every kernel appears `calls` times; figures below are per call
kernels = {}
for k in sorted(set(fetch) | set(write)):
    kernels[k] = {"fetch_bytes_raw": fetch.get(k, 0.0), "fetch_bytes_x2": 2 * fetch.get(k, 0.0), "write_bytes": write.get(k, 0.0),
                  "launches_fetch_pass": fl.get(k, 0), "launches_write_pass": wl.get(k, 0)}
json.dump({"workload": {"batch": batch, "width": w, "height": h}, "kernels_all_calls": kernels,
           "command": "rocprofv3 --pmc <counter> --kernel-include-regex 'k_par_|k_decode_headers|k_decode_entropy|k_inverse' -- python3 bench.py --workload decode1080p --steps 1 --warmup 0 --no-cpu"},
          open(os.path.join(root, "profiles", tag + "_pmc_decode1080p.json"), "w"), indent=1)
print(json.dumps(kernels, indent=1))

# What bench.py reports as the decode roofline's traffic: one call's bytes.  Reads doubled where the kernel streams 16 bytes per
# lane (scouts, placing: see above), raw elsewhere; writes as counted.
calls = max(1, fl.get("k_par_scout_lanes", 1))
doubled = ("k_par_scout_lanes", "k_par_place")
f_total = sum((2.0 if k in doubled else 1.0) * v for k, v in fetch.items()) / calls
w_total = sum(write.values()) / calls
json.dump({"batch": batch, "width": w, "height": h, "hbm_bytes_per_call": f_total + w_total, "fetch_bytes": f_total, "write_bytes": w_total,
           "note": "FETCH_SIZE doubled for k_par_scout_lanes and k_par_place (16-byte-per-lane streaming reads, the gfx950 correction); raw for "
                   "k_par_jump (4-byte reads at random: uncalibrated); WRITE_SIZE as counted: the scouts' 16-byte record stores, one line per "
                   "lane, are counted as whole 64-byte requests (17 GB for about 5 GB of records)",
           "source": "profiles/%s_pmc_decode1080p.json" % tag},
          open(os.path.join(root, "profiles", "decode_traffic.json"), "w"), indent=1)
