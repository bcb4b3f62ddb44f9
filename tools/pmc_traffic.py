#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) into HBM
bytes per launch per kernel.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for 16-B-per-lane stores.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_traffic.json
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(conv_igemm_kernel|conv_wgrad_kernel)<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        dt = "bf16" if a[0] == "unsigned short" else "f32"
        if m.group(1) == "conv_wgrad_kernel":
            return "conv_wgrad<%s,%dx%d>" % (dt, 64 * int(a[2]), 64 * int(a[3]))
        tile = {("2", "2"): "128c_x_128p", ("2", "4"): "128c_x_256p", ("4", "2"): "256c_x_128p", ("4", "4"): "256c_x_256p"}.get((a[1], a[2]))
        if tile is None:
            tile = "%dc_x_256p" % (16 * int(a[3]))
        return "conv_igemm<%s,%s>" % (dt, tile)
    if "conv_cin8_kernel" in name:
        return "conv_cin8<bf16>"
    if "convt_thin_kernel" in name:
        return "convt_thin<bf16>"
    m = re.search(r"([a-z_0-9]+_kernel)", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [1])), 1)
    w = sum(write.get(k, [0])) / max(len(write.get(k, [1])), 1)
    out[k] = {"launches": len(fetch.get(k, [])), "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
              "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
# the kernel timer of bench.py keys all filter-gradient tiles as one kernel: launch-weighted mean over the tiles
wg = [v for k, v in out.items() if k.startswith("conv_wgrad<bf16,")]
if wg:
    n = sum(v["launches"] for v in wg)
    out["conv_wgrad<bf16>"] = {"launches": n, "hbm_bytes_per_launch": int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in wg) / max(n, 1)),
                               "fetch_size_kb": round(sum(v["fetch_size_kb"] * v["launches"] for v in wg) / max(n, 1), 1),
                               "write_size_kb": round(sum(v["write_size_kb"] * v["launches"] for v in wg) / max(n, 1), 1)}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
