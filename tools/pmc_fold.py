#!/usr/bin/env python3
"""Fold rocprofv3 counter-collection CSVs (one --pmc pass each, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE cannot share a pass) into one per-kernel JSON.

    python tools/pmc_fold.py <pass1_counter_collection.csv> [<pass2...> ...] > profiles/rNN_pmc_<model>.json

Per kernel (names as bench.py's `kernels` table): launches, average duration from the dispatch timestamps of the
profiled pass (profiled passes run at a lower clock than un-profiled ones: use for ratios, not for headline times),
every counter averaged per launch, and
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024      gfx950: FETCH_SIZE tallies a 128-B request as 64 B on
                                                                    wide coalesced streams (LDS-DMA and global_load alike)
  mfma_busy_frac       = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * GRBM_GUI_ACTIVE / 8)
                         (GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs)
"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)


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
        bn = "+bn_bwd" if len(a) > 7 and a[7] == "true" else ""      # the BatchNorm hand-over epilogue variant
        return "conv_igemm<%s,%s>%s" % (dt, tile, bn)
    m = re.search(r"conv_halo_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        return "conv_halo<bf16,%sc_x_256p>%s" % (m.group(1), "+bn_bwd" if int(m.group(4)) & 64 else "")
    if "conv_wgrad_halo_kernel" in name:
        return "conv_wgrad_halo<bf16>"
    if "conv_cin8_kernel" in name:
        return "conv_cin8<bf16>"
    if "convt_thin_kernel" in name:
        return "convt_thin<bf16>"
    m = re.search(r"([a-z_0-9]+_kernel)", name)
    return m.group(1) if m else name[:40]


def main(paths):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> [per dispatch]
    durs = collections.defaultdict(list)
    for path in paths:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"], path)
            if key not in seen:
                seen.add(key)
                durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k in sorted(vals):
        c = {n: sum(v) / len(v) for n, v in vals[k].items()}
        rec = {"launches_profiled": max(len(v) for v in vals[k].values()), "avg_us_profiled": round(sum(durs[k]) / max(len(durs[k]), 1), 2),
               "counters_per_launch": {n: round(v, 1) for n, v in sorted(c.items())}}
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rec["hbm_bytes_per_launch"] = int((2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            rec["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * c["GRBM_GUI_ACTIVE"] / 8), 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            rec["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        out[k] = rec
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
