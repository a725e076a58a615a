#!/usr/bin/env python3
"""Condense rocprofv3 output of `bench.py` runs into the files kept under profiles/.

    python tools/pmc_summary.py --tag r1 --workload c3 --fft f64 \
        --stats gpurun_out/prof/x_kernel_stats.csv \
        --pmc gpurun_out/pmc_fetch/x_counter_collection.csv gpurun_out/pmc_write/x_counter_collection.csv

Writes profiles/<tag>_kernel_stats.csv (the library's kernels only) and
profiles/pmc_traffic.json: per kernel, mean FETCH_SIZE and WRITE_SIZE per launch
(KiB as the profiler reports them) and hbm_bytes_per_launch = 2 * FETCH + WRITE
in bytes - on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes.
"""
import argparse
import csv
import json
import os
import re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 over the kernel sources: the counters in pmc_traffic.json describe exactly these files."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "image-stitcher_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".hpp", ".h")):
            with open(os.path.join(src, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def short_name(full):
    """'void mistitch::k_fuse<1, 1, 4>(...)' -> the name the library's own profiler uses."""
    m = re.search(r"mistitch::(k_[a-z0-9_]+)(<([^>]*)>)?", full)
    if not m:
        return None
    name = m.group(1)
    if name == "k_fuse":
        mode = m.group(3).split(",")[0].strip()
        return "k_fuse_feather" if mode == "1" else "k_fuse_overwrite"
    aliases = {"k_transpose_strips64": "k_transpose_strips", "k_stage_strips64": "k_stage_strips",
               "k_stage_strips_any": "k_stage_strips", "k_edge_stats_staged": "k_edge_stats",
               "k_fft_long_fwd_reg": "k_fft_long_fwd", "k_fft_short_cps_reg": "k_fft_short_cps",
               "k_fft_long_inv_reg": "k_fft_long_inv", "k_ncc_sweep": "k_ncc_accumulate"}
    if name in aliases:
        return aliases[name]
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--fft", default="f64")
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    if args.stats:
        with open(args.stats) as fh:
            rows = list(csv.reader(fh))
        keep = [rows[0]] + [r for r in rows[1:] if "mistitch::" in r[0]]
        with open(os.path.join(out_dir, f"{args.tag}_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)

    sums = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(set))
    for path in args.pmc:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = short_name(row["Kernel_Name"])
                if name is None:
                    continue
                sums[name][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[name][row["Counter_Name"]].add(row["Dispatch_Id"])
    if sums:
        kernels = {}
        for name, counters in sorted(sums.items()):
            rec = {}
            for cname, total in counters.items():
                rec[cname + "_KiB_per_launch"] = total / len(launches[name][cname])
            fetch = rec.get("FETCH_SIZE_KiB_per_launch")
            write = rec.get("WRITE_SIZE_KiB_per_launch")
            if fetch is not None and write is not None:
                rec["hbm_bytes_per_launch"] = int((2.0 * fetch + write) * 1024)
                rec["launches_seen"] = len(launches[name]["FETCH_SIZE"])
                kernels[name] = rec
        doc = {"tag": args.tag, "workload": args.workload, "fft": args.fft, "kernel_source_sha256": kernel_source_hash(),
               "formula": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes",
               "kernels": kernels}
        with open(os.path.join(out_dir, "pmc_traffic.json"), "w") as fh:
            json.dump(doc, fh, indent=1, sort_keys=True)
            fh.write("\n")


if __name__ == "__main__":
    main()
