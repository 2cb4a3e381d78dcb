#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the render kernel from rocprofv3 counter_collection CSVs.

  python tools/pmc_summary.py DIR [DIR ...] [--json profiles/traffic_rNN.json --workload "..."]
Each DIR is the -d directory of one `rocprofv3 --pmc ...` pass.  With --json the merged per-launch averages
are written in the form bench.py reads for roofline.traffic: HBM bytes = FETCH_SIZE*1024 (raw, see
profiles/r01_fetch_size_calibration.md: the kernel's reads are scattered 16-byte gathers, which FETCH_SIZE
counts exactly) + WRITE_SIZE*1024."""
import argparse
import collections
import csv
import glob
import json

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--json")
ap.add_argument("--workload", default="")
ap.add_argument("--kernel", default="render")
args = ap.parse_args()
merged, kernel_names, launches = {}, set(), {}
for d in args.dirs:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if args.kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kernel_names.add(r["Kernel_Name"].split("(")[0].split("::")[-1])
        print(d, " ".join("%s=%.4g (n=%d)" % (k, sum(v) / len(v), len(v)) for k, v in sorted(acc.items())))
        for k, v in acc.items():
            merged[k] = sum(v) / len(v)
            launches[k] = len(v)
if args.json:
    fetch = merged.get("FETCH_SIZE", 0.0) * 1024.0
    write = merged.get("WRITE_SIZE", 0.0) * 1024.0
    out = {
        "kernel": ", ".join(sorted(kernel_names)),
        "workload": args.workload,
        "counters_per_launch": {k: round(v, 2) for k, v in sorted(merged.items())},
        "dispatches_averaged": launches.get("FETCH_SIZE"),
        "fetch_bytes_raw": round(fetch, 2),
        "write_bytes": round(write, 2),
        "hbm_bytes_per_launch": round(fetch + write, 2),
        "note": "separate rocprofv3 --pmc passes (tools/profile_round.sh), averaged over the render-kernel dispatches. "
                "FETCH_SIZE/WRITE_SIZE are in KiB.  Calibrated on this access pattern "
                "(profiles/r01_fetch_size_calibration.md): scattered 16-B gathers read exactly their 64-B sectors "
                "(ratio 0.99-1.00), coalesced streams read 1/2; the kernel's reads are scattered gathers, so the raw "
                "value is used.",
    }
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
